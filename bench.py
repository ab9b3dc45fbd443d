#!/usr/bin/env python
"""bench.py -- DeepSeek-V3 Q2_K batch-1 decode on MI355X: tok/s + achieved HBM GB/s vs roofline.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one decoded token (one dsk_forward: embed -> 61 blocks -> final norm -> lm_head ->
logits on the host), BASELINE.json configs[3]: DeepSeek-V3 shapes, Q2_K, 256 routed experts,
batch = 1.  No real weights exist on the machine (and 220 GB does not fit host storage), so the
weights are valid random Q2_K blocks generated directly in HBM; token ids are
U[0, vocab) with seed 0.  Inputs (weights, KV) are resident in HBM when the timed region starts.

N > 1: one process per GPU; routed experts are sharded across ranks (expert e lives on rank
e // ceil(E/N)), everything else is replicated, one RCCL all-reduce of the per-slot expert outputs
per MoE layer (include/dsk.h dsk_comm_init).  All ranks decode the SAME token stream, so the total
work is fixed as N grows: "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) extended with
  "roofline":     dominant kernel class, algorithmic bytes / launch / its HIP-event duration
  "cpu_baseline": the unmodified reference (oracle/_ref) timed on this box's host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))



def csrc_sha() -> str:
    """sha256 (16 hex) over the kernel / engine sources: ties a committed profile to the code it was taken on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "deepseek.cpp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":  # the Makefile carries code-generation flags
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable
README_TOK_S = 4.02     # reference README.md:26 (V3 Q2_K, MHA path, EPYC 7R13 16 threads) -- BASELINE.md section 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="v3", choices=["v3", "v2lite", "tiny_v3"])
    ap.add_argument("--quant", default="q2_k")
    ap.add_argument("--attn", default="mha", choices=["mha", "mla"], help="reference attention path (BlockMHA is the README's)")
    ap.add_argument("--layers", type=int, default=0, help="override n_layers (0 = the model's own depth)")
    ap.add_argument("--ctx", type=int, default=4200, help="KV-cache allocation (max_seq_len); the kv_len sweep needs > 4096")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (MLA path, kv_len sweep)")
    ap.add_argument("--dry-shard", default="", help="R/W: single-process dry run of expert shard R of W (no communicator)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="KEY=VALUE model option (include/dsk.h dsk_model_set_option), repeatable: A/B runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--profile-steps", type=int, default=4)
    return ap.parse_args()


def cpu_baseline(cfg_full, seconds: float):
    """Time the reference's own OpenMP CPU path on this box.

    A full-depth V3 checkpoint needs 220 GB of host RAM, so the sample is a reduced-depth,
    full-width checkpoint (1 dense + 1 MoE block, all 256 routed experts resident when the host has the RAM,
    8 active, full vocab) whose per-block times are extrapolated to the model depth -- labelled as such.
    """
    from oracle import orc
    from tools import synth
    try:
        R = orc.Ref()
        kind = "reference"
    except Exception as e:  # not built / not loadable on this host
        return dict(value=None, unit="tok/s", cores=0, kind="unavailable", sample=str(e)[:120])
    ncpu = os.cpu_count() or 1
    # README.md:79-84: the reference is fastest well below the core count (its OpenMP loops are short);
    # a short sweep picks the best setting on this box
    sweep = sorted({t for t in (8, 16, 32, 64, max(1, ncpu // 2)) if t <= ncpu})
    c = synth.preset(cfg_full.model_name, cfg_full.quant, cfg_full.use_mla)
    n_dense = min(c.first_k_dense_replace, c.n_layers)
    n_moe = c.n_layers - n_dense
    c.n_layers = (1 if n_dense else 0) + (1 if n_moe else 0)
    c.first_k_dense_replace = 1 if n_dense else 0
    # ALL routed experts stay resident (12 GB for the one MoE block of V3 Q2_K): with a few dozen experts the whole
    # checkpoint sits in the host's L3 (512 MB on the bench box) and the CPU path looks 2-3x faster than it is on the
    # real 220 GB model.  Needs ~30 GB of host RAM / tmp space; falls back to 32 experts when that is not there.
    try:
        free_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 1e9
        st = os.statvfs(tempfile.gettempdir())
        tmp_gb = st.f_bavail * st.f_frsize / 1e9
    except (ValueError, OSError):
        free_gb = tmp_gb = 0.0
    if c.n_routed_experts > 32 and (free_gb < 64 or tmp_gb < 24):
        c.n_routed_experts = 32
        c.n_group = min(c.n_group, 8)
    c.max_seq_len = 1024  # >= warm-up + the tokens the thread sweep decodes (160 + 5 settings x <= 64)
    T = synth.random_block_model(c, seed=0, tile_blocks=1 << 20)
    d = tempfile.mkdtemp(prefix="dsk_cpu_baseline_")
    try:
        synth.write_dseek(d, c, T, tokenizer=True)
        del T
        S = R.session(d, c, context=1024)
        import ctypes as C
        R.lib.ref_forward_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        per = (C.c_double * (c.n_layers + 1))()
        tok = np.random.default_rng(0).integers(0, c.vocab_size, 4096)
        # warm-up: every expert's pages must be mapped before timing (the reference mmaps the file: a cold expert
        # costs ~10 ms of minor faults).  Random tokens through a random router reach ~99 % of 256 experts in 160 steps.
        R.set_threads(min(32, ncpu))
        pos = 0
        for _ in range(160 if c.n_routed_experts > 32 else 4):
            R.lib.ref_forward_timed(S.h, int(tok[pos]), pos, per)
            pos += 1
        best = None
        for threads in sweep:
            R.set_threads(threads)
            acc = np.zeros(c.n_layers + 1)
            n, t0 = 0, time.time()
            while time.time() - t0 < seconds / len(sweep) and n < 64:
                R.lib.ref_forward_timed(S.h, int(tok[pos]), pos, per)
                acc += np.array(list(per))
                n += 1
                pos += 1
            acc /= max(n, 1)
            t_dense = acc[0] if n_dense else 0.0
            t_moe = acc[c.n_layers - 1] if n_moe else 0.0
            t_tok = n_dense * t_dense + n_moe * t_moe + acc[c.n_layers]
            if best is None or t_tok < best[0]:
                best = (t_tok, threads, n, t_dense, t_moe, acc[c.n_layers])
        t_tok, threads, n, t_dense, t_moe, t_head = best
        S.close()
        main_cli = main_cli_tok_s(d, threads)
    finally:
        for f in os.listdir(d):
            os.unlink(os.path.join(d, f))
        os.rmdir(d)
    return dict(value=round(1.0 / t_tok, 4), unit="tok/s", cores=threads, kind=kind, main_cli_tok_s=main_cli,
                sample=(f"unmodified reference (oracle/_ref, OpenMP, best of {sweep} threads on {ncpu} cpus = {threads}): {n} tokens on a "
                        f"{cfg_full.model_name}-shaped {c.quant} checkpoint with {c.n_layers} blocks "
                        f"({c.n_routed_experts} experts resident), per-block times "
                        f"[dense {t_dense*1e3:.2f} ms, moe {t_moe*1e3:.2f} ms, head {t_head*1e3:.2f} ms] "
                        f"extrapolated to {n_dense}+{n_moe} blocks"))


def main_cli_tok_s(ckpt_dir: str, threads: int):
    """the reference's own binary, `main DIR -m c -n 128 -t 0`, on the SAME reduced-depth checkpoint
    the shim was timed on (oracle/_ref/main: the unmodified reference built by oracle/Makefile).  Its printed throughput
    is for that checkpoint (2 blocks), NOT extrapolated: it sits beside the per-block figures as a cross-check of the shim."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "main")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, ckpt_dir, "-m", "c", "-n", "128", "-t", "0", "-i", "the quick brown fox jumps over the lazy dog"],
                           capture_output=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads)), timeout=120)
        m = re.search(r"Generation stats:\s+(\d+) tokens\s+throughput: ([0-9.eE+-]+)tok/s", r.stdout.decode("latin-1"))
        return dict(tok_s=float(m.group(2)), tokens=int(m.group(1)), threads=threads,
                    note="reference main -m c -n 128 -t 0 on the reduced-depth checkpoint itself (not extrapolated)") if m else None
    except Exception as e:  # noqa: BLE001 - a baseline helper must never fail the bench line
        return dict(error=repr(e)[:120])


# name -> (rows, n, tasks, kind, act_mode) of dsk_bench_gemv: the quantised GEMVs of a DeepSeek-V3 Q2_K token, as the model
# launches them (kind 0 plain, 1 GLU pair, 3 W2 of all slots with the fused combine; act 0 ready Q8_K, 1 f32, 2 f32 + rmsnorm)
GEMV_SHAPES = {"lm_head": (129280, 7168, 1, 0, 2), "wo": (7168, 16384, 1, 0, 0), "dense_w13": (18432, 7168, 1, 1, 2),
               "dense_w2": (7168, 18432, 1, 0, 1), "experts_w13": (2048, 7168, 9, 1, 0), "experts_w2": (7168, 2048, 9, 3, 1)}


def gemv_fracs(ctx, measured_bw):
    """north_star: ">= 70 % of the measured HBM roofline for the quantized GEMV", per shape, driver-checkable: each GEMV of
    the token alone on rotating weight sets (> 512 MB, so HBM and not the Infinity Cache), GB/s = weight bytes / launch."""
    out = {}
    for name, (rows, n, nt, kind, act) in GEMV_SHAPES.items():
        try:
            us, nb = ctx.bench_gemv(3, rows, n, nt, kind, act, 0, 0, 0, 0, 30)
            out[name] = dict(us=round(us, 2), gbps=round(nb / us / 1e3, 1), frac=round(nb / us / 1e3 / measured_bw, 4) if measured_bw else None)
        except Exception as e:  # noqa: BLE001
            out[name] = dict(error=repr(e)[:100])
    return out


def small_model_line(dsk, synth, ctx, torch, tokens, quant, measured_bw):
    """BASELINE.json configs C2 / C3: DeepSeek-V2-Lite (27 blocks, 64 experts top-6) on one GPU, full depth."""
    c = synth.preset("v2lite", quant, False)
    c.max_seq_len = 256
    M = dsk.Model(ctx, c, None, synth_seed=0)
    pos = 0
    for _ in range(8):
        M.forward_nocopy(int(tokens[pos]) % c.vocab_size, pos)
        pos += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 64
    for _ in range(n):
        M.forward_nocopy(int(tokens[pos % len(tokens)]) % c.vocab_size, pos)
        pos += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ab = M.active_bytes(8 + n // 2)
    M.close()
    return dict(tok_s=round(1.0 / dt, 1), ms_per_step=round(dt * 1e3, 4), algo_bytes_per_token=round(ab),
                token_frac=round(ab / dt / 1e9 / HBM_PEAK_GBPS, 4), frac_of_measured=round(ab / dt / 1e9 / measured_bw, 4) if measured_bw else None)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, the way the driver does (one process per
        # GPU, RCCL over xGMI), and pass their exit code on - never print a 1-GPU line for N > 1
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch  # plumbing only: rendezvous, barrier, max-reduce; loaded first so one HIP runtime is shared
    import torch.distributed as dist
    import dsk
    from tools import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    launched = "WORLD_SIZE" in os.environ  # under torch.distributed.run (also with one rank): the rank plumbing runs
    if launched:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if launched:
            dist.barrier()

    ctx = dsk.Ctx(local_rank)
    if launched:
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
    elif a.dry_shard:  # the sharded code path of rank R of W on one GPU (the exchange is skipped: timing only)
        r_, w_ = (int(v) for v in a.dry_shard.split("/"))
        ctx.comm_init_dry(r_, w_)

    c = synth.preset(a.model, a.quant, a.attn == "mla")
    c.model_name = a.model
    if a.layers > 0:
        c.n_layers = a.layers
        c.first_k_dense_replace = min(c.first_k_dense_replace, a.layers)
    c.max_seq_len = max(a.ctx, a.steps + a.warmup + a.profile_steps + 2)
    t_build = time.time()
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt}
    M = dsk.Model(ctx, c, None, synth_seed=0, options=opts)
    t_build = time.time() - t_build
    if a.no_graph:
        M.set_graph(False)
    tokens = np.random.default_rng(0).integers(0, c.vocab_size, a.steps + a.warmup + a.profile_steps + 1)

    pos = 0
    for _ in range(a.warmup):
        M.forward_nocopy(int(tokens[pos]), pos)
        pos += 1
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        M.forward_nocopy(int(tokens[pos]), pos)  # includes the logits D2H copy
        pos += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    tok_s = a.steps / dt
    # SURVEY 8d defines the rate over >= 128 decode steps: whatever --steps was, the line also carries that figure (positions
    # continue behind the timed region; the MHA cache grows 5 MB per position, so it sits ~2 % under a 20-step figure)
    tok_s_128 = None
    if world == 1 and not a.no_extras and not a.dry_shard:
        p128 = pos
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(128):
            M.forward_nocopy(int(tokens[p128 % len(tokens)]), p128)
            p128 += 1
        torch.cuda.synchronize()
        tok_s_128 = dict(tok_s=round(128 / (time.perf_counter() - t1), 3), steps=128, positions=[pos, p128 - 1])
    mid_pos = a.warmup + a.steps // 2
    algo_bytes = M.active_bytes(mid_pos)
    M_device_gb = M.device_bytes() / 1e9

    # ---- roofline of the dominant kernel: eager forwards bracketed by HIP events on the engine stream
    agg = {}
    for _ in range(a.profile_steps):
        for k in M.profile_forward(int(tokens[pos]), pos):
            g = agg.setdefault(k["name"], dict(launches=0, total_ms=0.0, algo_bytes=0.0))
            g["launches"] += k["launches"]
            g["total_ms"] += k["total_ms"]
            g["algo_bytes"] += k["algo_bytes"]
        pos += 1
    roof, kernels = None, {}
    if agg:
        # per-class duration inside the model: the class's launches of a token, back to back between two
        # HIP events on the engine stream (include/dsk.h dsk_time_kernel_class); the eager per-launch events
        # above only discover the classes (an event pair around every launch adds ~8 us of queue bubbles)
        for name, g in agg.items():
            per_tok = g["launches"] // a.profile_steps
            # GEMV classes: the eager pass above used hipExtLaunchKernel start / stop events, i.e. each kernel's
            # own dispatch timestamps inside the real token sequence (what rocprofv3 --kernel-trace reports);
            # other classes: their launches of a token, back to back between two events (dsk_time_kernel_class)
            us, nb, n_l = g["total_ms"] / g["launches"] * 1e3, g["algo_bytes"] / g["launches"], per_tok
            if not (name.startswith("gemv_") or name == "moe_ffn"):  # moe_ffn: its own dispatch timestamps too
                try:
                    us, nb, n_l = M.time_kernel_class(name, pos, reps=6)
                except dsk.DskError:
                    pass
            kernels[name] = dict(launches_per_step=n_l, us_per_launch=round(us, 2), ms_per_step=round(us * n_l / 1e3, 4),
                                 bytes_per_launch=round(nb), gbps=round(nb / max(us, 1e-9) / 1e3, 1))
        dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
        k = kernels[dom]
        # HBM traffic is NOT measured in this run: PMC counters need their own rocprofv3 passes.  The figure of the committed
        # pass is reported only if it was taken on these very kernel sources (sha of deepseek.cpp_amd/csrc), and labelled.
        traffic, traffic_source = None, "not measured in this run (no committed PMC pass for these kernel sources)"
        import glob
        for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):  # the newest round's first
            try:
                pm = json.load(open(pmc_path))
                if pm.get("csrc_sha") == csrc_sha():
                    traffic = pm.get("traffic_bytes_per_launch", {}).get(dom)
                    traffic_source = (f"profiles/{os.path.basename(pmc_path)} (separate rocprofv3 --pmc passes on the same kernel sources, "
                                      f"csrc sha {pm.get('csrc_sha')})")
                    break
            except Exception:
                pass
        roof = dict(bound="hbm", kernel=dom, achieved=k["gbps"], peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(k["gbps"] / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_source,
                    bytes_per_launch=k["bytes_per_launch"], avg_launch_us=k["us_per_launch"],
                    launches_per_token=k["launches_per_step"],
                    token_gbps=round(algo_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                    token_frac=round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4))
    def sweep(model, tag):
        """ms per decode step at kv_len 128 / 1024 / 4096: 6 steps each after 2 of warm-up; the cache rows
        below the position hold whatever earlier steps left (zeros mostly): attention streams them all the same"""
        out = {}
        for kv in (128, 1024, 4096):
            if kv + 10 > c.max_seq_len:
                continue
            p0 = kv - 1
            for i in range(2):
                model.forward_nocopy(int(tokens[i]), p0 + i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(6):
                model.forward_nocopy(int(tokens[2 + i]), p0 + 2 + i)
            torch.cuda.synchronize()
            out[str(kv)] = round((time.perf_counter() - t0) / 6 * 1e3, 4)
        return out

    extras = {}
    full_v3 = a.model == "v3" and a.quant == "q2_k"
    if rank == 0 and world == 1 and not a.no_extras and not a.dry_shard:
        try:
            extras["kv_sweep"] = {a.attn: sweep(M, a.attn)}
        except Exception as e:
            extras["kv_sweep"] = {"error": repr(e)[:160]}
    measured_bw = None
    M_info = [None, None, None, None]
    try:
        M_info = [M.info("fused_moe_layers"), M.info("handoff_fallbacks"), M.info("tiled_tensors"), M.info("exchange_calls")]
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        try:
            M.close()
            measured_bw = round(ctx.measure_read_bw(4 << 30, 5), 1)
        except Exception:
            measured_bw = None
    if rank == 0 and world == 1 and not a.no_extras and not a.dry_shard and a.attn == "mha" and full_v3:
        # the north star's MLA path on the same shapes (BlockMLA, absorbed weights): its own model (226 GB: after the MHA one is freed)
        try:
            c2 = synth.preset(a.model, a.quant, True)
            if a.layers > 0:
                c2.n_layers = a.layers
                c2.first_k_dense_replace = min(c2.first_k_dense_replace, a.layers)
            c2.max_seq_len = c.max_seq_len
            M2 = dsk.Model(ctx, c2, None, synth_seed=0)
            p2 = 0
            for _ in range(a.warmup):
                M2.forward_nocopy(int(tokens[p2]), p2)
                p2 += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                M2.forward_nocopy(int(tokens[p2]), p2)
                p2 += 1
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            ab2 = M2.active_bytes(a.warmup + a.steps // 2)
            extras["mla"] = {"tok_s": round(a.steps / dt2, 3), "ms_per_step": round(dt2 / a.steps * 1e3, 4),
                             "algo_bytes_per_token": round(ab2),
                             "token_frac": round(ab2 / (dt2 / a.steps) / 1e9 / HBM_PEAK_GBPS, 4)}
            extras["kv_sweep"]["mla"] = sweep(M2, "mla")
            M2.close()
        except Exception as e:
            extras["mla"] = {"error": repr(e)[:160]}
    if rank == 0 and world == 1 and not a.no_extras and not a.dry_shard and full_v3:
        try:
            extras["gemv_frac_of_measured"] = gemv_fracs(ctx, measured_bw)
        except Exception as e:  # noqa: BLE001
            extras["gemv_frac_of_measured"] = {"error": repr(e)[:160]}
        v2 = {}
        for q in ("q2_k", "f8e5m2"):
            try:
                v2[q] = small_model_line(dsk, synth, ctx, torch, tokens, q, measured_bw)
            except Exception as e:  # noqa: BLE001
                v2[q] = {"error": repr(e)[:160]}
        extras["v2lite"] = v2
    if rank == 0 and world == 1 and not a.no_extras and not a.dry_shard and full_v3 and a.attn == "mha":
        # the prompt phase through dsk_hydrate (batched launches, every weight read once per chunk) on the same
        # model at the same options as the timed decode (since round 6 the default layout batches: tile copies of its plane
        # matrices, hydrate.cpp), next to the per-token loop of that model (what the reference does with a prompt,
        # src/main.cpp:312-319)
        try:
            from tools import hydrate_bench
            c3 = synth.preset(a.model, a.quant, False)
            if a.layers > 0:
                c3.n_layers = a.layers
                c3.first_k_dense_replace = min(c3.first_k_dense_replace, a.layers)
            c3.max_seq_len = 1100
            extras["hydrate"] = hydrate_bench.measure(ctx, c3, [16, 64, 128, 512], reps=2)
        except Exception as e:  # noqa: BLE001
            extras["hydrate"] = {"error": repr(e)[:160]}
    block_floor = None
    if rank == 0 and world == 1 and not a.no_extras and full_v3:
        # the floor of a five-launch block on THIS chip (tools/block_floor.hip: five dependent pure-streaming kernels with the
        # block's byte profile); built by __graft_entry__.build()
        try:
            import subprocess
            exe = os.path.join(ROOT, "tools", "_build", "block_floor")
            if os.path.exists(exe):
                outp = subprocess.run([exe, "58"], capture_output=True, timeout=120).stdout.decode()
                for line in outp.splitlines():
                    if line.startswith("256 x 16 waves:"):
                        block_floor = float(line.split(":")[1].split("us")[0])
        except Exception:  # noqa: BLE001
            block_floor = None
    if roof is not None:
        roof["block_floor_us"] = block_floor
        if block_floor and kernels:
            per_block = sum(kernels[k]["us_per_launch"] for k in ("gemv_qkv_a", "attn_mha", "gemv_wo", "router_gate", "moe_ffn") if k in kernels)
            roof["block_us"] = round(per_block, 2) if per_block else None
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(c, a.cpu_seconds)
        except Exception as e:
            cpu = dict(value=None, unit="tok/s", cores=0, kind="failed", sample=repr(e)[:200])
    if rank == 0:
        full = a.model == "v3" and a.layers in (0, 61) and a.quant == "q2_k"
        out = {
            "metric": ("decode tok/s (batch=1) + achieved HBM GB/s vs roofline, DeepSeek-V3 Q2_K" if full else
                       f"decode tok/s (batch=1) + achieved HBM GB/s vs roofline, {a.model} {a.quant} ({c.n_layers} blocks)"),
            "value": round(tok_s, 3), "unit": "tok/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": round(tok_s / README_TOK_S, 2) if (full and a.attn == "mha") else None,
            "dtype": "u2xs8->s32 (W2A8 integer dots, f32 scales/activations)" if a.quant in ("q2_k", "q3_k") else "f32",
            "data": "synthetic (valid random Q2_K blocks generated in HBM, random token ids seed 0)",
            "config": {"workload": f"DeepSeek-{a.model} {a.quant} batch=1 decode, {c.n_layers} blocks, "
                                   f"{c.n_routed_experts} routed experts top-{c.n_active_routed}, {a.attn.upper()} path, "
                                   f"pos {a.warmup}..{a.warmup + a.steps - 1}, logits D2H included",
                       "parallelism": (f"dry run of expert shard {a.dry_shard} on one GPU (no exchange)" if a.dry_shard else
                                       "1 GPU" if world == 1 else f"experts sharded over {world} GPUs (RCCL all-reduce per MoE layer; EXPERIMENTAL: never run on > 1 GPU)"),
                       "hip_graph": not a.no_graph, "model_build_s": round(t_build, 1),
                       "device_gb": round(M_device_gb, 1), "algo_bytes_per_token": round(algo_bytes)},
            "roofline": roof, "kernels": kernels, "measured_read_gbps": measured_bw,
            # the same two fractions against what THIS GPU streams with 16-byte loads (dsk_measure_read_bw), the north
            # star's "measured HBM-bandwidth roofline"; the per-kernel table carries each GEMV's own GB/s
            "frac_of_measured": ({"kernel": round(roof["achieved"] / measured_bw, 4), "token": round(roof["token_gbps"] / measured_bw, 4)}
                                 if roof and measured_bw else None),
            "cpu_baseline": cpu,
            "tok_s_128": tok_s_128,
            # which code ran: fused expert launches in the model, times a hand-off give-up retired them (0 on an unshared GPU),
            # weight tensors stored as matrix-pipe tiles (option q2k_tiles)
            "engine": {"fused_moe_layers": M_info[0], "handoff_fallbacks": M_info[1], "tiled_tensors": M_info[2], "exchange_calls": M_info[3]},
            "csrc_sha": csrc_sha(),
        }
        out.update(extras)
        print(json.dumps(out), flush=True)
    barrier()
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
