"""Prompt ingestion rate of dsk_hydrate on the full DeepSeek-V3 Q2_K model (61 blocks, 256 experts; the engine's DEFAULT options
since round 6: the layout bench.py times for decode, the batched path on tile copies of the plane matrices), next to the
single-token rate of the same model: what the batched path buys over the reference's one forward per prompt token
(src/main.cpp:312-319).

  python tools/hydrate_bench.py [--P 16,64,128] [--layers 0] [--reps 2] [--attn mla] [--pos0 N] [--opt q2k_tiles=2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepseek.cpp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def measure(ctx, c, Ps, reps=2, seed=0, chunk=0, opts=None, loop=True, pos0=0):
    import dsk
    o = {}
    if chunk:
        o["hydrate_chunk"] = chunk  # (0: the engine's default)
    o.update(opts or {})
    M = dsk.Model(ctx, c, None, synth_seed=seed, options=o)
    why = M.hydrate_why_not()
    rng = np.random.default_rng(1)
    out = {"why_not": why, "device_gb": round(M.device_bytes() / 1e9, 1)}
    # the loop's rate: single-token forwards in HYDRATE mode (no classifier), graph replay
    toks = rng.integers(0, c.vocab_size, 40)
    dt = None
    if loop:
        for i in range(8):
            M.forward(int(toks[i]), i, dsk.MODE_HYDRATE_KV_CACHE)
        t0 = time.perf_counter()
        for i in range(8, 40):
            M.forward(int(toks[i]), i, dsk.MODE_HYDRATE_KV_CACHE)
        dt = (time.perf_counter() - t0) / 32
        out["loop_tok_s"] = round(1.0 / dt, 1)
    res = {}
    for P in Ps:
        toks = rng.integers(0, c.vocab_size, P)
        M.hydrate(toks, pos0, dsk.MODE_HYDRATE_KV_CACHE)  # warm-up (allocations, tile copies, code objects)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            M.hydrate(toks, pos0, dsk.MODE_HYDRATE_KV_CACHE)
            best = min(best, time.perf_counter() - t0)
        res[str(P)] = {"ms": round(best * 1e3, 3), "tok_s": round(P / best, 1), "x_loop": round(P / best * dt, 2) if dt else None}
    out["hydrate"] = res
    # distinct experts the LAST chunk's last MoE block touched (the bytes a chunk streams are proportional to it)
    try:
        cap = chunk or 512
        last = (Ps[-1] - 1) % cap + 1  # tokens of the last chunk
        re = M.hydrate_buffer("route_e", 0, last, max(1, c.n_active_routed), np.int32)
        cnt = np.bincount(re.reshape(-1), minlength=c.n_routed_experts)
        out["last_chunk_experts"] = {"distinct": int((cnt > 0).sum()), "max_tokens": int(cnt.max())}
    except Exception as e:  # noqa: BLE001
        out["last_chunk_experts"] = {"error": repr(e)[:80]}
    out["batched_tokens"] = M.info("hydrate_batched_tokens")
    out["looped_tokens"] = M.info("hydrate_looped_tokens")
    out["q2k_tiles"] = o.get("q2k_tiles", 1)
    out["tile_copy_mb"] = M.info("hydrate_tile_copy_mb")
    out["pos0"] = pos0
    M.close()
    return out


def main():
    import dsk
    from tools import synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", default="16,64,128")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=0, help="hydrate_chunk (0: the engine's default)")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--no-loop", action="store_true", help="skip the per-token loop (a kernel trace of the batched path only)")
    ap.add_argument("--attn", default="mha", choices=["mha", "mla"])
    ap.add_argument("--pos0", type=int, default=0, help="first position of the timed prompt (rows below it: whatever the cache holds)")
    a = ap.parse_args()
    c = synth.preset("v3", "q2_k", a.attn == "mla")
    if a.layers:
        c.n_layers = a.layers
        c.first_k_dense_replace = min(c.first_k_dense_replace, a.layers)
    c.max_seq_len = 1100
    ctx = dsk.Ctx(0)
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt}
    print(json.dumps(measure(ctx, c, [int(p) for p in a.P.split(",")], a.reps, chunk=a.chunk, opts=opts, loop=not a.no_loop, pos0=a.pos0)))


if __name__ == "__main__":
    main()
