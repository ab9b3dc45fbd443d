"""Measurement tooling on CPU: bench.py's CLI contract and tools/prof_summary.py on a synthetic rocprofv3 database."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_cli_contract_and_loud_failure_without_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")  # no JSON line is printed for a run that did not happen


def test_prof_summary_on_a_synthetic_database(tmp_path):
    tr, pm = tmp_path / "trace", tmp_path / "pmc"
    os.makedirs(tr)
    os.makedirs(pm)
    con = sqlite3.connect(str(tr / "a.db"))
    con.execute("create table kernels (name text, start integer, end integer)")
    w13 = "void gemv_kernel<3, 1, 4, true, 16>(GemvLaunch const*, void const*)"
    con.executemany("insert into kernels values (?,?,?)", [(w13, 0, 24000), (w13, 100000, 126000), ("router_gate_kernel<2>(RouterArgs)", 30000, 40000)])
    con.commit()
    con.close()
    con = sqlite3.connect(str(pm / "b.db"))
    con.execute("create table counters_collection (kernel_name text, counter_name text, value real, dispatch_id integer)")
    rows = [("read_bw_kernel(x)", "FETCH_SIZE", (4 << 30) / 2048 / 8, 1)] * 8  # 8 XCD instances of one dispatch
    rows += [(w13, "FETCH_SIZE", 90e6 / 2048 / 8, 2)] * 8
    con.executemany("insert into counters_collection values (?,?,?,?)", rows)
    con.commit()
    con.close()
    out = str(tmp_path / "r99")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), "--trace", str(tr), "--pmc", str(pm), "--out", out, "--note", "t"],
                          stdout=subprocess.DEVNULL)
    k = json.load(open(out + "_kernel_trace.json"))["kernels"]
    assert k[0]["kernel"] == w13 and k[0]["calls"] == 2 and k[0]["avg_us"] == 25.0 and k[0]["min_us"] == 24.0
    p = json.load(open(out + "_pmc.json"))
    assert abs(p["calibration"]["bytes_per_fetch_size_unit"] - 2048) < 1e-6
    assert abs(p["traffic_bytes_per_launch"]["gemv_experts_w13"] - 90e6) < 2
    assert "kernel" in open(out + "_kernel_trace.txt").read()


def test_repack_gpu_layout_planes_roundtrip(tmp_path):
    """tools/repack.py: every K-quant tensor of a checkpoint becomes byte planes that hold exactly the
    bytes of the reference blocks (a permutation), the header still parses (dsk_dseek_read_config), everything else
    is copied verbatim."""
    import json
    import struct
    import numpy as np
    import dsk
    from tools import repack, synth
    for quant, bsz in (("q2_k", 84), ("q3_k", 110)):
        c = synth.preset("tiny_v3", quant, True)
        T = synth.synth_model(c, seed=3)
        src, dst = str(tmp_path / f"src_{quant}"), str(tmp_path / f"dst_{quant}")
        synth.write_dseek(src, c, T, shards=2, tokenizer=True)
        repack.repack(src, dst)
        cfg_a, cfg_b = dsk.read_dseek_config(src)[0], dsk.read_dseek_config(dst)[0]
        assert bytes(cfg_a) == bytes(cfg_b)
        assert repack.read_shard(f"{dst}/shard_000.dseek")[0]["__metadata__"]["gpu_layout"] == "planes-v1"
        shards = []
        for fn in ("shard_000.dseek", "shard_001.dseek"):
            hdr, data0 = repack.read_shard(f"{dst}/{fn}")
            shards.append((hdr, data0, np.fromfile(f"{dst}/{fn}", np.uint8)))
        name = "model.layers.1.mlp.w1.weight"  # an expert stack

        def get(n):
            for hdr, data0, raw in shards:
                if n in hdr:
                    return raw[data0 + hdr[n]["data_offsets"][0]:data0 + hdr[n]["data_offsets"][1]]
            raise KeyError(n)
        blocks = T[name].data.reshape(-1, bsz)
        if quant == "q2_k":
            assert np.array_equal(get(name + ".qs").reshape(-1, 64), blocks[:, 16:80])
            assert np.array_equal(get(name + ".dm").reshape(-1, 4), blocks[:, 80:84])
            sc = get(name + ".sc").reshape(-1, 16)
            for jj in range(16):  # sub-block 8h + 2s + lh sits at quarter 2h + lh, slot s
                h, s, lh = jj >> 3, (jj >> 1) & 3, jj & 1
                assert np.array_equal(sc[:, (2 * h + lh) * 4 + s], blocks[:, jj])
        else:
            assert np.array_equal(get(name + ".hm").reshape(-1, 32), blocks[:, :32])
            assert np.array_equal(get(name + ".qs").reshape(-1, 64), blocks[:, 32:96])
            assert np.array_equal(get(name + ".sc").reshape(-1, 12), blocks[:, 96:108])
            assert np.array_equal(get(name + ".dm").reshape(-1, 2), blocks[:, 108:110])
        assert np.array_equal(get("model.layers.1.moegate.weight").view(np.float32), T["model.layers.1.moegate.weight"].data.reshape(-1))


def test_random_block_checkpoints_are_valid_for_the_oracle():
    """tools/synth.random_block_model (the full-width fixtures of tests/test_teacher_forced_gpu.py and bench.py's CPU leg):
    random Q2_K / Q3_K blocks with chosen d decode to finite, unit-scale logits through the CPU oracle, and tiling a
    tensor from a shorter random pattern keeps its shape and byte count."""
    import numpy as np
    from oracle import orc
    from tools import synth
    O_ = orc.Oracle()
    for quant, bsz in (("q2_k", 84), ("q3_k", 110)):
        c = synth.preset("tiny_v3", quant, False)
        T = synth.random_block_model(c, seed=3)
        Tt = synth.random_block_model(c, seed=3, tile_blocks=7)
        for name, t in T.items():
            assert Tt[name].data.shape == t.data.shape and Tt[name].quant == t.quant, name
        w = T["model.layers.1.mlp.experts.w1.weight"] if "model.layers.1.mlp.experts.w1.weight" in T else next(t for n, t in T.items() if n.endswith("w1.weight"))
        assert w.data.dtype == np.uint8 and w.data.shape[-1] % bsz == 0
        O = O_.model(c, T)
        lg = O.forward(5, 0)
        assert np.all(np.isfinite(lg)) and 0.05 < float(lg.std()) < 20.0, (quant, float(lg.std()))
        O.close()


def test_kept_experiment_patches_still_apply():
    """tools/patches/*.patch are measured-and-rejected kernel variants kept for reproduction (EXPERIMENTS.md): each must still
    apply to the sources it was cut from, or the A/B tables under profiles/ cannot be re-made"""
    import glob
    import shutil
    import pytest
    if not shutil.which("git"):
        pytest.skip("git not available")
    patches = sorted(glob.glob(os.path.join(ROOT, "tools", "patches", "*.patch")))
    assert patches
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(p), r.stderr[:300])
