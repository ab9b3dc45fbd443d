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
