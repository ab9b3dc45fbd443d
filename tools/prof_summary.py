#!/usr/bin/env python
"""Turn rocprofv3 output databases into the small summaries committed under profiles/.

    # on the GPU box (see profiles/README.md for the exact commands of each round)
    rocprofv3 --kernel-trace --stats -d gpurun_out/trace -- python bench.py --steps 16 --warmup 3 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc -- python bench.py --layers 8 --steps 4 --warmup 1 --no-graph --no-cpu-baseline
    python tools/prof_summary.py --trace gpurun_out/trace --pmc gpurun_out/pmc --out profiles/r01

Writes <out>_kernel_trace.txt / .json (per-kernel count, avg / min / max duration, share) and
<out>_pmc.json (FETCH_SIZE per dispatch, calibrated on the engine's own streaming-read kernel whose
byte count is known: MI355X_MICROARCH.md says FETCH_SIZE under-reports wide coalesced reads by 2x on
gfx950 and is uncalibrated for other widths, so the factor is measured, not assumed).
"""
import argparse
import glob
import json
import os
import sqlite3

# bench.py kernel class -> substring of the kernel symbol that implements it in the V3 Q2_K bench
CLASS_KERNEL = {
    "moe_ffn": "moe_ffn_",  # routed experts w1/w3 + W2 + the shared expert's W2 + combine: one launch (moe_ffn_kernel; round 4: moe_ffn_tile_kernel)
    # (two-launch form, DSK_NO_FUSE_MOE=1): one kernel, two populations of dispatches (8 routed experts / dense w1/w3)
    "gemv_experts_w13": ("gemv_kernel<3, 1, 4, true, 16>", "lo"),
    "gemv_dense_w13": ("gemv_kernel<3, 1, 4, true, 16>", "hi"),
    "gemv_experts_w2": "gemv_kernel<3, 2, 4, false, 4>",
    "gemv_wo": "gemv_kernel<3, 1, 4, false, 16>",
    # wq_a || wkv_a (5 MB): 64 lanes per row, 2 column steps (gemv_kernel<3, 1, 8, false, 16> = dense w2 and lm_head)
    "gemv_qkv_a": "gemv_kernel<3, 1, 2, false, 16>",
    "router_gate": "router_shared_",  # router + the shared expert's w1/w3 (router_shared_kernel / router_shared_tile_kernel; router_gate_kernel when not fused)
    "attn_mha": "head_attn_kernel",
    "attn_mla": "mla_head_kernel",
}
READ_BW_BYTES = 4 << 30  # bench.py: ctx.measure_read_bw(4 << 30, 5)


def dbs(d):
    return sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))


def trace_summary(d):
    rows = {}
    for db in dbs(d):
        con = sqlite3.connect(db)
        for name, dur in con.execute("select name, end - start from kernels"):
            r = rows.setdefault(name, [0, 0, 1 << 62, 0])
            r[0] += 1
            r[1] += dur
            r[2] = min(r[2], dur)
            r[3] = max(r[3], dur)
    tot = sum(r[1] for r in rows.values()) or 1
    out = [dict(kernel=k, calls=r[0], avg_us=round(r[1] / r[0] / 1e3, 2), min_us=round(r[2] / 1e3, 2), max_us=round(r[3] / 1e3, 2),
                total_ms=round(r[1] / 1e6, 3), pct=round(100.0 * r[1] / tot, 2)) for k, r in rows.items()]
    out.sort(key=lambda x: -x["total_ms"])
    return out


def pmc_summary(d):
    agg = {}
    for db in dbs(d):
        con = sqlite3.connect(db)
        for name, counter, value, disp in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            a = agg.setdefault((name, counter), {})
            a[disp] = a.get(disp, 0.0) + value  # a counter is reported per XCD / instance: sum them
    out = {}
    for (name, counter), per in agg.items():
        v = list(per.values())
        mid = (min(v) + max(v)) / 2
        lo, hi = [x for x in v if x <= mid], [x for x in v if x > mid]
        out.setdefault(name, {})[counter] = dict(dispatches=len(v), avg=sum(v) / len(v), min=min(v), max=max(v),
                                                 lo_avg=sum(lo) / len(lo), hi_avg=sum(hi) / len(hi) if hi else sum(lo) / len(lo))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    if a.trace:
        t = trace_summary(a.trace)
        json.dump(dict(note=a.note, kernels=t), open(a.out + "_kernel_trace.json", "w"), indent=1)
        with open(a.out + "_kernel_trace.txt", "w") as f:
            f.write(f"# rocprofv3 --kernel-trace summary. {a.note}\n")
            f.write(f"{'kernel':92s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>10s} {'pct':>6s}\n")
            for r in t:
                f.write(f"{r['kernel'][:92]:92s} {r['calls']:7d} {r['avg_us']:9.2f} {r['min_us']:9.2f} {r['max_us']:9.2f} {r['total_ms']:10.3f} {r['pct']:6.2f}\n")
        print(open(a.out + "_kernel_trace.txt").read()[:3000])
    if a.pmc:
        p = pmc_summary(a.pmc)
        cal = None
        for name, cs in p.items():
            if "read_bw_kernel" in name and "FETCH_SIZE" in cs:
                cal = READ_BW_BYTES / cs["FETCH_SIZE"]["avg"]
        traffic = {}
        if cal:
            for cls, sub in CLASS_KERNEL.items():
                sub, which = sub if isinstance(sub, tuple) else (sub, "avg")
                for name, cs in p.items():
                    if sub in name and "FETCH_SIZE" in cs:
                        f = cs["FETCH_SIZE"]
                        spread = f["max"] > 1.05 * f["min"]  # two populations only if the counter really splits
                        traffic[cls] = round(f[{"lo": "lo_avg", "hi": "hi_avg", "min": "min"}.get(which, "avg") if spread else "avg"] * cal)
            if "moe_ffn" in traffic:  # fused build: gemv_kernel<3,1,4,true,16> is the dense w1/w3 only
                traffic.pop("gemv_experts_w13", None)
                traffic.pop("gemv_experts_w2", None)
        import hashlib
        h = hashlib.sha256()
        cs_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepseek.cpp_amd", "csrc")
        for fn in sorted(os.listdir(cs_dir)):
            if fn.endswith((".hip", ".h", ".cpp")) or fn == "Makefile":
                h.update(fn.encode())
                h.update(open(os.path.join(cs_dir, fn), "rb").read())
        json.dump(dict(note=a.note, csrc_sha=h.hexdigest()[:16], calibration=dict(kernel="read_bw_kernel (16 B/lane streaming read of 4 GiB)", bytes_per_fetch_size_unit=cal),
                       traffic_bytes_per_launch=traffic, counters=p), open(a.out + "_pmc.json", "w"), indent=1)
        print("calibration bytes per FETCH_SIZE unit:", cal)
        print("traffic per launch:", traffic)


if __name__ == "__main__":
    main()
