"""BASELINE.json configs[0]: DeepSeek-V2-Lite FP16, the reference's own CPU OpenMP path, `-n 128` completion -- plumbing, no GPU.

Runs the UNMODIFIED reference binary (oracle/_ref/main, built from /root/reference by oracle/Makefile) the way
BASELINE.md section 4 prescribes: `main DIR -m c -n 128 -t 0 -i <prompt>`, OMP_NUM_THREADS swept, and prints its
"throughput" line as JSON.  A full V2-Lite FP16 checkpoint is 31 GB; the sample is a reduced-depth one (1 dense + 3 MoE
blocks, full width, all 64 experts, full vocabulary) and the per-token time is extrapolated to 27 blocks from the
per-block share, labelled as such.   python tools/cpu_c1.py > profiles/r02_cpu_c1.json"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth

MAIN = os.path.join(ROOT, "oracle", "_ref", "main")


def run(d, threads):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    r = subprocess.run([MAIN, d, "-m", "c", "-n", "128", "-t", "0", "-i", "the quick brown fox jumps over the lazy dog and keeps on running"],
                       capture_output=True, env=env, timeout=900)
    out = r.stdout.decode("latin-1")
    m = re.search(r"Generation stats:\s+(\d+) tokens\s+throughput: ([0-9.eE+-]+)tok/s\s+latency: ([0-9.eE+-]+)s/tok", out)
    b = re.search(r"bandwidth: ([0-9.eE+-]+)GB/s", out[m.start():] if m else "")
    return (int(m.group(1)), float(m.group(2)), float(m.group(3)), float(b.group(1)) if b else 0.0) if m else None


def main():
    if not os.path.exists(MAIN):
        print(json.dumps({"error": "oracle/_ref/main not built"}))
        return
    res = {}
    for layers in (2, 6):  # two depths: the difference is the cost of the extra MoE blocks
        c = synth.preset("v2lite", "fp16", False, n_layers=layers, first_k_dense_replace=1, max_seq_len=512)
        T = synth.random_block_model(c, seed=1)
        d = tempfile.mkdtemp(prefix="dsk_c1_")
        try:
            synth.write_dseek(d, c, T, shards=1, tokenizer=True)
            del T
            ncpu = os.cpu_count() or 1
            best = None
            for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu}):
                r = run(d, th)
                if r and (best is None or r[1] > best[1]):
                    best = (th,) + r
            res[layers] = best
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {"config": "DeepSeek-V2-Lite FP16 (synthetic weights), reference CPU OpenMP path, main -m c -n 128 -t 0", "cpus": os.cpu_count(),
           "runs": {str(k): dict(threads=v[0], tokens=v[1], tok_s=v[2], s_per_tok=v[3], reference_reported_gbps=v[4]) for k, v in res.items() if v}}
    if res.get(2) and res.get(6):
        per_moe = max(0.0, (res[6][3] - res[2][3]) / 4.0)
        rest = res[2][3] - per_moe  # 1 dense block + embedding + classifier (the one MoE block subtracted)
        full = rest + 26 * per_moe  # V2-Lite: 1 dense + 26 MoE blocks
        out["extrapolated_27_blocks"] = {"s_per_tok": round(full, 5), "tok_s": round(1.0 / full, 3),
                                         "note": "linear in the MoE block count from the 2- and 6-block runs; on a host with a large L3 the reduced-depth "
                                                 "checkpoint's active experts stay cached, so this is an UPPER bound of the full model's rate"}
        # second estimate: the bandwidth the reference itself reports (src/main.cpp:351, Model::active_bytes) on the deeper run,
        # against the 4.91 GB a full V2-Lite FP16 token touches
        if res[6][4] > 0:
            out["by_reported_bandwidth"] = {"tok_s": round(res[6][4] / 4.91, 3), "note": "reference-reported GB/s of the 6-block run / 4.91 GB per full-depth token"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
