"""One line per library build for A/B measurements of the decode step (no torch: starts in seconds).
    DSK_LIB=deepseek.cpp_amd/_ab/libdsk_X.so python tools/moe_ab.py [--layers 61] [--steps 32] [--opt KEY=VALUE]
Prints: ms per token (hipGraph replays, logits D2H included, wall clock around the steps), the in-situ average of the launch
classes of a MoE block (the kernels' own dispatch timestamps: dsk_profile_forward) and a hash of the logits of every step, which
must be identical between builds that only move work around."""
import argparse, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import numpy as np
import dsk
from tools import synth

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=61)
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--warmup", type=int, default=6)
ap.add_argument("--attn", default="mha")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
c = synth.preset("v3", "q2_k", a.attn == "mla", n_layers=a.layers, max_seq_len=256)
ctx = dsk.Ctx(0)
M = dsk.Model(ctx, c, None, synth_seed=0, options={kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt})
rng = np.random.default_rng(0)
toks = rng.integers(0, c.vocab_size, a.warmup + a.steps + 8)
h = hashlib.sha256()
pos = 0
for _ in range(a.warmup):
    h.update(M.forward_nocopy(int(toks[pos]), pos).tobytes()); pos += 1
t0 = time.perf_counter()
for _ in range(a.steps):
    M.forward_nocopy(int(toks[pos]), pos); pos += 1
ms = (time.perf_counter() - t0) * 1e3 / a.steps
h.update(M.forward_nocopy(int(toks[pos]), pos).tobytes()); pos += 1
acc = {}
for _ in range(4):
    for k in M.profile_forward(int(toks[pos]), pos):
        e = acc.setdefault(k["name"], [0, 0.0]); e[0] += k["launches"]; e[1] += k["total_ms"]
    pos += 1
us = {n: round(v[1] * 1e3 / max(1, v[0]), 2) for n, v in acc.items()}
keep = ["moe_ffn", "router_gate", "gemv_wo", "attn_mha", "attn_mla", "gemv_qkv_a", "gemv_qkv_b", "gemv_lm_head"]
label = os.path.basename(os.environ.get('DSK_LIB', 'libdsk_hip.so')) + (' ' + ' '.join(a.opt) if a.opt else '')
print(f"{label:36s} {ms:7.4f} ms/token  {1e3 / ms:7.2f} tok/s  "
      + "  ".join(f"{n} {us[n]}" for n in keep if n in us)
      + f"  fused {M.info('fused_moe_layers')} fallbacks {M.info('handoff_fallbacks')}  logits {h.hexdigest()[:12]}", flush=True)
M.close()
