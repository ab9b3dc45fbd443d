"""Where the fused routed-expert launch (kernels_moe.hip) spends its time: per-workgroup wall-clock stamps of the last
MoE block of a token.   python tools/moe_timeline.py [--layers 8]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk
from tools import synth

ap = argparse.ArgumentParser(); ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--opt", action="append", default=[], help="KEY=VALUE model option, repeatable (e.g. q2k_tiles=0)")
a = ap.parse_args()
c = synth.preset("v3", "q2_k", False, n_layers=a.layers, max_seq_len=64)
ctx = dsk.Ctx(0); opts = {"timeline": 1}
opts.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt})
M = dsk.Model(ctx, c, None, synth_seed=0, options=opts)
for pos in range(6):
    M.forward(17 + pos, pos)
n = 256
buf = np.zeros((n, 8), np.uint64)
f = dsk.lib().dsk_model_get_moe_timeline; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
dsk.check(f(M.h, buf.ctypes.data, n))
t = buf.astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
names = ["entry", "staged x", "phase A done", "hand-off passed", "hidden staged", "rows done", "exit"]
print("stamp                 min      median   max   (us after the first workgroup's entry)")
for i, nm in enumerate(names):
    v = t[:, i] - t0
    print(f"{nm:18s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
if buf[:, 7].any():  # diagnostics builds (-DMOE_TL7=1: end of the workgroup's hand-over duty, bit 0 = it quantised a block; 2: polls passed)
    raw = buf[:, 7]
    v = (raw & ~np.uint64(1)).astype(np.float64) / 100.0 - t0
    last = (raw & np.uint64(1)).astype(bool)
    print(f"{'stamp 7':18s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
    if last.any() and not last.all():
        print(f"{'  bit 0 set':18s} {v[last].min():8.2f} {np.median(v[last]):8.2f} {v[last].max():8.2f}   ({int(last.sum())} workgroups)")
        print(f"{'  bit 0 clear':18s} {v[~last].min():8.2f} {np.median(v[~last]):8.2f} {v[~last].max():8.2f}")
        dd = v - (t[:, 2] - t0)
        print(f"{'  stamp 7 - phase A done':18s} bit 0 set: median {np.median(dd[last]):.2f} max {dd[last].max():.2f}; clear: median {np.median(dd[~last]):.2f} max {dd[~last].max():.2f}")
d = np.diff(t[:, :7], axis=1)
print("segment medians (us):", {names[i + 1]: round(float(np.median(d[:, i])), 2) for i in range(6)})

# who is slow?  block b runs on XCD b % 8 (MI355X_MICROARCH.md); phase-A unit b belongs to slot b // 32
a_done = t[:, 2] - t0
print("phase A done by XCD  (median us):", [round(float(np.median(a_done[np.arange(n) % 8 == x])), 1) for x in range(8)])
print("phase A done by slot (median us):", [round(float(np.median(a_done[np.arange(n) // 32 == k])), 1) for k in range(8)])
print("phase A done by slot (max us):   ", [round(float(np.max(a_done[np.arange(n) // 32 == k])), 1) for k in range(8)])
order = np.argsort(a_done)
print("slowest 12 workgroups:", [(int(b), round(float(a_done[b]), 1)) for b in order[-12:]])
print("fastest 8 workgroups: ", [(int(b), round(float(a_done[b]), 1)) for b in order[:8]])
ent = t[:, 0] - t0
print("entry by XCD (median us):", [round(float(np.median(ent[np.arange(n) % 8 == x])), 2) for x in range(8)])
