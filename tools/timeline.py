"""Where each launch of a MoE block spends its time: per-workgroup wall-clock stamps (100 MHz) of the LAST launch of each kind
in a token.   python tools/timeline.py [--layers 8] [--attn mla] [--pos 6]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk
from tools import synth

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=8); ap.add_argument("--attn", default="mha"); ap.add_argument("--pos", type=int, default=6); ap.add_argument("--kv", type=int, default=0); ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
c = synth.preset("v3", "q2_k", a.attn == "mla", n_layers=a.layers, max_seq_len=max(64, a.pos + 8, a.kv + 16))
ctx = dsk.Ctx(0); M = dsk.Model(ctx, c, None, synth_seed=0, options=dict({"timeline": 1}, **{kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt}))
for pos in range(a.pos):
    M.forward(17 + pos, pos)
for i in range(4 if a.kv else 0):  # a long context: the cache rows below hold zeros, the kernels stream them all the same
    M.forward(17 + i, a.kv + i)
f = dsk.lib().dsk_model_get_timeline; f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
GEMV = ["entry", "rows[0] done", "", "exit", "stage: loaded", "stage: quantised", "stage begin", "staged", ]
KINDS = {0: ("first-stage projections (gemv)", None), 1: ("per-head attention", (["entry", "q staged", "attention (or merge) done", "wv_b rows done", "exit (Q8 of the output)"] if a.attn == "mla" else
                                    ["entry", "latents staged", "head rows done", "rope + cache row", "attention done", "exit (Q8 of the output)"])),
         2: ("wo (gemv)", None), 3: ("shared expert w1/w3 (rider gemv)", None), 5: ("router", ["entry", "norm scale", "rows done", "arrived", "gate done (last only)"]),
         6: ("MLA long-context scores / values (mla_flash_kernel)", ["entry", "Q tile staged", "first 32 cache rows staged", "scores done", "softmax done", "values done", "all blocks done", "exit (partials written)"]),
         4: ("routed experts", ["entry", "staged x", "phase A done", "hand-off passed", "hidden staged", "rows done", "exit"])}
for kind in (0, 6, 1, 2, 5, 3, 4):
    name, stamps = KINDS[kind]
    buf = np.zeros((1024, 8), np.uint64)
    dsk.check(f(M.h, kind, buf.ctypes.data, 1024))
    used = buf[:, 0] > 0
    if not used.any():
        continue
    t = buf[used].astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    print(f"--- {name}: {int(used.sum())} workgroups; us after the first workgroup's entry: min / median / max")
    if stamps is None:  # gemv_body: 0 entry, 6 stage begin, 4, 5, 7 staged, 1 after the barrier, 2 first row group, 3 exit
        order = [(0, "entry"), (6, "stage begin"), (4, "stage: vector loaded"), (5, "stage: quantised"), (1, "staged (barrier)"), (2, "first row group done"), (3, "exit")]
    else:
        order = list(enumerate(stamps))
    for i, nm in order:
        v = t[:, i]
        v = v[v > 0] - t0
        if v.size:
            print(f"    {nm:26s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}   ({v.size})")
