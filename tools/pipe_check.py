"""moe_pipe=1 (kernels_moe_pipe.hip) against the default fused expert launch: logits, slot outputs and routing bit for bit on one dense
+ one MoE block at DeepSeek-V3 width with 256 experts (MHA and MLA), a few positions.   python tools/pipe_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import numpy as np
import dsk
from tools import synth

ctx = dsk.Ctx(0)
bad = 0
PIPE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for mla in (False, True):
    c = synth.preset("v3", "q2_k", mla, n_layers=3, first_k_dense_replace=1, max_seq_len=64)
    A = dsk.Model(ctx, c, None, synth_seed=5)
    P = dsk.Model(ctx, c, None, synth_seed=5, options={"moe_pipe": PIPE})
    for pos, t in enumerate([3, 77, 1500, 9, 100000, 5]):
        la, lp = A.forward(t, pos), P.forward(t, pos)
        ok = np.array_equal(la, lp) and np.array_equal(A.slot_outputs(), P.slot_outputs()) and np.array_equal(A.routing()[0], P.routing()[0])
        bad += not ok
        print(f"{'mla' if mla else 'mha'} pos {pos}: {'identical' if ok else 'DIFFERENT'}  max|d| {np.abs(la - lp).max():.3e} of {np.abs(la).max():.3e}; fallbacks {P.info('handoff_fallbacks')}", flush=True)
    A.close(); P.close()
print("pipe_check:", "IDENTICAL" if bad == 0 else f"{bad} differences")
