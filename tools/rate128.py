"""The SURVEY 8d rate by itself: 128 decode steps at positions 72-199 of the full DeepSeek-V3 Q2_K model (what bench.py reports as
tok_s_128), repeated.   [DSK_LIB=...] python tools/rate128.py [mha|mla] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import numpy as np
import dsk
from tools import synth

mla = len(sys.argv) > 1 and sys.argv[1] == "mla"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = synth.preset("v3", "q2_k", mla, max_seq_len=1100)
ctx = dsk.Ctx(0)
M = dsk.Model(ctx, c, None, synth_seed=0)
toks = np.random.default_rng(0).integers(0, c.vocab_size, 256)
out = []
for r in range(reps):
    for pos in range(72):
        M.forward_nocopy(int(toks[pos]), pos)
    t0 = time.perf_counter()
    for pos in range(72, 200):
        M.forward_nocopy(int(toks[pos]), pos)
    out.append(round(128 / (time.perf_counter() - t0), 2))
print(os.path.basename(os.environ.get("DSK_LIB", "libdsk_hip.so")), "mla" if mla else "mha", "tok/s over positions 72-199:", out)
M.close(); ctx.close()
