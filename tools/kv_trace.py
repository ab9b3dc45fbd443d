"""A few decode steps of the full DeepSeek-V3 Q2_K model at a given kv_len, for `rocprofv3 --kernel-trace` (SURVEY 8d's
kv_len sweep: profiles/r02_kv4096_kernel_trace.txt).   python tools/kv_trace.py 4096 [mha|mla]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk
from tools import synth

kv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mla = len(sys.argv) > 2 and sys.argv[2] == "mla"
c = synth.preset("v3", "q2_k", mla, max_seq_len=4200)
ctx = dsk.Ctx(0)
M = dsk.Model(ctx, c, None, synth_seed=0)
for i in range(8):
    M.forward_nocopy(100 + i, kv - 1 + i)
M.close()
ctx.close()
