"""One GEMV shape of a DeepSeek-V3 Q2_K token launched alone 20 times (include/dsk.h dsk_bench_gemv) - the workload of the
SQ counter passes of tools/pmc_gemv.sh.   python tools/pmc_gemv.py lm_head|wo|dense_w13"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk
ctx = dsk.Ctx(0)
which = sys.argv[1] if len(sys.argv) > 1 else "lm_head"
shapes = {"lm_head": (129280, 7168, 1, 0, 2), "wo": (7168, 16384, 1, 0, 0), "dense_w13": (18432, 7168, 1, 1, 2)}
rows, n, nt, kind, act = shapes[which]
us, nb = ctx.bench_gemv(3, rows, n, nt, kind, act, 0, 0, 0, 0, 20)
print(which, round(us, 2), "us", round(nb / us / 1e3, 1), "GB/s")
