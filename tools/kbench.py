"""GEMV micro-benchmarks on the DeepSeek-V3 Q2_K decode shapes (run on the GPU box).

    python tools/kbench.py            # planner's choice for every V3 shape (+ the same with a ready Q8 input)
    python tools/kbench.py sweep [i]  # sweep (lpr, R, U, grid) on shape i (default: the big ones)

Prints achieved GB/s = weight bytes / kernel time (HIP events over back-to-back launches on
rotating weight sets, so HBM rather than the Infinity Cache is measured).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk  # noqa: E402

Q2K = 3
# name, rows, n, n_tasks, kind (0 plain, 1 GLU, 2 accumulate), act_mode (0 q8, 1 f32, 2 f32+norm)
V3 = [
    ("lm_head 129280x7168", 129280, 7168, 1, 0, 2),
    ("experts_w13 9x(2048x7168)x2", 2048, 7168, 9, 1, 2),
    ("experts_w2+combine 9x(7168x2048)", 7168, 2048, 9, 3, 1),
    ("wo 7168x16384", 7168, 16384, 1, 0, 0),
    ("dense_w13 (18432x7168)x2", 18432, 7168, 1, 1, 2),
    ("dense_w2 7168x18432", 7168, 18432, 1, 0, 1),
    ("wq_b 24576x1536", 24576, 1536, 1, 0, 2),
    ("wkv_b 32768x512", 32768, 512, 1, 0, 2),
    ("wq_a 1536x7168", 1536, 7168, 1, 0, 2),
    ("wkv_a 576x7168", 576, 7168, 1, 0, 2),
    ("shared_w13 (2048x7168)x2", 2048, 7168, 1, 1, 2),
    ("mla wq_rope_b||wc 73728x1536", 73728, 1536, 1, 0, 2),
    ("mla wv_b (block-diagonal stand-in) 16384x512", 16384, 512, 1, 0, 0),
]


def main():
    ctx = dsk.Ctx(0)
    print("read bw GB/s", round(ctx.measure_read_bw(4 << 30, 3), 1))
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        which = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
        for i in which:
            name, rows, n, nt, kind, act = V3[i]
            print("==", name)
            res = []
            for lpr in (0, 8, 16, 32, 64):
                for R, U in ((1, 8), (2, 4), (1, 4), (2, 2), (1, 2), (2, 1), (1, 1)):
                    if kind == 1 and R > 2:
                        continue
                    for wgs in (512, 1024, 2048):
                        try:
                            us, nb = ctx.bench_gemv(Q2K, rows, n, nt, kind, act, lpr, R, U, wgs, 20)
                        except dsk.DskError:
                            continue
                        res.append((nb / us / 1e3, lpr, R, U, wgs, us))
            res.sort(reverse=True)
            for g, lpr, R, U, wgs, us in res[:12]:
                print(f"  lpr={lpr:2d} R={R} U={U} wgs={wgs:4d}: {us:8.2f} us {g:7.1f} GB/s")
            print("  worst", res[-1])
    else:
        for name, rows, n, nt, kind, act in V3:
            us, nb = ctx.bench_gemv(Q2K, rows, n, nt, kind, act, 0, 0, 0, 0, 50)
            us0, _ = ctx.bench_gemv(Q2K, rows, n, nt, kind, 0, 0, 0, 0, 0, 50) if kind < 2 else (float("nan"), 0)
            print(f"{name:36s} {us:8.2f} us  {nb/1e6:8.2f} MB  {nb/us/1e3:7.1f} GB/s   | ready-Q8 input: {us0:8.2f} us {nb/us0/1e3:7.1f} GB/s")
    ctx.close()


if __name__ == "__main__":
    main()
