"""Load rate of the direct-to-HBM `.dseek` loader (run on the GPU box).

    python tools/loadbench.py [n_experts] [tmpdir]

Writes a DeepSeek-V3-width Q2_K checkpoint (1 dense + 1 MoE block with `n_experts` routed experts of the true shape,
random valid blocks) and loads it twice with dsk_model_load_dseek (the second time from the page cache), then binds
the same tensors from host memory one by one (dsk_model_bind) for comparison.  Prints GB/s of tensor bytes.
"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk  # noqa: E402
from tools import synth  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    base = sys.argv[2] if len(sys.argv) > 2 else None
    c = synth.preset("v3", "q2_k", False, n_layers=2, first_k_dense_replace=1, n_routed_experts=E, n_group=8, topk_group=4,
                     max_seq_len=64)
    t0 = time.time()
    T = synth.random_block_model(c, seed=0, tile_blocks=1 << 20)
    d = tempfile.mkdtemp(prefix="dsk_loadbench_", dir=base)
    try:
        synth.write_dseek(d, c, T, shards=2)
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        print(f"checkpoint: {nbytes / 1e9:.2f} GB in {d} (built in {time.time() - t0:.1f} s)")
        ctx = dsk.Ctx(0)
        for label in ("first load", "second load (page cache)"):
            M = dsk.Model.from_dseek(ctx, d)
            s = M.load_stats
            print(f"{label:26s}: {s.file_bytes / 1e9:.2f} GB, {s.n_tensors} tensors in {s.seconds:.2f} s = {s.file_bytes / s.seconds / 1e9:.2f} GB/s "
                  f"(pread into pinned buffers {s.read_seconds:.2f} s = {s.file_bytes / max(s.read_seconds, 1e-9) / 1e9:.2f} GB/s)")
            lg = M.forward(5, 0)
            M.close()
        # the same checkpoint in the engine's plane layout (tools/repack.py): no staging copy, no repack kernels
        from tools import repack
        d2 = tempfile.mkdtemp(prefix="dsk_loadbench_planes_", dir=base)
        try:
            repack.repack(d, d2)
            for label in ("planes: first load", "planes: second load"):
                M = dsk.Model.from_dseek(ctx, d2)
                s = M.load_stats
                print(f"{label:26s}: {s.file_bytes / 1e9:.2f} GB, {s.n_tensors} tensors in {s.seconds:.2f} s = {s.file_bytes / s.seconds / 1e9:.2f} GB/s "
                      f"(pread into pinned buffers {s.read_seconds:.2f} s)")
                import numpy as np
                assert np.array_equal(lg, M.forward(5, 0))
                M.close()
        finally:
            shutil.rmtree(d2, ignore_errors=True)
        t1 = time.time()
        M = dsk.Model(ctx, c, T)
        dt = time.time() - t1
        print(f"{'bind walk from memory':26s}: {nbytes / 1e9:.2f} GB in {dt:.2f} s = {nbytes / dt / 1e9:.2f} GB/s")
        import numpy as np
        assert np.array_equal(lg, M.forward(5, 0))
        M.close()
        ctx.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
