"""Localise a difference between dsk_hydrate's batched path and the per-token loop, stage by stage, in ONE call.

  python tools/hydrate_debug.py [--preset tiny_v3|v3] [--P 5] [--pos0 0]

For every block l: the batched chunk is re-run with option hydrate_tap_layer = l and the block's tapped intermediates are
compared with the stages dsk_model_run_block leaves when the SAME block runs on the loop's own residual stream (x of the
previous block, per token).  Prints the first stage that differs per (block, token) and how far off it is.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepseek.cpp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from tools import synth  # noqa: E402


def main():
    import dsk
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="tiny_v3")
    ap.add_argument("--P", type=int, default=5)
    ap.add_argument("--pos0", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0)
    a = ap.parse_args()
    ctx = dsk.Ctx(0)
    if a.preset == "v3":
        c = synth.preset("v3", "q2_k", False, n_layers=a.layers or 2, first_k_dense_replace=1, max_seq_len=192)
        T, seed = None, 11
    else:
        c = synth.preset("tiny_v3", "q2_k", False)
        T, seed = synth.synth_model(c, seed=41), None
    rng = np.random.default_rng(a.P * 31 + a.pos0)
    tokens = [int(t) for t in rng.integers(0, c.vocab_size, a.P)]
    pre = [(7 * i + 3) % c.vocab_size for i in range(a.pos0)]
    P, K = a.P, max(1, c.n_active_routed)
    H, hd, vd, nv = c.n_heads, c.qk_nope_head_dim + c.qk_rope_head_dim, c.v_head_dim, c.qk_nope_head_dim + c.v_head_dim
    A = dsk.Model(ctx, c, T, synth_seed=seed, options={"q2k_tiles": 2})
    A.set_trace(True)
    for i, t in enumerate(pre):
        A.forward(t, i, dsk.MODE_HYDRATE_KV_CACHE)
    xs = []
    for i, t in enumerate(tokens):
        A.forward(t, a.pos0 + i, dsk.MODE_HYDRATE_KV_CACHE)
        xs.append(np.stack([A.trace_x(l) for l in range(c.n_layers)]))
    nbad = 0
    for l in range(c.n_layers):
        B = dsk.Model(ctx, c, T, synth_seed=seed, options={"q2k_tiles": 2, "hydrate_tap_layer": l, "hydrate_chunk": max(P, 4)})
        B.set_trace(True)
        for i, t in enumerate(pre):  # the context comes from the loop on B itself (the same decode path as A)
            B.forward(t, i, dsk.MODE_HYDRATE_KV_CACHE)
        B.hydrate(tokens, a.pos0, dsk.MODE_HYDRATE_KV_CACHE)
        moe = l >= c.first_k_dense_replace
        bufs = {"q_a": c.q_lora_rank, "kv_a": c.kv_lora_rank + c.qk_rope_head_dim, "att_out": H * vd}
        if moe:
            bufs.update({"route_e": K, "route_w": K, "eout": K * c.dim})
            if c.n_shared_experts:
                bufs["eout_sh"] = c.dim
        got = {k: B.hydrate_buffer(k, 0, P, w, np.int32 if k == "route_e" else np.float32) for k, w in bufs.items()}
        got["x"] = np.stack([B.hydrate_trace_x(l, i) for i in range(P)])
        for i in range(P):
            if l > 0:  # (block 0's input is the embedding row, which has no accessor: only its output is compared)
                # the loop's own stages: block l on the loop's residual stream at this position (the cache rows of earlier positions
                # are already there: A decoded the whole prompt; the row of this position is rewritten with the same values)
                A.run_block(l, xs[i][l - 1], a.pos0 + i)
                want = {"q_a": A.stage("q_a", c.q_lora_rank), "kv_a": A.stage("kv_a", c.kv_lora_rank + c.qk_rope_head_dim),
                        "att_out": A.stage("att_out", H * vd)}
                if moe:
                    want["route_e"] = A.stage("route_e", K, np.int32)
                    want["route_w"] = A.stage("route_w", K)
                    eo = A.stage("eout", (K + (1 if c.n_shared_experts else 0)) * c.dim).reshape(-1, c.dim)
                    want["eout"] = eo[:K]
                    if c.n_shared_experts:
                        want["eout_sh"] = eo[K]
                for k in ("q_a", "kv_a", "att_out", "route_e", "route_w", "eout", "eout_sh"):
                    if k not in want:
                        continue
                    g = got[k][i]
                    w = want[k]
                    if not np.array_equal(np.asarray(g).reshape(-1), np.asarray(w).reshape(-1)):
                        gd, wd = np.asarray(g, np.float64).reshape(-1), np.asarray(w, np.float64).reshape(-1)
                        print(f"block {l} token {i}: stage {k} differs: max|d| {np.abs(gd - wd).max():.3e} of {np.abs(wd).max():.3e}, "
                              f"{int((gd != wd).sum())} / {gd.size} elements; first at {int(np.nonzero(gd != wd)[0][0])}")
                        nbad += 1
                        break
            xo = got["x"][i]
            if not np.array_equal(xo, xs[i][l]):
                d = np.abs(xo - xs[i][l])
                print(f"block {l} token {i}: x after the block differs: max|d| {d.max():.3e} of {np.abs(xs[i][l]).max():.3e}, {int((d > 0).sum())} / {d.size}")
                nbad += 1
        B.close()
    print("hydrate_debug:", "IDENTICAL" if nbad == 0 else f"{nbad} differences")
    A.close()


if __name__ == "__main__":
    main()
