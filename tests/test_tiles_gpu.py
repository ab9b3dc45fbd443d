"""Q2_K weights in the tiled layout (csrc/tile_device.h): the row products of ggml_vec_dot_q2_K_q8_K (src/quant.cpp:666-783)
with the sub-block dots on v_mfma_i32_16x16x64_i8.

The three layout levels of option `q2k_tiles` (0 planes + dot4, 1 the experts' matrices tiled, 2 every converted role) run the
SAME model: integer sub-block sums are exact in both forms, only the association of the f32 block sums differs, so each
projection agrees to float precision with the oracle's integer GEMV on the codes the device staged (the teacher-forced audit
does that for the default level); here the levels are compared with each other end to end, op level against the oracle, and
the tiled fused expert launch against the tiled two-launch form bit for bit (the association of tile_device.h does not depend
on the launch geometry).
"""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu


def _logits(ctx, c, T, level, tokens, **opts):
    import dsk
    o = {"q2k_tiles": level}
    o.update(opts)
    M = dsk.Model(ctx, c, T, options=o)
    out, routes = [], []
    for pos, t in enumerate(tokens):
        out.append(M.forward(int(t), pos).copy())
        routes.append(M.routing()[0].copy())
    fused = M.info("fused_moe_layers")
    M.close()
    return np.stack(out), routes, fused


@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
def test_tile_levels_agree(ctx, mla):
    """levels 0 / 1 / 2 on one tiny DeepSeek-V3 Q2_K model: the first token (no int8 rounding has compounded yet) agrees to float
    precision, the sequence within the W2A8 guard of tests/test_teacher_forced_gpu.py (a last-bit difference in front of a
    quantisation point can flip a rounding that sits on a tie); the routing of the first token is identical."""
    c = synth.preset("tiny_v3", "q2_k", mla)
    T = synth.synth_model(c, seed=23)
    tokens = [5, 9, 700, 3, 44, 1000]
    ref, r0, fused0 = _logits(ctx, c, T, 0, tokens)
    scale = np.abs(ref).max()
    for level in (1, 2):
        got, r, fused = _logits(ctx, c, T, level, tokens)
        assert fused == fused0, "the layout must not change which launches fuse"
        assert np.array_equal(r[0], r0[0])
        assert np.abs(got[0] - ref[0]).max() <= 3e-4 * scale, (level, np.abs(got[0] - ref[0]).max() / scale)
        assert np.abs(got - ref).max() <= 5e-2 * scale, (level, np.abs(got - ref).max() / scale)


def test_tiled_fused_equals_tiled_two_launch_and_unfused_rider(ctx):
    """level 2 (every launch tiled): fused expert launch == two-launch form == no rider, bit for bit, eager and graph"""
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=29)
    tokens = [1, 2, 3, 900, 17]
    a, _, fa = _logits(ctx, c, T, 2, tokens)
    b, _, fb = _logits(ctx, c, T, 2, tokens, fuse_moe=0)
    d, _, _ = _logits(ctx, c, T, 2, tokens, fuse_moe=0, fuse_shared=0)
    assert fb == 0  # (a model this small may not fuse at all: no rider in its router launch; the full-width case is below)
    assert np.array_equal(a, b)
    assert np.array_equal(a, d)


@pytest.mark.parametrize("rows,n", [(16, 256), (37, 512), (64, 2048), (130, 2304), (48, 7168), (16, 11008)])
def test_tiled_gemv_matches_oracle(ctx, oracle, rows, n):
    """dsk_gemv on Q2_K runs the tiled kernel: rows padded to 16, rows of <= 8 blocks (one item per block) and longer ones
    (4-block items, a ragged last item at 43 blocks), against the oracle's ggml_vec_dot_q2_K_q8_K"""
    rng = np.random.default_rng(rows * 131 + n)
    w = synth.encode_q2k((rng.standard_normal((rows, n)) / np.sqrt(n)).astype(np.float32))
    x = rng.standard_normal(n).astype(np.float32)
    got = ctx.gemv(3, w, rows, n, x)
    want = oracle.gemv(3, w, rows, n, x)
    assert np.abs(got - want).max() <= 2e-5 * max(1e-6, np.abs(want).max())


@pytest.mark.parametrize("rows,n", [(16, 256), (37, 512), (64, 2048), (130, 2304), (48, 7168), (16, 11008)])
def test_tiled_gemv_is_bit_exact_against_the_restated_association(ctx, oracle, rows, n):
    """the same launches against orc_gemv_q2k_tiles - the oracle's restatement of tile_device.h's f32 association on the
    integers of ggml_vec_dot_q2_K_q8_K - fed with the Q8_K codes the DEVICE makes of x: every bit must agree (the integer
    sub-block sums on the matrix pipe are exact, the float steps are the same fused multiply-adds in the same order)"""
    rng = np.random.default_rng(rows * 977 + n)
    w = synth.encode_q2k((rng.standard_normal((rows, n)) / np.sqrt(n)).astype(np.float32))
    x = (rng.standard_normal(n) * rng.uniform(0.05, 20)).astype(np.float32)
    qs, d, _ = ctx.q8k_quantize(x)
    oq, od, _ = oracle.q8k_quantize(x)
    assert np.array_equal(qs, oq) and np.array_equal(d, od)  # quantize_row_q8_K_ref (src/quant.cpp:616-653), bit for bit
    got = ctx.gemv(3, w, rows, n, x)
    want = oracle.gemv_q2k_tiles(w, rows, n, qs, d)
    assert np.array_equal(got, want), (np.abs(got - want).max(), np.abs(want).max())


@pytest.mark.parametrize("experts,mla", [(64, False), (256, False), (256, True)], ids=["64-mha", "256-mha", "256-mla"])
def test_full_width_block_tiled_fused_equals_two_launch(ctx, experts, mla):
    """one dense + one MoE block at DeepSeek-V3 width (top-8, 7168 / 2048) with 64 and with ALL 256 routed experts (the benchmarked
    count: 1.23 GB per W2 stack, the fused launch's 31-bit buffer offsets), MHA and MLA: the tiled fused expert launch
    (kernels_moe_tile.hip: 8 steps in registers + parked steps, Q8_K hand-over) against the tiled two-launch form and against the
    two-launch form without the shared expert's rider, bit for bit (logits and slot outputs)"""
    import dsk
    c = synth.preset("v3", "q2_k", mla, n_layers=2, first_k_dense_replace=1, n_routed_experts=experts, max_seq_len=64)
    A = dsk.Model(ctx, c, None, synth_seed=5)
    B = dsk.Model(ctx, c, None, synth_seed=5, options={"fuse_moe": 0})
    D = dsk.Model(ctx, c, None, synth_seed=5, options={"fuse_moe": 0, "fuse_shared": 0})
    assert A.info("fused_moe_layers") == 1 and B.info("fused_moe_layers") == 0 and D.info("fused_moe_layers") == 0
    assert A.info("tiled_tensors") > 0
    for pos, t in enumerate([3, 77, 1500, 9]):
        la, lb, ld = A.forward(t, pos), B.forward(t, pos), D.forward(t, pos)
        assert np.array_equal(la, lb), pos
        assert np.array_equal(la, ld), pos
        assert np.array_equal(A.slot_outputs(), B.slot_outputs()), pos
        assert np.array_equal(A.slot_outputs(), D.slot_outputs()), pos
        assert np.array_equal(A.routing()[0], B.routing()[0]), pos
    for M in (A, B, D):
        M.close()


@pytest.mark.parametrize("pipe", [1, 2], ids=["slot-halves", "staged-second-half"])
def test_experimental_schedules_of_the_fused_expert_launch_are_bit_identical(ctx, pipe):
    """option "moe_pipe" (kernels_moe_pipe.hip; off by default: measured slower, EXPERIMENTS.md 6.1): the fused expert launch pipelined
    by slot halves with service waves publishing through the scalar memory path (1), and the one-phase launch with its second half
    staged around the hand-off (2).  Other schedules of the same arithmetic (src/infer.cpp:853-904): logits, slot outputs and routing
    bit for bit, DeepSeek-V3 width with 256 experts, MHA and MLA"""
    import dsk
    for mla in (False, True):
        c = synth.preset("v3", "q2_k", mla, n_layers=3, first_k_dense_replace=1, max_seq_len=64)
        A = dsk.Model(ctx, c, None, synth_seed=5)
        P = dsk.Model(ctx, c, None, synth_seed=5, options={"moe_pipe": pipe})
        assert P.info("fused_moe_layers") == 2
        for pos, t in enumerate([3, 77, 1500, 9, 100000]):
            la, lp = A.forward(t, pos), P.forward(t, pos)
            assert np.array_equal(la, lp), (mla, pos)
            assert np.array_equal(A.slot_outputs(), P.slot_outputs()), (mla, pos)
            assert np.array_equal(A.routing()[0], P.routing()[0]), (mla, pos)
        assert P.info("handoff_fallbacks") == 0
        A.close()
        P.close()


def test_synthesized_weights_do_not_depend_on_the_layout(ctx):
    """dsk_model_synthesize fills tile records THROUGH the plane layout: one seed = one logical model at every q2k_tiles
    level, so an A/B across levels compares layouts, not models - first-token logits agree to float precision, routing identical"""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    outs = []
    for level in (0, 1, 2):
        M = dsk.Model(ctx, c, None, synth_seed=3, options={"q2k_tiles": level})
        outs.append((M.forward(17, 0).copy(), M.routing()[0].copy()))
        M.close()
    scale = np.abs(outs[0][0]).max()
    assert np.isfinite(scale) and scale > 0
    for lo, r in outs[1:]:
        assert np.array_equal(r, outs[0][1])
        assert np.abs(lo - outs[0][0]).max() <= 3e-4 * scale, np.abs(lo - outs[0][0]).max() / scale


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,mla", [("v3", False), ("v3", True), ("v2lite", False)], ids=["v3-mha", "v3-mla", "v2lite-mha"])
def test_weights_requested_ahead_of_the_staging_are_bit_identical(ctx, model, mla):
    """Round 6 (kernels_gemv.hip gemv_ahead_kernel / gemv_ahead_q8_kernel / gemv_kvwrite_ahead_kernel, option "gemv_ahead"): the
    first-stage projections, wo and the MLA second stage request their first weights behind the loads of their vector and ahead
    of its staging - another ORDER of the same loads and the same multiplies, so every logit and every cache row must keep its
    bits.  DeepSeek-V3 and DeepSeek-V2-Lite width (the kernels are instantiated for their row lengths), 1 dense + 2 MoE blocks,
    eight positions, graph replay; the counter says the kernels actually ran (and that the option turns them off)."""
    import dsk
    c = synth.preset(model, "q2_k", mla, n_layers=3, first_k_dense_replace=1, max_seq_len=64)
    A = dsk.Model(ctx, c, None, synth_seed=23, options={"gemv_ahead": 0})
    B = dsk.Model(ctx, c, None, synth_seed=23)
    assert A.info("gemv_ahead_plans") == 0
    assert B.info("gemv_ahead_plans") == c.n_layers * (3 if mla else 2)
    toks = [int(t) for t in np.random.default_rng(2).integers(0, c.vocab_size, 8)]
    for pos, t in enumerate(toks):
        la, lb = A.forward(t, pos), B.forward(t, pos)
        assert np.array_equal(la, lb), pos
    H, hd, vd = c.n_heads, c.qk_nope_head_dim + c.qk_rope_head_dim, c.v_head_dim
    for l in range(c.n_layers):
        names = (("nope_cache", c.kv_lora_rank), ("rope_cache", c.qk_rope_head_dim)) if mla else (("k_cache", H * hd), ("v_cache", H * vd))
        for name, width in names:
            assert np.array_equal(A.get_cache_rows(l, name, 0, len(toks), width), B.get_cache_rows(l, name, 0, len(toks), width))
    A.close()
    B.close()
