"""GPU parity at BASELINE.json's FULL sizes (DeepSeek-V3 shapes, Q2_K), through the C ABI.

The oracle cannot run a 61-block model in seconds, so full sizes are covered by
  * the V3 GEMV shapes against the oracle on a random SAMPLE of rows (rows are independent),
  * size-independent properties: power-of-two scaling of the activation is exact (Q8_K is
    scale-equivariant for powers of two, the integer dots do not change), bit-reproducibility,
    rows of an expert stack equal the same matrix bound alone,
  * a full-WIDTH, reduced-depth model (dim 7168, vocab 129280, 128 heads, 1 dense + 1 MoE block with
    16 experts of the true expert shape) against the oracle, with the statistical W2A8 criterion of
    tests/util.py.
"""
import zlib

import numpy as np
import pytest

from tests.util import rel_inf
from tools import synth

pytestmark = pytest.mark.gpu
Q2K = 3


def rand_q2k(rng, rows, n):
    b = rng.integers(0, 256, (rows * (n // 256), 84), dtype=np.uint8)
    d = (rng.uniform(0.5, 1.5, rows * (n // 256)) / np.sqrt(n) / 13.9).astype(np.float16)
    b[:, 80:82] = d.view(np.uint8).reshape(-1, 2)
    b[:, 82:84] = (1.5 * d.astype(np.float32)).astype(np.float16).view(np.uint8).reshape(-1, 2)
    return b.reshape(rows, -1)


V3_SHAPES = [("wo", 7168, 16384), ("dense_w2", 7168, 18432), ("wq_b", 24576, 1536), ("wkv_b", 32768, 512),
             ("wq_a", 1536, 7168), ("dense_w1", 18432, 7168), ("lm_head_slice", 16384, 7168)]


@pytest.mark.parametrize("name,rows,n", V3_SHAPES, ids=[s[0] for s in V3_SHAPES])
def test_v3_gemv_shapes_vs_oracle_on_sampled_rows(ctx, oracle, name, rows, n):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    w = rand_q2k(rng, rows, n)
    x = rng.standard_normal(n).astype(np.float32)
    out = ctx.gemv(Q2K, w, rows, n, x)
    assert np.all(np.isfinite(out))
    sample = np.unique(np.concatenate([[0, 1, rows - 1, rows // 2], rng.integers(0, rows, 60)]))
    ref = oracle.gemv(Q2K, np.ascontiguousarray(w[sample]), len(sample), n, x)
    # integer dots are exact; the f32 super-block sums follow a different (fixed) tree than the AVX2 lanes
    assert rel_inf(out[sample], ref) < 2e-5, (name, rel_inf(out[sample], ref))
    # power-of-two scaling: same int8 codes, scale 4x => every output exactly 4x
    out4 = ctx.gemv(Q2K, w, rows, n, (x * np.float32(4.0)).astype(np.float32))
    assert np.array_equal(out4, out * np.float32(4.0))
    # bit-reproducible
    assert np.array_equal(out, ctx.gemv(Q2K, w, rows, n, x))


Q3_SHAPES = [("wo", 7168, 16384), ("lm_head_slice", 32768, 7168), ("dense_w2", 7168, 18432), ("wq_b", 24576, 1536)]


@pytest.mark.parametrize("name,rows,n", Q3_SHAPES, ids=[s[0] for s in Q3_SHAPES])
def test_v3_q3k_gemv_shapes_vs_oracle_on_sampled_rows(ctx, oracle, name, rows, n):
    """Q3_K (W3A8) at the V3 widths: the 16-wave variants with 4 column steps in flight (the planner's cap for
    Q3_K) and the 64-lane rows of wo; random bytes are valid Q3_K blocks once d is a sane f16."""
    rng = np.random.default_rng(zlib.crc32(("q3" + name).encode()))
    b = rng.integers(0, 256, (rows * (n // 256), 110), dtype=np.uint8)
    d = (rng.uniform(0.5, 1.5, rows * (n // 256)) / np.sqrt(n) / 40.0).astype(np.float16)
    b[:, 108:110] = d.view(np.uint8).reshape(-1, 2)
    w = b.reshape(rows, -1)
    x = rng.standard_normal(n).astype(np.float32)
    out = ctx.gemv(4, w, rows, n, x)
    assert np.all(np.isfinite(out))
    sample = np.unique(np.concatenate([[0, 1, rows - 1, rows // 2], rng.integers(0, rows, 60)]))
    ref = oracle.gemv(4, np.ascontiguousarray(w[sample]), len(sample), n, x)
    assert rel_inf(out[sample], ref) < 2e-5, (name, rel_inf(out[sample], ref))
    out4 = ctx.gemv(4, w, rows, n, (x * np.float32(4.0)).astype(np.float32))
    assert np.array_equal(out4, out * np.float32(4.0))
    assert np.array_equal(out, ctx.gemv(4, w, rows, n, x))


def test_v3_expert_stack_slices_equal_standalone_matrices(ctx, oracle):
    rng = np.random.default_rng(7)
    E, rows, n = 6, 2048, 7168  # the routed-expert shape (w1 / w3)
    w = np.stack([rand_q2k(rng, rows, n) for _ in range(E)])
    x = rng.standard_normal(n).astype(np.float32)
    for e in (0, 3, 5):
        a = ctx.gemv_expert(Q2K, w, E, e, rows, n, x)
        b = ctx.gemv(Q2K, np.ascontiguousarray(w[e]), rows, n, x)
        assert np.array_equal(a, b), e
    sample = np.array([0, 17, 1023, 2047])
    ref = oracle.gemv(Q2K, np.ascontiguousarray(w[5][sample]), len(sample), n, x)
    assert rel_inf(ctx.gemv_expert(Q2K, w, E, 5, rows, n, x)[sample], ref) < 2e-5


def test_q8k_full_width_vectors_bit_exact(ctx, oracle):
    rng = np.random.default_rng(3)
    for n in (7168, 16384, 18432):
        x = (rng.standard_normal(n) * rng.uniform(0.01, 30.0)).astype(np.float32)
        x[rng.integers(0, n, 40)] = 0.0
        x[256:512] = 0.0  # an all-zero block: d = 0 (src/quant.cpp:626-631)
        a, b = ctx.q8k_quantize(x), oracle.q8k_quantize(x)
        for u, v in zip(a, b):
            assert np.array_equal(u, v), n


@pytest.mark.timeout(1500)
def test_full_width_reduced_depth_v3_model_vs_oracle(ctx, oracle):
    """dim 7168 / vocab 129280 / 128 heads / expert shape 2048x7168; 1 dense + 1 MoE block, 16 experts.

    At this width W2A8 is chaotic, not just discontinuous: a token quantises ~20 vectors of 7168..18432
    activations, so some int8 rounding almost always flips under a 1e-7 perturbation (a different but
    fixed summation tree in RMSNorm), the flip perturbs the next vector by ~1e-4, which flips dozens of
    roundings there, and the difference saturates at the intrinsic noise of 8-bit activations: ~2.5e-2 of
    the logit scale.  Measured three ways on this very model (MI355X box, scratch run recorded in
    DESIGN.md section 5): oracle vs the unmodified reference 2.4-2.7e-2, HIP vs reference 1.8-2.7e-2, HIP vs
    oracle 2.5e-2 -- and 1e-6 for the one token where no rounding flipped.  So the assertions are:
      * a token without a flip exists among the trials and agrees to 1e-5 (same arithmetic),
      * no trial is grossly wrong (< 0.1; a layout / indexing bug at full width gives O(1)),
      * where the prebuilt reference loads, HIP is within the noise floor of it (or twice the oracle's distance).
    """
    import dsk
    c = synth.preset("v3", "q2_k", False, n_layers=2, first_k_dense_replace=1, n_routed_experts=16, n_group=4,
                     topk_group=2, max_seq_len=64)
    T = synth.random_block_model(c, seed=5)
    M, O = dsk.Model(ctx, c, T), oracle.model(c, T)
    toks0 = [3, 4096, 99999, 123456, 77]
    lo = [O.forward(t, 0) for t in toks0]
    ro = [O.routing()[0].copy() for _ in [0]]  # routing of the last oracle token
    lh = [M.forward(t, 0) for t in toks0]
    errs = [rel_inf(a, b) for a, b in zip(lh, lo)]
    assert min(errs) < 1e-5, errs           # no flip => identical arithmetic
    assert max(errs) < 0.1, errs            # never grossly wrong
    assert np.array_equal(M.routing()[0], ro[0]) or errs[-1] > 1e-5
    # a short free-running sequence (KV cache, pos > 0) stays within the same noise floor
    seq = [11, 70000, 129279]
    e_seq = [rel_inf(M.forward(t, p), O.forward(t, p)) for p, t in enumerate(seq)]
    assert max(e_seq) < 0.1, e_seq
    # graph replay == eager, bit for bit, at full width too
    a = M.forward(11, 0)
    M.set_graph(False)
    b = M.forward(11, 0)
    assert np.array_equal(a, b)
    # the unmodified reference as the arbiter, when its prebuilt library is present on this box
    try:
        from oracle import orc as orcmod
        R = orcmod.Ref()
    except Exception:
        R = None
    if R is not None:
        import os
        import tempfile
        d = tempfile.mkdtemp(prefix="dsk_fullwidth_")
        try:
            synth.write_dseek(d, c, T)
            S = R.session(d, c, context=64)
            for t, l_h, l_o in zip(toks0, lh, lo):
                l_r = S.forward(int(t), 0)
                e_h, e_o = rel_inf(l_h, l_r), rel_inf(l_o, l_r)
                # HIP may flip on a token where the oracle (same serial sums as the reference) does not:
                # then e_o ~ 1e-6 and e_h sits at the noise floor
                assert e_h < max(2.0 * e_o, 0.05), (t, e_h, e_o)
            S.close()
        finally:
            for f in os.listdir(d):
                os.unlink(os.path.join(d, f))
            os.rmdir(d)
    M.close()
    O.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("quant", ["f8e5m2", "q2_k"])
def test_v2lite_full_width_reduced_depth_vs_oracle(ctx, oracle, quant):
    """BASELINE.json configs[1] / [2]: DeepSeek-V2-Lite shapes (dim 2048, 16 heads, vocab 102400, 64 routed
    experts top-6 greedy softmax, 2 shared, no q_lora; pad256 variant for Q2_K), 1 dense + 1 MoE block.
    F8E5M2 is a float path: logits within 1e-3, routing identical.  Q2_K: flip-free trials to 1e-5, none gross."""
    import dsk
    c = synth.preset("v2lite", quant, False, n_layers=2, first_k_dense_replace=1, max_seq_len=64)
    T = synth.random_block_model(c, seed=8)
    M, O = dsk.Model(ctx, c, T), oracle.model(c, T)
    errs, same = [], 0
    toks = [7, 1234, 50000, 102399, 99]
    for t in toks:
        lo, lh = O.forward(t, 0), M.forward(t, 0)
        errs.append(rel_inf(lh, lo))
        same += int(np.array_equal(M.routing()[0], O.routing()[0]))
    seq = [rel_inf(M.forward(t, p), O.forward(t, p)) for p, t in enumerate([5, 6, 7, 8])]
    M.close()
    O.close()
    if quant == "f8e5m2":
        assert max(errs + seq) < 1e-3, (errs, seq)
        assert same == len(toks)
    else:
        assert min(errs) < 1e-5 and max(errs + seq) < 0.1, (errs, seq)
        assert same >= len(toks) - 2


@pytest.mark.parametrize("quant", [3, 4], ids=["q2_k", "q3_k"])
def test_ragged_lane_split_v2lite_dense_w2(ctx, oracle, quant):
    """DeepSeek-V2-Lite pad256 dense w2: 11008-wide rows = 172 items, no power-of-two lane count >= 8 divides them.
    The planner takes 64 lanes per row with a ragged third step (dead lanes re-read the last block with a zero
    scale): same integers, same result as the oracle."""
    rng = np.random.default_rng(5)
    rows, n = 2048, 11008
    if quant == 3:
        w = rand_q2k(rng, rows, n)
    else:
        w = synth.encode_q3k(rng.standard_normal((rows, n)).astype(np.float32) / np.sqrt(n))
    x = rng.standard_normal(n).astype(np.float32)
    out = ctx.gemv(quant, w, rows, n, x)
    sample = np.unique(np.concatenate([[0, rows - 1], rng.integers(0, rows, 40)]))
    ref = oracle.gemv(quant, np.ascontiguousarray(w[sample]), len(sample), n, x)
    assert rel_inf(out[sample], ref) < 2e-5, rel_inf(out[sample], ref)
    assert np.array_equal(out, ctx.gemv(quant, w, rows, n, x))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
def test_ride_along_launches_equal_separate_launches_bit_for_bit(ctx, monkeypatch, mla):
    """At DeepSeek-V3 width two small jobs ride in a neighbour's launch: the shared expert's w1/w3 GLU in the router launch
    (router_shared_kernel) and, on the MLA path, the latent's cache write in the second-stage projection launch
    (gemv_kvwrite_kernel).  Both are the same arithmetic on other workgroups: logits, routing and slot outputs must be
    BIT-identical to a model that keeps the separate launches (options "fuse_shared" / "ride_kvwrite" = 0), across a
    free-running sequence (the cache written by the riding workgroup feeds the next tokens), eager and graph replay."""
    import dsk
    c = synth.preset("v3", "q2_k", mla, n_layers=2, first_k_dense_replace=1, n_routed_experts=16, n_group=4, topk_group=2, max_seq_len=64)
    A = dsk.Model(ctx, c, None, synth_seed=3)
    B = dsk.Model(ctx, c, None, synth_seed=3, options={"fuse_shared": 0, "ride_kvwrite": 0})
    tok = 11
    for pos in range(7):
        la, lb = A.forward(tok, pos), B.forward(tok, pos)
        assert np.array_equal(la, lb), pos
        assert np.array_equal(A.routing()[0], B.routing()[0]) and np.array_equal(A.routing()[1], B.routing()[1])
        assert np.array_equal(A.slot_outputs(), B.slot_outputs())
        tok = int(np.argmax(la))
    A.close()
    B.close()
