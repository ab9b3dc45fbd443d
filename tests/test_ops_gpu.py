"""GPU parity, op level: every stage of the hot path through the C ABI vs the CPU oracle (and the
reference's recorded outputs in tests/golden/ops.npz) on the same inputs.

Bars: bit-exact for integer / index work (Q8_K ints and sums, top-k expert indices);
float outputs within 1e-5 of the output scale (only the f32 summation ORDER differs: the
integer sub-block sums and every product are the same as the reference's).
"""
import numpy as np
import pytest

from tests.util import rel_inf
from tools import synth

pytestmark = pytest.mark.gpu

FTOL = 1e-5


# ----------------------------------------------------------------------------- Q8_K activations
@pytest.mark.parametrize("n", [256, 512, 1536, 7168, 18432])
def test_q8k_bitexact(ctx, oracle, n):
    rng = np.random.default_rng(n)
    for scale in (1e-3, 1.0, 77.0):
        x = (rng.standard_normal(n) * scale).astype(np.float32)
        a, b = ctx.q8k_quantize(x), oracle.q8k_quantize(x)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])


def test_q8k_edge_cases_vs_reference(ctx, ops_gold):
    # zero block, +/- tie on the max magnitude (first index wins), tiny magnitudes
    qs, d, bs = ctx.q8k_quantize(ops_gold["q8_x"])
    assert np.array_equal(qs, ops_gold["q8_qs"]) and np.array_equal(bs, ops_gold["q8_bsums"])
    assert np.array_equal(d, ops_gold["q8_d"])


def test_q8k_rejects_ragged(ctx):
    import dsk
    with pytest.raises(dsk.DskError):
        ctx.q8k_quantize(np.zeros(300, np.float32))


# ----------------------------------------------------------------------------- GEMV, all quants
KQ_SHAPES = [(5, 256), (16, 512), (24, 1536), (67, 2048), (9, 7168), (3, 11008), (300, 512), (33, 16384), (1030, 768)]


@pytest.mark.parametrize("quant,enc", [(3, synth.encode_q2k), (4, synth.encode_q3k)], ids=["q2_k", "q3_k"])
@pytest.mark.parametrize("d,n", KQ_SHAPES)
def test_gemv_kquant(ctx, oracle, quant, enc, d, n):
    rng = np.random.default_rng(d * 131 + n)
    w = (rng.standard_normal((d, n)) / np.sqrt(n)).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    wb = enc(w)
    y, yo = ctx.gemv(quant, wb, d, n, x), oracle.gemv(quant, wb, d, n, x)
    assert rel_inf(y, yo) < FTOL


@pytest.mark.parametrize("quant", [3, 4], ids=["q2_k", "q3_k"])
def test_gemv_kquant_random_bytes(ctx, oracle, quant):
    """Adversarial blocks: every bit pattern of qs / scales / hmask, not just encoder outputs."""
    rng = np.random.default_rng(99 + quant)
    d, n = 40, 1024
    bpb = 84 if quant == 3 else 110
    wb = rng.integers(0, 256, (d, n // 256 * bpb), dtype=np.uint8)
    blocks = wb.reshape(d, n // 256, bpb)
    # keep the f16 scales finite and modest: d, dmin in [2^-8, 2^-7)
    blocks[:, :, bpb - 1] = 0x1C
    if quant == 3:
        blocks[:, :, bpb - 3] = 0x1C
    x = rng.standard_normal(n).astype(np.float32)
    y, yo = ctx.gemv(quant, wb, d, n, x), oracle.gemv(quant, wb, d, n, x)
    assert rel_inf(y, yo) < FTOL


def test_gemv_power_of_two_scaling_is_exact(ctx):
    """W.A8 property: Q8_K quantisation commutes exactly with scaling x by 2^k, so the output
    scales exactly (size-independent check, also used at full V3 sizes)."""
    rng = np.random.default_rng(5)
    d, n = 64, 2048
    wb = synth.encode_q2k((rng.standard_normal((d, n)) / 45).astype(np.float32))
    x = rng.standard_normal(n).astype(np.float32)
    y1, y8 = ctx.gemv(3, wb, d, n, x), ctx.gemv(3, wb, d, n, x * 8)
    assert np.array_equal(y1 * 8, y8)
    assert np.array_equal(ctx.gemv(3, wb, d, n, np.zeros(n, np.float32)), np.zeros(d, np.float32))


@pytest.mark.parametrize("d,n", [(7, 128), (130, 2048), (256, 1408), (300, 10944), (64, 512)])
def test_gemv_f8e5m2_blocks(ctx, oracle, d, n):
    rng = np.random.default_rng(d + n)
    w = (rng.standard_normal((d, n)) / np.sqrt(n)).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    w8, s8 = synth.encode_f8_blocks(w, (128, 128))
    y, yo = ctx.gemv(2, w8, d, n, x, s8, (128, 128)), oracle.gemv(2, w8, d, n, x, s8, (128, 128))
    assert rel_inf(y, yo) < FTOL


@pytest.mark.parametrize("quant,dt", [(0, np.float32), (1, np.float16)], ids=["f32", "f16"])
@pytest.mark.parametrize("d,n", [(2, 16), (64, 2048), (257, 7168), (19, 48)])
def test_gemv_f32_f16(ctx, oracle, quant, dt, d, n):
    rng = np.random.default_rng(d * 7 + n)
    w = (rng.standard_normal((d, n)) / np.sqrt(n)).astype(dt)
    x = rng.standard_normal(n).astype(np.float32)
    y, yo = ctx.gemv(quant, w, d, n, x), oracle.gemv(quant, w, d, n, x)
    assert rel_inf(y, yo) < FTOL


def test_gemv_reference_kat(ctx):
    # the reference's own known-answer vectors, src/test.cpp:132-167 (shape {2,16})
    from tests.test_oracle_pin import KAT_W, KAT_X, KAT_Y
    assert np.allclose(ctx.gemv(0, KAT_W, 2, 16, KAT_X), KAT_Y, atol=1e-4)
    assert np.allclose(ctx.gemv(1, KAT_W.astype(np.float16), 2, 16, KAT_X), KAT_Y, atol=1e-3)
    w8 = (KAT_W.astype(np.float16).view(np.uint16) >> 8).astype(np.uint8)
    assert np.allclose(ctx.gemv(2, w8, 2, 16, KAT_X), [-3.36792, -2.92358], atol=2e-5)


def test_gemv_expert_slices(ctx, oracle, ops_gold):
    for e in range(4):
        y = ctx.gemv_expert(3, ops_gold["we_q2k"], 4, e, 32, 512, ops_gold["xe"])
        assert rel_inf(y, ops_gold["ye_q2k"][e]) < FTOL
    rng = np.random.default_rng(8)
    we = (rng.standard_normal((3, 130, 256)) / 16).astype(np.float32)
    x = rng.standard_normal(256).astype(np.float32)
    parts = [synth.encode_f8_blocks(we[e], (128, 128)) for e in range(3)]
    w8, s8 = np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts])
    for e in range(3):
        y = ctx.gemv_expert(2, w8, 3, e, 130, 256, x, s8, (128, 128))
        assert rel_inf(y, oracle.gemv_expert(2, w8, e, 130, 256, x, s8, (128, 128))) < FTOL


def test_gemv_vs_reference_golden(ctx, ops_gold):
    g = ops_gold
    for key, quant in (("q2k", 3), ("q3k", 4), ("q2k_ref", 3), ("q3k_ref", 4), ("f16", 1), ("f32", 0)):
        assert rel_inf(ctx.gemv(quant, g["w_" + key], 48, 1024, g["gemv_x"]), g["y_" + key]) < FTOL, key
    assert rel_inf(ctx.gemv(2, g["w_f8"], 256, 1024, g["gemv_x"], g["s_f8"], (128, 128)), g["y_f8"]) < FTOL


def test_gemv_bad_arguments(ctx):
    import dsk
    with pytest.raises(dsk.DskError):
        ctx.gemv(3, np.zeros((4, 84), np.uint8), 4, 300, np.zeros(300, np.float32))  # n % 256 (src/quantizer.cpp:8)
    with pytest.raises(dsk.DskError):
        ctx.gemv(3, np.zeros((4, 80), np.uint8), 4, 256, np.zeros(256, np.float32))  # wrong byte count
    with pytest.raises(dsk.DskError):
        ctx.gemv(9, np.zeros((4, 84), np.uint8), 4, 256, np.zeros(256, np.float32))  # unknown quant


# ----------------------------------------------------------------------------- embedding rows
@pytest.mark.parametrize("quant", ["q2_k", "q3_k", "f8e5m2", "fp16", "fp32"])
def test_embed_row(ctx, oracle, quant):
    rng = np.random.default_rng(11)
    V, dim = 300, 512
    t = synth._encode(rng.standard_normal((V, dim)).astype(np.float32), quant, (128, 128))
    for token in (0, 1, 127, 128, 299):
        y = ctx.embed_row(t.quant, t.data, V, dim, token, t.scale, (128, 128))
        yo = oracle.embed_row(t.quant, t.data, dim, token, t.scale, (128, 128))
        assert rel_inf(y, yo) < 1e-6


# ----------------------------------------------------------------------------- small ops
@pytest.mark.parametrize("n", [512, 1536, 7168])
def test_rmsnorm(ctx, oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    assert rel_inf(ctx.rmsnorm(x, w, 1e-6), oracle.rmsnorm(x, w, 1e-6)) < 1e-6


def test_moe_gate_indices_bitexact(ctx, oracle, ops_gold):
    g = ops_gold
    e, w = ctx.moe_gate(g["gate3_s"], g["gate3_b"], 8, True, 2.5, 1, 1, 8, 4)  # ties: lowest index wins
    assert np.array_equal(e, g["gate3_e"]) and rel_inf(w, g["gate3_w"]) < 1e-5
    e, w = ctx.moe_gate(g["gate2_s"], None, 6, False, 1.0, 0, 0, 1, 1)
    assert np.array_equal(e, g["gate2_e"]) and rel_inf(w, g["gate2_w"]) < 1e-5
    rng = np.random.default_rng(12)
    for trial in range(40):
        v3 = trial % 2 == 0
        E, K = (256, 8) if v3 else (64, 6)
        s = rng.standard_normal(E).astype(np.float32)
        if trial % 5 == 0:
            s[rng.integers(0, E, 6)] = s[0]  # inject exact ties
        b = (0.1 * rng.standard_normal(E)).astype(np.float32) if v3 else None
        args = (8, True, 2.5, 1, 1, 8, 4) if v3 else (6, False, 1.0, 0, 0, 1, 1)
        eg, wg = ctx.moe_gate(s, b, *args)
        eo, wo, _ = oracle.moe_gate(s, b, *args)
        if v3 or len(set(np.exp(s - s.max()).astype(np.float32))) == E:  # softmax may merge near-equal scores
            assert np.array_equal(eg, eo), (trial, eg, eo)
            assert rel_inf(wg, wo) < 1e-5
    # shapes whose groups / candidate lists are not multiples of 16: the one-lane-per-candidate rank loops (the shapes above take
    # the four-lanes-per-candidate form of router_device.h gate_body), with exact ties
    for trial, (E, K, args) in enumerate([(96, 6, (6, True, 1.5, 1, 1, 4, 2)), (40, 6, (6, True, 1.0, 1, 0, 1, 1)), (200, 8, (8, True, 2.5, 1, 1, 8, 4))] * 4):
        s = rng.standard_normal(E).astype(np.float32)
        if trial % 2 == 0:
            s[rng.integers(0, E, 8)] = s[1]
        b = (0.1 * rng.standard_normal(E)).astype(np.float32)
        eg, wg = ctx.moe_gate(s, b, *args)
        eo, wo, _ = oracle.moe_gate(s, b, *args)
        assert np.array_equal(eg, eo), (E, trial, eg, eo)
        assert rel_inf(wg, wo) < 1e-5


def test_rope(ctx, oracle, ops_gold):
    g = ops_gold
    assert np.allclose(ctx.rope(g["rope_in"], 1, 64, 1234, 10000.0, False), g["rope_v2"], atol=1e-6)
    assert np.allclose(ctx.rope(g["rope_in"], 1, 64, 1234, 10000.0, True), g["rope_v3"], atol=1e-6)
    rng = np.random.default_rng(13)
    v = rng.standard_normal(4 * 32).astype(np.float32)
    for v3 in (False, True):
        want = np.concatenate([oracle.rope(v[h * 32:(h + 1) * 32], 32, 77, 10000.0, v3) for h in range(4)])
        assert np.allclose(ctx.rope(v, 4, 32, 77, 10000.0, v3), want, atol=1e-6)


@pytest.mark.parametrize("kv_len", [1, 2, 70, 300, 1500])
def test_attn_mha(ctx, oracle, kv_len):
    rng = np.random.default_rng(kv_len)
    H, hd, vd = 3, 192, 128
    q = rng.standard_normal(H * hd).astype(np.float32)
    kb = (rng.standard_normal((kv_len, H * hd)) * 0.5).astype(np.float16).view(np.uint16)
    vb = rng.standard_normal((kv_len, H * vd)).astype(np.float16).view(np.uint16)
    assert rel_inf(ctx.attn_mha(q, kb, vb, H, hd, vd, kv_len), oracle.attn_mha(q, kb, vb, H, hd, vd, kv_len)) < FTOL


@pytest.mark.parametrize("kv_len", [1, 70, 700])
def test_attn_mla(ctx, oracle, kv_len):
    rng = np.random.default_rng(kv_len + 1)
    H, lora, rope = 5, 512, 64
    qc = (rng.standard_normal(H * lora) / 8).astype(np.float32)
    qr = rng.standard_normal(H * rope).astype(np.float32)
    ckv = rng.standard_normal((kv_len, lora)).astype(np.float16).view(np.uint16)
    kr = rng.standard_normal((kv_len, rope)).astype(np.float16).view(np.uint16)
    y = ctx.attn_mla(qc, qr, ckv, kr, H, 192, lora, rope, kv_len)
    assert rel_inf(y, oracle.attn_mla(qc, qr, ckv, kr, H, 192, lora, rope, kv_len)) < FTOL


def test_attn_vs_reference_golden_and_kat(ctx, ops_gold):
    g = ops_gold
    assert rel_inf(ctx.attn_mha(g["att_q"], g["att_k"], g["att_v"], 4, 192, 128, 70), g["att_y"]) < FTOL
    assert rel_inf(ctx.attn_mla(g["mla_qc"], g["mla_qr"], g["mla_ckv"], g["mla_kr"], 4, 192, 512, 64, 70), g["mla_y"]) < FTOL
    # src/test.cpp:84-125 one-hot case; head_dim padded to a multiple of 4 with zeros
    kb = np.zeros((4, 4), np.float32)
    kb[:, :3] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]]
    kb16 = kb.astype(np.float16).view(np.uint16)
    for q, want in (([0., 1e4, 0., 0.], [0, 1, 0, 0]), ([0., 0., 1e4, 0.], [0, 0, 1, 0])):
        out = ctx.attn_mha(np.array(q, np.float32), kb16, kb16, 1, 4, 4, 4)
        assert np.allclose(out, want, atol=1e-6)


@pytest.mark.parametrize("kv_len", [320, 333, 768, 1001, 2100, 4100])
def test_attn_mla_long_context_matrix_core_path(ctx, oracle, kv_len):
    """kv_len >= 320 takes the MFMA path (f16 matrix cores on the f16 cache entries, q and the softmax weights as hi + lo
    halves: exact products, f32 accumulation; chunked online softmax + merge): same result as the reference's attn_mla
    (src/infer.cpp:766-804) up to f32 summation order; 1 and 2-3 blocks of 32 positions per chunk, ragged last blocks."""
    rng = np.random.default_rng(kv_len)
    H, lora, rope, hd = 40, 512, 64, 192  # 40 heads: a ragged last head group
    q_c = rng.standard_normal((H, lora)).astype(np.float32) * 0.2
    q_r = rng.standard_normal((H, rope)).astype(np.float32) * 0.2
    ckv = rng.standard_normal((kv_len, lora)).astype(np.float16)
    kr = rng.standard_normal((kv_len, rope)).astype(np.float16)
    got = ctx.attn_mla(q_c, q_r, ckv.view(np.uint16), kr.view(np.uint16), H, hd, lora, rope, kv_len)
    ref = oracle.attn_mla(q_c, q_r, ckv.view(np.uint16), kr.view(np.uint16), H, hd, lora, rope, kv_len)
    assert np.max(np.abs(got - ref)) < 2e-5 * max(1.0, float(np.max(np.abs(ref)))), float(np.max(np.abs(got - ref)))


# --------------------------------------------------------------------------- the sampler on the device
def _cdf_distance(logits, t, p, coin, tok):
    """How far (in cumulative probability, f64) r = coin * top_p lies outside token `tok`'s CDF interval; 0 if inside."""
    z = (logits.astype(np.float64) - logits.max()) / t
    cdf = np.cumsum(np.exp(z) / np.exp(z).sum())
    r = float(np.float32(coin) * np.float32(p))
    lo = cdf[tok - 1] if tok > 0 else 0.0
    return max(0.0, lo - r, r - cdf[tok])


def _near_tie(logits, t, p, coin, a, b):
    """Both tokens are the inverse CDF at r up to the f32 rounding of a 129k-term running sum."""
    return _cdf_distance(logits, t, p, coin, a) < 1e-4 and _cdf_distance(logits, t, p, coin, b) < 1e-4


def test_sampler_golden_cases(ctx, oracle):
    """dsk_sample against the reference's own Sampler::sample outputs (tests/golden/sampler.npz)."""
    import os
    from tools.make_golden import sampler_cases
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler.npz"))
    V, cases = sampler_cases()
    for k, (logits, t, p, seed) in enumerate(cases):
        if int(np.sum(logits.view(np.uint32) % 65521)) != int(g["logits_crc"][k]):
            pytest.skip("numpy generates different logits than when the fixture was made")
        tok = ctx.sample(logits, float(t), float(p), float(g["coin"][k]))
        assert tok == int(g["token"][k]) or (t > 0 and _near_tie(logits, float(t), float(p), float(g["coin"][k]), tok, int(g["token"][k]))), k


def test_sampler_full_vocabulary_vs_oracle(ctx, oracle):
    """129 280 logits (DeepSeek-V3's vocabulary).  The reference's left-to-right f32 running sums (oracle/dsk_oracle.c
    orc_sample, pinned to the reference) and the device's fixed-tree sums are both the inverse CDF at r = coin * top_p,
    each with its own rounding: r must lie within 1e-4 of cumulative probability of BOTH tokens' CDF intervals (the exact
    f64 CDF is the yardstick; the oracle itself is up to ~3e-5 away from it), and on peaked distributions - what a
    language model emits - the tokens are the same."""
    rng = np.random.default_rng(5)
    V = 129280
    same_peaked = n_peaked = 0
    for k in range(60):
        spread = float(rng.uniform(0.5, 8.0))
        logits = (rng.standard_normal(V) * spread).astype(np.float32)
        t = float(rng.choice([0.6, 1.0, 1.4]))
        p = float(rng.choice([0.5, 0.95, 1.0]))
        coin = float(np.float32(rng.uniform()))
        a, b = ctx.sample(logits, t, p, coin), oracle.sample(logits, t, p, coin)
        assert a == b or _near_tie(logits, t, p, coin, a, b), (k, a, b, _cdf_distance(logits, t, p, coin, a), _cdf_distance(logits, t, p, coin, b))
        assert _cdf_distance(logits, t, p, coin, a) < 1e-4, k
        assert ctx.sample(logits, t, p, coin) == a  # deterministic
        if spread / t > 5.0:  # a few hundred tokens hold the mass
            n_peaked += 1
            same_peaked += int(a == b)
    assert n_peaked >= 5 and same_peaked >= n_peaked - 1, (same_peaked, n_peaked)
    # edges: temperature 0 = first maximum; coin 0 -> token 0 (cumsum >= 0 at once); coin * top_p beyond the total mass -> V - 1
    logits = rng.standard_normal(V).astype(np.float32)
    logits[[77, 5000, 129279]] = logits.max() + 1.0
    assert ctx.sample(logits, 0.0, 0.95, 0.3) == 77 == oracle.sample(logits, 0.0, 0.95, 0.3)
    assert ctx.sample(logits, 1.0, 0.95, 0.0) == 0 == oracle.sample(logits, 1.0, 0.95, 0.0)
    assert ctx.sample(logits, 1.0, 1.0, 1.01) == V - 1 == oracle.sample(logits, 1.0, 1.0, 1.01)
