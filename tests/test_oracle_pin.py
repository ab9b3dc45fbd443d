"""Pin the CPU oracle (oracle/dsk_oracle.c) to the reference.

Three layers of evidence, all CPU-only:
  1. the known-answer vectors inside the reference's own src/test.cpp;
  2. tests/golden/ops.npz + model_*.npz: outputs of the unmodified reference compiled from
     /root/reference (tools/make_golden.py), committed as fixtures;
  3. when oracle/_ref/libdskref.so is loadable, live comparisons on fresh random inputs.
Integer work (Q8_K ints, top-k indices, codecs) must be bit-exact; the AVX2-lane-exact GEMVs are
bit-exact too; stages whose float order the reference leaves to -ffast-math are by tolerance.
"""
import os

import numpy as np
import pytest

from tests.util import MODEL_CASES, assert_model_parity, case_id, is_kquant, load_case, model_parity_stats, rel_inf
from tools import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# --------------------------------------------------------------------------- 1. reference KATs
KAT_X = np.array([2.0624e-01, 1.6975e+00, 8.4918e-01, -1.7186e-01, -9.0164e-01, 6.1108e-01, 2.2116e-01, 1.0412e+00,
                  -1.6616e-03, 8.2840e-01, 2.2667e-01, -1.3993e+00, 4.1013e-01, -1.2223e+00, 2.2723e-01, 6.3558e-01],
                 np.float32)  # src/test.cpp:132-137
KAT_W = np.array([[-1.1210, -0.0235, -1.3527, 0.6300, 0.2566, -0.4517, -0.3528, 0.4422,
                   -0.4032, -1.0949, -0.7834, 1.1425, 0.6263, -0.3680, 0.3226, -0.2984],
                  [0.1176, -1.1462, -0.8181, -2.0047, 0.0932, 1.4665, -0.8682, -0.8490,
                   -1.3017, -1.0068, -0.2890, 0.0167, 1.1607, 0.7196, 1.7701, 0.2891]], np.float32)  # src/test.cpp:138-145
KAT_Y = np.array([-3.7454, -3.2738], np.float32)  # src/test.cpp:150,157,164 (with shape {2,16})


def test_kat_matmul_f32_f16(oracle):
    assert np.allclose(oracle.gemv(0, KAT_W, 2, 16, KAT_X), KAT_Y, atol=1e-4)
    assert np.allclose(oracle.gemv(1, KAT_W.astype(np.float16), 2, 16, KAT_X), KAT_Y, atol=1e-3)


def test_kat_f8e5m2(oracle):
    L = oracle.lib
    for v in (1.0, -1.5, 0.109375):  # src/test.cpp:129-131
        assert L.orc_f8e5m2_to_float(L.orc_float_to_f8e5m2(v)) == v
    w8 = np.array([L.orc_float_to_f8e5m2(float(v)) for v in KAT_W.ravel()], np.uint8).reshape(2, 16)
    y = oracle.gemv(2, w8, 2, 16, KAT_X)
    assert np.allclose(y, KAT_Y, atol=3.78e-1)                     # the file's own tolerance (src/test.cpp:167)
    assert np.allclose(y, [-3.36792, -2.92358], atol=2e-5)         # what the shipped (truncating) code gives
    rt = np.array([L.orc_f8e5m2_to_float(L.orc_float_to_f8e5m2(float(v))) for v in KAT_X], np.float32)
    assert np.array_equal(rt, np.array([0.1875, 1.5, 0.75, -0.15625, -0.875, 0.5, 0.21875, 1, -0.00146484, 0.75, 0.21875,
                                        -1.25, 0.375, -1, 0.21875, 0.625], np.float32).astype(np.float16).astype(np.float32))


def test_kat_attn_onehot(oracle):
    # src/test.cpp:84-125 with the single-KV-head stride the shipped attn() expects
    kb = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]], np.float32).astype(np.float16).view(np.uint16)
    for q, want in (([0., 1e4, 0.], [0, 1, 0]), ([0., 0., 1e4], [0, 0, 1])):
        out = oracle.attn_mha(np.array(q, np.float32), kb, kb, 1, 3, 3, 4)
        assert np.allclose(out, want, atol=1e-6)


# --------------------------------------------------------------------------- 2. golden fixtures
def test_golden_q8k(oracle, ops_gold):
    qs, d, bs = oracle.q8k_quantize(ops_gold["q8_x"])
    assert np.array_equal(qs, ops_gold["q8_qs"])
    assert np.array_equal(bs, ops_gold["q8_bsums"])
    assert np.array_equal(d, ops_gold["q8_d"])


@pytest.mark.parametrize("key,quant", [("q2k", 3), ("q3k", 4), ("q2k_ref", 3), ("q3k_ref", 4), ("f16", 1), ("f32", 0)])
def test_golden_gemv_bitexact(oracle, ops_gold, key, quant):
    w = ops_gold["w_" + key]
    y = oracle.gemv(quant, w, 48, 1024, ops_gold["gemv_x"])
    assert np.array_equal(y, ops_gold["y_" + key])


def test_golden_gemv_f8(oracle, ops_gold):
    y = oracle.gemv(2, ops_gold["w_f8"], 256, 1024, ops_gold["gemv_x"], ops_gold["s_f8"], (128, 128))
    assert np.array_equal(y, ops_gold["y_f8"])


def test_golden_experts_and_dequant(oracle, ops_gold):
    for e in range(4):
        y = oracle.gemv_expert(3, ops_gold["we_q2k"], e, 32, 512, ops_gold["xe"])
        assert np.array_equal(y, ops_gold["ye_q2k"][e])
    assert np.array_equal(oracle.dequant_row(3, ops_gold["w_q2k_ref"][5], 1024), ops_gold["deq_q2k"])
    assert np.array_equal(oracle.dequant_row(4, ops_gold["w_q3k_ref"][5], 1024), ops_gold["deq_q3k"])


def test_golden_small_ops(oracle, ops_gold):
    g = ops_gold
    assert rel_inf(oracle.rmsnorm(g["rms_x"], g["rms_w"], 1e-6), g["rms_y"]) < 1e-6
    e, w, sc = oracle.moe_gate(g["gate3_s"], g["gate3_b"], 8, True, 2.5, 1, 1, 8, 4)
    assert np.array_equal(e, g["gate3_e"]) and rel_inf(w, g["gate3_w"]) < 1e-6 and rel_inf(sc, g["gate3_scores"]) < 1e-6
    e, w, sc = oracle.moe_gate(g["gate2_s"], None, 6, False, 1.0, 0, 0, 1, 1)
    assert np.array_equal(e, g["gate2_e"]) and rel_inf(w, g["gate2_w"]) < 1e-6
    assert np.allclose(oracle.rope(g["rope_in"], 64, 1234, 10000.0, False), g["rope_v2"], atol=1e-6)
    assert np.allclose(oracle.rope(g["rope_in"], 64, 1234, 10000.0, True), g["rope_v3"], atol=1e-6)
    assert rel_inf(oracle.attn_mha(g["att_q"], g["att_k"], g["att_v"], 4, 192, 128, 70), g["att_y"]) < 1e-5
    assert rel_inf(oracle.attn_mla(g["mla_qc"], g["mla_qr"], g["mla_ckv"], g["mla_kr"], 4, 192, 512, 64, 70), g["mla_y"]) < 1e-5


def test_golden_codecs(oracle, ops_gold):
    L = oracle.lib
    vals = ops_gold["codec_in"]
    assert np.array_equal(np.array([L.orc_float_to_half(float(v)) for v in vals], np.uint16), ops_gold["codec_f16"])
    assert np.array_equal(np.array([L.orc_float_to_f8e5m2(float(v)) for v in vals], np.uint8), ops_gold["codec_f8"])
    h2f = np.array([L.orc_half_to_float(int(h)) for h in range(0, 65536, 97)], np.float32)
    assert np.array_equal(h2f.view(np.uint32), ops_gold["codec_h2f"].view(np.uint32))


@pytest.mark.parametrize("case", MODEL_CASES, ids=case_id)
def test_golden_model(oracle, case):
    """Free-running token steps: oracle vs the reference's recorded logits / routing.

    Float (fp32/fp16/f8) models must agree to 1e-3 of the logit scale on every token.  W2A8/W3A8
    models are discontinuous in their inputs (a 1e-7 perturbation of an activation flips an int8
    rounding, SURVEY 7 "hard parts"), so on these tiny dims a flipped rounding moves the logits by
    up to ~1e-2: the median token must still be within 1e-3 and every token within 5e-2.
    """
    c, T, g, sha_ok = load_case(case)
    if not sha_ok:
        pytest.skip("synthetic weights differ from the ones the fixture was generated with (numpy RNG change)")
    M = oracle.model(c, T)
    st = model_parity_stats(M, c, g)
    M.close()
    assert_model_parity(st, is_kquant(c.quant), "oracle vs reference")


# --------------------------------------------------------------------------- 3. live vs the reference build
def test_live_gemv_and_q8_bitexact(oracle, ref):
    rng = np.random.default_rng(3)
    for n, d in ((256, 5), (512, 16), (1536, 24), (7168, 8)):
        x = (rng.standard_normal(n) * rng.uniform(0.01, 30)).astype(np.float32)
        a, b = oracle.q8k_quantize(x), ref.q8k_quantize(x)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))
        w = (rng.standard_normal((d, n)) / np.sqrt(n)).astype(np.float32)
        for quant, enc in ((3, synth.encode_q2k), (4, synth.encode_q3k)):
            wb = enc(w)
            assert np.array_equal(oracle.gemv(quant, wb, d, n, x), ref.gemv(quant, wb, d, n, x))
        w8, s8 = synth.encode_f8_blocks(w, (128, 128))
        assert np.array_equal(oracle.gemv(2, w8, d, n, x, s8, (128, 128)), ref.gemv(2, w8, d, n, x, s8, (128, 128)))


@pytest.mark.parametrize("rows,n", [(16, 256), (37, 512), (64, 2048), (130, 2304), (48, 7168), (16, 11008)])
def test_tiled_association_stays_within_float_rounding_of_the_reference_order(oracle, rows, n):
    """orc_gemv_q2k_tiles restates the f32 association of the device's tiled Q2_K kernels (csrc/tile_device.h) on the integers
    of ggml_vec_dot_q2_K_q8_K (src/quant.cpp:666-783).  Against orc_gemv_q8 - pinned bit for bit to the reference above - the
    two may differ by float rounding only: a wrong sub-block, scale or field would show at 1e-3 and up.  Rows of one block, of
    <= 8 blocks (one item per block) and of more (four-block items, a ragged last item at 43 blocks)."""
    rng = np.random.default_rng(rows * 131 + n)
    w = synth.encode_q2k((rng.standard_normal((rows, n)) / np.sqrt(n)).astype(np.float32))
    x = (rng.standard_normal(n) * rng.uniform(0.05, 20)).astype(np.float32)
    qs, d, _ = oracle.q8k_quantize(x)
    want = oracle.gemv_q8(3, w, rows, n, qs, d)
    got = oracle.gemv_q2k_tiles(w, rows, n, qs, d)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    # exact where f32 cannot round: one block per row, integer-valued scales (d = 1, dmin = 1, dx = 1)
    if n == 256:
        w1 = w.copy().reshape(rows, 84)
        w1[:, 80:84] = np.frombuffer(np.array([1.0, 1.0], np.float16).tobytes(), np.uint8)
        q1 = rng.integers(-127, 128, 256).astype(np.int8)
        one = np.ones(1, np.float32)
        assert np.array_equal(oracle.gemv_q2k_tiles(w1, rows, n, q1, one), oracle.gemv_q8(3, w1, rows, n, q1, one))


def test_scaled_int8_expansion_of_q2k_reproduces_the_reference_block_sums(oracle):
    """a 2-bit code times its sub-block's 4-bit scale fits an int8 (<= 45), and a packed 32-bit multiply of four masked
    codes by one scale does not carry between bytes - so Q2_K weights expand to scaled int8 in registers and a PLAIN int8 dot over the
    256 columns gives the isum of ggml_vec_dot_q2_K_q8_K (src/quant.cpp:666-783) exactly.  Checked here on the packed arithmetic
    itself, against the oracle's vec_dot on blocks with d = dmin = 1 and a unit activation scale (f32 cannot round those)."""
    rng = np.random.default_rng(77)
    rows, n = 8, 1024
    w = synth.encode_q2k((rng.standard_normal((rows, n)) / np.sqrt(n)).astype(np.float32)).reshape(rows, n // 256, 84).copy()
    w[:, :, 80:84] = np.frombuffer(np.array([1.0, 1.0], np.float16).tobytes(), np.uint8)
    a = rng.integers(-127, 128, n).astype(np.int8)
    want = oracle.gemv_q8(3, np.ascontiguousarray(w.reshape(rows, -1)), rows, n, a, np.ones(n // 256, np.float32))
    got = np.zeros(rows, np.float64)
    for r in range(rows):
        for b in range(n // 256):
            blk = w[r, b]
            sc, qs = blk[:16].astype(np.uint32), blk[16:80].copy().view("<u4")  # 16 dwords of packed codes
            act = a[b * 256:(b + 1) * 256].astype(np.int64)
            isum = 0
            for dw in range(16):          # dword dw holds bytes 4 dw .. 4 dw + 3 of the 64: byte t of half h = 32 h + l
                h, l0 = (4 * dw) // 32, (4 * dw) % 32
                for s_ in range(4):       # field s_ of byte (h, l) is element 128 h + 32 s_ + l, sub-block 8 h + 2 s_ + l // 16
                    j = 8 * h + 2 * s_ + l0 // 16
                    packed = ((int(qs[dw]) >> (2 * s_)) & 0x03030303) * int(sc[j] & 0xF)   # ONE 32-bit multiply for four weights
                    assert packed < 1 << 32
                    by = [(packed >> (8 * t)) & 0xFF for t in range(4)]
                    assert max(by) <= 45                                                   # fits an int8, no carry between bytes
                    e0 = 128 * h + 32 * s_ + l0
                    isum += sum(by[t] * int(act[e0 + t]) for t in range(4))
            summs = sum(int(sc[j] >> 4) * int(act[16 * j:16 * j + 16].sum()) for j in range(16))
            got[r] += isum - summs
    assert np.array_equal(got.astype(np.float32), want)


def test_group_operands_and_digit_words_of_the_16_token_gemm_reproduce_the_reference_block_sums(oracle):
    """DESIGN.md 4.9, second GEMM form, modelled lane by lane on the CPU: one 16x16x64 matrix instruction per sub-block GROUP g,
    lane (row n, K-group kg') supplying sub-block 4 g + kg' = field 2 (g & 1) + (kg' >> 1) of the qs bytes of K-group
    2 (g >> 1) + (kg' & 1) (its own 16 bytes or those of the lane 32 away), times its 4-bit scale; the token's 16 codes of that
    sub-block in natural order on the other side; and the min term as ONE more instruction per group: the sub-block sums split
    b = 8 (v1 + v2) + v0 into int8 digits against {m, 8 m, 8 m, 0}.  The sums must be the isum / summs of ggml_vec_dot_q2_K_q8_K
    (src/quant.cpp:666-783) exactly."""
    rng = np.random.default_rng(78)
    # the digit split over the whole range of a Q8_K sub-block sum
    for b in range(-2032, 2033):
        v0, q = b & 7, b >> 3
        v1 = q >> 1
        v2 = q - v1
        assert 0 <= v0 <= 7 and -128 <= v1 <= 127 and -128 <= v2 <= 127 and 8 * (v1 + v2) + v0 == b
    rows, n, T = 16, 512, 5
    w = synth.encode_q2k((rng.standard_normal((rows, n)) / np.sqrt(n)).astype(np.float32)).reshape(rows, n // 256, 84).copy()
    w[:, :, 80:84] = np.frombuffer(np.array([1.0, 1.0], np.float16).tobytes(), np.uint8)
    acts = rng.integers(-127, 128, (T, n)).astype(np.int8)
    for t in range(T):
        want = oracle.gemv_q8(3, np.ascontiguousarray(w.reshape(rows, -1)), rows, n, acts[t], np.ones(n // 256, np.float32))
        got = np.zeros(rows, np.float64)
        for b in range(n // 256):
            a = acts[t, b * 256:(b + 1) * 256].astype(np.int64)
            bsum = [int(a[16 * j:16 * j + 16].sum()) for j in range(16)]
            for r in range(rows):
                blk = w[r, b]
                sc, qs = blk[:16].astype(np.int64), blk[16:80].astype(np.int64)
                lane_bytes = [qs[16 * kg:16 * kg + 16] for kg in range(4)]      # the tile record: lane (n, kg) holds qs bytes [16 kg, 16 kg + 16)
                for g in range(4):
                    D = M = 0
                    for kgp in range(4):                                        # K-group kg' of the instruction for group g
                        kg_src = 2 * (g >> 1) + (kgp & 1)                       # own lane iff kg_src == kgp, else the lane 32 away
                        assert kg_src == kgp or abs(kg_src - kgp) == 2
                        field = 2 * (g & 1) + (kgp >> 1)
                        j = 4 * g + kgp
                        assert j == 8 * (kg_src >> 1) + 2 * field + (kg_src & 1)   # the sub-block those bytes' field belongs to
                        wk = ((lane_bytes[kg_src] >> (2 * field)) & 3) * (sc[j] & 0xF)
                        assert wk.max() <= 45
                        D += int((wk * a[16 * j:16 * j + 16]).sum())            # the token's codes of sub-block j, natural order
                        v0, q = bsum[j] & 7, bsum[j] >> 3
                        v1 = q >> 1
                        m = int(sc[j] >> 4)
                        assert 8 * m <= 120
                        M += v0 * m + v1 * 8 * m + (q - v1) * 8 * m
                    got[r] += D - M
        assert np.array_equal(got.astype(np.float32), want), t


def test_live_router_gate_exact(oracle, ref):
    rng = np.random.default_rng(4)
    for _ in range(20):
        s = rng.standard_normal(256).astype(np.float32)
        b = (0.1 * rng.standard_normal(256)).astype(np.float32)
        eo, wo, _ = oracle.moe_gate(s, b, 8, True, 2.5, 1, 1, 8, 4)
        er, wr, _ = ref.moe_gate(s, b, 8, True, 2.5, 1, 1, 8, 4)
        assert np.array_equal(eo, er) and rel_inf(wo, wr) < 1e-6


# --------------------------------------------------------------------------- 4. the sampler
def test_golden_sampler(oracle):
    """orc_sample against Sampler::sample / sample_argmax of the reference itself (tests/golden/sampler.npz, made by
    tools/make_golden.py gen_sampler through oracle/ref_shim.cpp ref_sample): same token for every case - flat and peaked
    distributions, ties of the maximum, temperature 0 / 0.7 / 1 / 1.5, top_p 0.5 ... 1."""
    from tools.make_golden import sampler_cases
    g = np.load(os.path.join(GOLD, "sampler.npz"))
    V, cases = sampler_cases()
    assert len(cases) == len(g["token"])
    for k, (logits, t, p, seed) in enumerate(cases):
        if int(np.sum(logits.view(np.uint32) % 65521)) != int(g["logits_crc"][k]):
            pytest.skip("numpy generates different logits than when the fixture was made")
        assert (t, p, seed) == (g["temperature"][k], g["top_p"][k], g["seed"][k])
        assert oracle.sample(logits, float(t), float(p), float(g["coin"][k])) == int(g["token"][k]), k


def test_live_sampler(oracle, ref, tmp_path):
    c = synth.preset("tiny_v3", "fp16", False)
    d = str(tmp_path / "m")
    synth.write_dseek(d, c, synth.synth_model(c, seed=1))
    S = ref.session(d, c)
    rng = np.random.default_rng(12)
    for k in range(300):
        logits = (rng.standard_normal(c.vocab_size) * rng.uniform(0.3, 10)).astype(np.float32)
        t = float(rng.choice([0.0, 0.5, 1.0, 1.3]))
        p = float(rng.choice([0.5, 0.9, 0.95, 1.0]))
        tok, coin = S.sample(logits, t, p, 100 + k)
        assert oracle.sample(logits, t, p, coin) == tok, (k, t, p, coin)
        idx = int(rng.integers(0, c.vocab_size))
        a, b = oracle.sample_prob(logits, idx), S.sample_prob(logits, idx)
        # the reference is built with -ffast-math: gcc may vectorise its sum loop (8 partial sums, libmvec expf), so the
        # probability agrees to the rounding of a re-associated f32 sum (observed up to 2.4e-6), not bit for bit
        assert abs(a - b) <= 1e-5 * abs(b), (k, a, b)
    S.close()
