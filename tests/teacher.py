"""Teacher-forced, flip-audited parity of ONE block (test infrastructure).

North star: "top-k expert indices identical, logits within 1e-3 relative".  A free-running W2A8 / W3A8 model cannot be
held to that at full width: every activation vector is quantised to int8 (quantize_row_q8_K_ref, src/quant.cpp:616-653),
a 1e-7 difference in an input flips a rounding that sits on a tie, the flip moves the next vector by ~4e-4, and the
difference compounds (the reference shows the same against any re-associated copy of itself).  So the block is verified
the way a discontinuous function has to be: piece by piece, each piece on the DEVICE'S OWN inputs.

For every Q8_K staging point of the block the device hands over (dsk_model_run_block / dsk_model_get_stage) the float
vector it quantised from and the int8 codes + block scales it actually staged.  The audit then proves, stage by stage:

  (1) codes: the oracle quantises the same float vector; every code that differs must be a PROVEN NEAR-TIE
      (|frac(iscale * y) - 0.5| < TIE_TOL: the two roundings are both legitimate for inputs that differ in the
      last bits), block scales agree to a few ulp.  Points without a norm in front (attention output, hidden vectors,
      MLA latent) must agree bit for bit.  The flips are counted and reported.
  (2) arithmetic: with the device's codes injected into the oracle's integer GEMV (orc_gemv_q8), the device's output
      of the stage must match to FLOAT tolerance (2e-5: only the association of the f32 super-block sums differs).
  (3) routing: the oracle's moe_gate on the device's router logits must select IDENTICAL experts in identical order
      (unless two scores are closer than 1e-6: a proven tie), weights to 1e-5.

By induction over the stages the block equals the reference's block up to proven near-ties of int8 roundings, and the
block output given the injected codes is within 1e-5 -- far inside the north star's 1e-3.

`Device` is anything with run_block / stage / stage_q8 (deepseek.cpp_amd/dsk.py Model; tests/test_teacher_cpu.py drives
the same audit with a device made of oracle ops, which checks the harness itself without a GPU).
"""
from __future__ import annotations

import numpy as np

TIE_TOL = 2e-4      # |frac - 0.5| of iscale * y for a code that may legitimately round either way
GEMV_TOL = 2e-5     # rel_inf of an integer GEMV output given identical codes (f32 association only)
FLOAT_TOL = 1e-4    # rel_inf of float stages (attention, softmax, expf)
Q2K, Q3K, F32 = 3, 4, 0


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def ulp_diff_f32(a, b):
    """distance in units of the last place between two float32 arrays (same sign assumed where it matters)"""
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def f16_ulp_diff(a_bits, b_bits):
    ia = np.asarray(a_bits, np.uint16).astype(np.int64)
    ib = np.asarray(b_bits, np.uint16).astype(np.int64)
    ia = np.where(ia & 0x8000, -(ia & 0x7fff), ia)
    ib = np.where(ib & 0x8000, -(ib & 0x7fff), ib)
    return np.abs(ia - ib)


def f16_row_diff(row_bits, ref_f32):
    """last-place distance between a cached f16 row and the f16 rounding (RNE, src/codec.h:26-27) of the expected f32
    values; an element whose ABSOLUTE difference is below 1e-6 of the row's scale counts as equal (a rotated pair
    v0*c - v1*s that cancels is tiny and carries the f32 rounding of its terms: any number of f16 places, no information)"""
    ref_f32 = np.asarray(ref_f32, np.float32)
    d = f16_ulp_diff(row_bits, ref_f32.astype(np.float16).view(np.uint16))
    absd = np.abs(np.asarray(row_bits, np.uint16).view(np.float16).astype(np.float64) - ref_f32.astype(np.float16).astype(np.float64))
    return np.where(absd <= 1e-6 * max(float(np.max(np.abs(ref_f32))), 1e-30), 0, d)


class Audit:
    """Collects the evidence of one block."""

    def __init__(self):
        self.errs = {}     # stage -> rel_inf
        self.flips = {}    # point -> (codes that differ, all of them proven near-ties; largest tie distance)
        self.notes = []

    def chk(self, name, got, ref, tol):
        e = rel_inf(got, ref)
        self.errs[name] = e
        assert np.all(np.isfinite(got)), name
        assert e < tol, (name, e, tol)

    def total_flips(self):
        return sum(v[0] for v in self.flips.values())

    def summary(self):
        worst = max(self.errs.items(), key=lambda kv: kv[1]) if self.errs else ("-", 0.0)
        return (f"{len(self.errs)} stages, worst {worst[0]} {worst[1]:.2e}; "
                f"{self.total_flips()} int8 near-tie flips over {len(self.flips)} Q8_K points")


def audit_codes(A: Audit, orc, point, y, qh, dh, exact):
    """y: the float vector the device quantised (for norm-fused points: the ORACLE's rmsnorm of the device's input).
    qh / dh: the device's codes and block scales.  exact: no float stage in front -> bit equality required."""
    y = np.ascontiguousarray(y, np.float32)
    n = y.size
    qo, do, _ = orc.q8k_quantize(y)
    qh = np.asarray(qh, np.int8)
    dh = np.asarray(dh, np.float32)
    assert qh.size == n and dh.size == n // 256, (point, qh.size, dh.size, n)
    assert not np.any(np.isnan(dh)), (point, "tap did not fire (poisoned block scales)")
    if exact:
        assert np.array_equal(dh.view(np.uint32), do.view(np.uint32)), (point, "block scales differ on identical inputs")
        assert np.array_equal(qh, qo), (point, "codes differ on identical inputs", int(np.sum(qh != qo)))
        A.flips[point] = (0, 0.0)
        return
    # a norm in front: the sum of squares over the vector is associated differently (sequential in the oracle, a fixed
    # tree on the device; the reference's own is an auto-vectorised 8-lane sum): a common factor of 1 +- O(sqrt(n) eps)
    # on the whole vector, which moves every block scale alike and no code (codes depend on ratios within the block)
    rel = np.abs(dh.astype(np.float64) - do) / np.maximum(np.abs(do.astype(np.float64)), 1e-30)
    assert np.max(rel) <= 1e-5, (point, "block scales", float(np.max(rel)))
    bad = np.nonzero(qh != qo)[0]
    worst = 0.0
    if bad.size:
        assert np.max(np.abs(qh[bad].astype(np.int32) - qo[bad].astype(np.int32))) == 1, (point, "a code differs by more than one step")
        # prove each one a near-tie: iscale * y within TIE_TOL of a half-integer (oracle arithmetic, src/quant.cpp:630-640)
        yb = y.reshape(-1, 256)
        idx = np.argmax(np.abs(yb), axis=1)
        mx = yb[np.arange(yb.shape[0]), idx]
        iscale = np.where(mx != 0, np.float32(-127.0) / np.where(mx != 0, mx, 1).astype(np.float32), 0).astype(np.float32)
        v = (iscale[bad // 256] * y[bad]).astype(np.float64)
        dist = np.abs(np.abs(v - np.floor(v)) - 0.5)
        worst = float(np.max(dist))
        assert worst < TIE_TOL, (point, "a differing code is not a rounding tie", worst, int(bad.size))
    assert bad.size <= max(4, n // 500), (point, "too many flips to be ties", int(bad.size), n)
    A.flips[point] = (int(bad.size), worst)


def silu(x):
    x = np.asarray(x, np.float32)
    return (x / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def gelu(x):
    x = np.asarray(x, np.float32)
    return (np.float32(0.5) * x * (np.float32(1.0) + np.tanh(np.float32(0.797885) * (x + np.float32(0.044715) * x * x * x)))).astype(np.float32)


class BlockAuditor:
    def __init__(self, orc, c, T):
        self.orc, self.c, self.T = orc, c, T
        self.Q = {"q2_k": Q2K, "q3_k": Q3K}[c.quant]

    def w(self, layer, name):
        return self.T[f"model.layers.{layer}.{name}.weight"].data

    def gemv(self, w, d, n, q, dq):
        return self.orc.gemv_q8(self.Q, np.ascontiguousarray(w), d, n, q, dq)

    # ---------------------------------------------------------------- attention half
    def _attention_mha(self, A, dev, l, x_in, pos, kv_len, kv_pos):
        c, orc = self.c, self.orc
        H, hd, nope, rope, vd, lora = c.n_heads, c.head_dim, c.qk_nope_head_dim, c.qk_rope_head_dim, c.v_head_dim, c.kv_lora_rank
        v3 = c.has_moegate_bias
        q1, d1 = dev.stage_q8("x_attn", c.dim)
        audit_codes(A, orc, "x_attn", orc.rmsnorm(x_in, self.w(l, "attn.norm"), c.norm_eps), q1, d1, False)
        kv_a = dev.stage("kv_a", lora + rope)
        A.chk("kv_a", kv_a, self.gemv(self.w(l, "attn.wkv_a"), lora + rope, c.dim, q1, d1), GEMV_TOL)
        q3, d3 = dev.stage_q8("kv_a", lora)
        audit_codes(A, orc, "kv_a", orc.rmsnorm(kv_a[:lora], self.w(l, "attn.kv_a_norm"), c.norm_eps), q3, d3, False)
        if c.q_lora_rank > 0:
            q_a = dev.stage("q_a", c.q_lora_rank)
            A.chk("q_a", q_a, self.gemv(self.w(l, "attn.wq_a"), c.q_lora_rank, c.dim, q1, d1), GEMV_TOL)
            q2, d2 = dev.stage_q8("q_a", c.q_lora_rank)
            audit_codes(A, orc, "q_a", orc.rmsnorm(q_a, self.w(l, "attn.q_a_norm"), c.norm_eps), q2, d2, False)
            q = self.gemv(self.w(l, "attn.wq_b"), H * hd, c.q_lora_rank, q2, d2)
        else:
            q = self.gemv(self.w(l, "attn.wq"), H * hd, c.dim, q1, d1)
        kv_b = self.gemv(self.w(l, "attn.wkv_b"), H * (nope + vd), lora, q3, d3)
        q = q.reshape(H, hd).copy()
        for h in range(H):  # src/infer.cpp:956-960
            q[h, nope:] = orc.rope(q[h, nope:], rope, pos, c.rope_theta, v3)
        k_rope = orc.rope(kv_a[lora:], rope, pos, c.rope_theta, v3)
        kvb = kv_b.reshape(H, nope + vd)
        k = np.concatenate([kvb[:, :nope], np.broadcast_to(k_rope, (H, rope))], axis=1).astype(np.float32)
        v = kvb[:, nope:].astype(np.float32)
        kc = dev.stage("k_cache", kv_len * H * hd, np.uint16).reshape(kv_len, H * hd)
        vc = dev.stage("v_cache", kv_len * H * vd, np.uint16).reshape(kv_len, H * vd)
        # this position's cache row: f32 -> f16 RNE of values that agree to ~1e-6 may differ in the last f16 place
        dk = f16_row_diff(kc[kv_pos], k.reshape(-1))
        dv = f16_row_diff(vc[kv_pos], v.reshape(-1))
        assert dk.max() <= 1 and dv.max() <= 1, ("cache row", int(dk.max()), int(dv.max()))
        assert (dk > 0).mean() < 0.02 and (dv > 0).mean() < 0.02, ("cache row: too many last-place differences", float((dk > 0).mean()))
        A.notes.append(f"cache row: {float(max((dk > 0).mean(), (dv > 0).mean())):.4f} of the f16 values differ in the last place")
        att = dev.stage("att_out", H * vd)
        A.chk("att_out", att, orc.attn_mha(q.reshape(-1), kc, vc, H, hd, vd, kv_len), FLOAT_TOL)
        return att

    def _attention_mla(self, A, dev, l, x_in, pos, kv_len, kv_pos):
        c, orc = self.c, self.orc
        H, rope, vd, lora = c.n_heads, c.qk_rope_head_dim, c.v_head_dim, c.kv_lora_rank
        v3 = c.has_moegate_bias
        q1, d1 = dev.stage_q8("x_attn", c.dim)
        audit_codes(A, orc, "x_attn", orc.rmsnorm(x_in, self.w(l, "attn.norm"), c.norm_eps), q1, d1, False)
        q_a = dev.stage("q_a", c.q_lora_rank)
        kv_a = dev.stage("kv_a", lora + rope)
        A.chk("q_a", q_a, self.gemv(self.w(l, "attn.wq_a"), c.q_lora_rank, c.dim, q1, d1), GEMV_TOL)
        A.chk("kv_a", kv_a, self.gemv(self.w(l, "attn.wkv_a"), lora + rope, c.dim, q1, d1), GEMV_TOL)
        q2, d2 = dev.stage_q8("q_a", c.q_lora_rank)
        audit_codes(A, orc, "q_a", orc.rmsnorm(q_a, self.w(l, "attn.q_a_norm"), c.norm_eps), q2, d2, False)
        q_rope = dev.stage("q_rope", H * rope)
        q_c = dev.stage("q_c", H * lora)
        A.chk("q_rope", q_rope, self.gemv(self.w(l, "attn.wq_rope_b"), H * rope, c.q_lora_rank, q2, d2), GEMV_TOL)
        A.chk("q_c", q_c, self.gemv(self.w(l, "attn.wc"), H * lora, c.q_lora_rank, q2, d2), GEMV_TOL)
        # this position's latent cache entries (src/infer.cpp:1089-1097)
        nc = dev.stage("nope_cache", kv_len * lora, np.uint16).reshape(kv_len, lora)
        rc = dev.stage("rope_cache", kv_len * rope, np.uint16).reshape(kv_len, rope)
        lat = orc.rmsnorm(kv_a[:lora], self.w(l, "attn.kv_a_norm"), c.norm_eps)
        k_rope = orc.rope(kv_a[lora:], rope, pos, c.rope_theta, v3)
        dn = f16_row_diff(nc[kv_pos], lat)
        dr = f16_row_diff(rc[kv_pos], k_rope)
        assert dn.max() <= 1 and dr.max() <= 1, ("latent cache row", int(dn.max()), int(dr.max()))
        assert (dn > 0).mean() < 0.05, float((dn > 0).mean())
        qr = q_rope.reshape(H, rope).copy()
        for h in range(H):
            qr[h] = orc.rope(qr[h], rope, pos, c.rope_theta, v3)
        latent = dev.stage("latent_out", H * lora)
        A.chk("latent_out", latent, orc.attn_mla(q_c, qr.reshape(-1), nc, rc, H, c.head_dim, lora, rope, kv_len), FLOAT_TOL)
        qL, dL = dev.stage_q8("latent", H * lora)
        wv = self.w(l, "attn.wv_b")  # (H * v, lora) viewed as H stacked (v, lora) matrices, src/infer.cpp:1134-1137
        vb_ref = np.zeros(H * vd, np.float32)
        qL, dL, latent2 = qL.reshape(H, lora), dL.reshape(H, lora // 256), latent.reshape(H, lora)
        for h in range(H):
            audit_codes(A, orc, f"latent[{h}]", latent2[h], qL[h], dL[h], True)
            vb_ref[h * vd:(h + 1) * vd] = self.gemv(wv[h * vd:(h + 1) * vd], vd, lora, qL[h], dL[h])
        flips = {k: v for k, v in A.flips.items() if k.startswith("latent[")}
        for k in flips:
            del A.flips[k]
        A.flips["latent"] = (sum(v[0] for v in flips.values()), 0.0)
        vb = dev.stage("vb_out", H * vd)
        A.chk("vb_out", vb, vb_ref, GEMV_TOL)
        return vb

    # ---------------------------------------------------------------- the block
    def run(self, dev, l, x_in, pos):
        c, orc = self.c, self.orc
        x_in = np.ascontiguousarray(x_in, np.float32)
        # KV ring with KV_SINKS = 2 attention sinks (src/infer.cpp:1271-1277, src/model.h:14).  Past the ring's length the
        # device re-rotates the two sink keys in place before attending; the audit then checks attention over the cache rows
        # AS THE DEVICE LEFT THEM and this position's row (the sink rotation itself: tests/test_model_gpu.py, small models).
        W = c.rs_original_max_position_embeddings
        sink = 2 if pos >= W else 0
        kv_pos = sink + (pos - sink) % (W - sink)
        kv_len = W if pos >= W else pos + 1
        A = Audit()
        x_out = dev.run_block(l, x_in, pos)
        H, vd = c.n_heads, c.v_head_dim
        att = (self._attention_mla if c.use_mla else self._attention_mha)(A, dev, l, x_in, pos, kv_len, kv_pos)
        # attention output -> Q8_K (no norm in front: bit-exact) -> wo -> residual (src/infer.cpp:1048, 832-834)
        q4, d4 = dev.stage_q8("att", H * vd)
        audit_codes(A, orc, "att", att, q4, d4, True)
        x_mid = dev.stage("x_mid", c.dim)
        A.chk("x_mid", x_mid, x_in + self.gemv(self.w(l, "attn.wo"), c.dim, H * vd, q4, d4), GEMV_TOL)
        moe = c.n_routed_experts > 0 and l >= c.first_k_dense_replace
        y5 = orc.rmsnorm(x_mid, self.w(l, "mlp.norm"), c.norm_eps)
        act = silu if c.act == "silu" else gelu
        if not moe:
            q5, d5 = dev.stage_q8("x_ffn_tap", c.dim)
            audit_codes(A, orc, "x_ffn", y5, q5, d5, False)
            hb = dev.stage("hb", c.hidden_dim)
            h1 = self.gemv(self.w(l, "mlp.w1"), c.hidden_dim, c.dim, q5, d5)
            h3 = self.gemv(self.w(l, "mlp.w3"), c.hidden_dim, c.dim, q5, d5)
            A.chk("hb", hb, act(h1) * h3, GEMV_TOL)
            q6, d6 = dev.stage_q8("hb", c.hidden_dim)
            audit_codes(A, orc, "hb", hb, q6, d6, True)
            A.chk("x_out", x_out, x_mid + self.gemv(self.w(l, "mlp.w2"), c.dim, c.hidden_dim, q6, d6), GEMV_TOL)
            return A, x_out
        K, E, mi = c.n_active_routed, c.n_routed_experts, c.moe_intermediate_size
        shared_n = c.n_shared_experts * mi
        stride = max(mi, shared_n, 1)
        slots = K + (1 if c.n_shared_experts > 0 else 0)
        q5, d5 = dev.stage_q8("x_ffn", c.dim)
        audit_codes(A, orc, "x_ffn", y5, q5, d5, False)
        # router: F32 GEMV (src/infer.cpp:847) on the normed x, then moe_gate on the DEVICE's logits
        logits = dev.stage("router_logits", E)
        A.chk("router_logits", logits, orc.gemv(F32, self.w(l, "moegate"), E, c.dim, y5), GEMV_TOL)
        bias = self.T.get(f"model.layers.{l}.moegate.bias")
        e_ref, w_ref, scores = orc.moe_gate(logits, None if bias is None else bias.data, K, c.norm_topk_prob, c.routed_scaling_factor,
                                            1 if c.scoring_func == "sigmoid" else 0, 1 if c.topk_method == "group_limited_greedy" else 0,
                                            c.n_group, c.topk_group)
        e_dev, w_dev = dev.stage("route_e", K, np.int32), dev.stage("route_w", K)
        if not np.array_equal(e_dev, e_ref):  # only a proven tie of two scores may explain it
            diff = np.nonzero(e_dev != e_ref)[0]
            gaps = [abs(float(scores[e_dev[k]]) - float(scores[e_ref[k]])) for k in diff]
            assert max(gaps) < 1e-6, ("expert indices differ without a score tie", e_dev, e_ref, gaps)
            A.notes.append(f"routing tie at slots {diff.tolist()} (gaps {gaps})")
            A.errs["route_tie_gap"] = max(gaps)
        else:
            assert np.max(np.abs(w_dev - w_ref)) <= 1e-5 * max(1.0, float(np.max(np.abs(w_ref)))), (w_dev, w_ref)
            A.errs["route_w"] = float(np.max(np.abs(w_dev - w_ref)))
        # experts: GLU on the staged x (src/infer.cpp:853-872), hidden -> Q8_K (bit-exact) -> w2 (:873)
        hb = dev.stage("hb", slots * stride).reshape(slots, stride)
        q6, d6 = dev.stage_q8("hb", slots * stride)
        q6, d6 = q6.reshape(slots, stride), d6.reshape(slots, stride // 256)
        eout = dev.stage("eout", slots * c.dim).reshape(slots, c.dim)
        w1, w2, w3 = self.w(l, "mlp.w1"), self.w(l, "mlp.w2"), self.w(l, "mlp.w3")
        for k in range(K):
            e = int(e_dev[k])
            assert 0 <= e < E
            h1, h3 = self.gemv(w1[e], mi, c.dim, q5, d5), self.gemv(w3[e], mi, c.dim, q5, d5)
            A.chk(f"hb[{k}]", hb[k, :mi], act(h1) * h3, GEMV_TOL)
            audit_codes(A, orc, f"hb[{k}]", hb[k, :mi], q6[k, :mi], d6[k, :mi // 256], True)
            A.chk(f"eout[{k}]", eout[k], self.gemv(w2[e], c.dim, mi, q6[k, :mi], d6[k, :mi // 256]), GEMV_TOL)
        x_ref = x_mid.copy()
        for k in range(K):  # x += w_k * out_k in k order (src/infer.cpp:874-877)
            x_ref = (x_ref + eout[k] * np.float32(w_dev[k])).astype(np.float32)
        if c.n_shared_experts > 0:
            # the shared expert quantises the same normed x (its launch-mates derive it on their own: must be the same codes)
            try:
                qs_, ds_ = dev.stage_q8("x_ffn_shared", c.dim)
                if not np.isnan(ds_).any():
                    assert np.array_equal(qs_, q5) and np.array_equal(ds_.view(np.uint32), d5.view(np.uint32)), "shared expert staged other codes"
            except KeyError:
                pass
            h1 = self.gemv(self.w(l, "shared_mlp.w1"), shared_n, c.dim, q5, d5)
            h3 = self.gemv(self.w(l, "shared_mlp.w3"), shared_n, c.dim, q5, d5)
            A.chk("hb[shared]", hb[K, :shared_n], act(h1) * h3, GEMV_TOL)
            audit_codes(A, orc, "hb[shared]", hb[K, :shared_n], q6[K, :shared_n], d6[K, :shared_n // 256], True)
            A.chk("eout[shared]", eout[K], self.gemv(self.w(l, "shared_mlp.w2"), c.dim, shared_n, q6[K, :shared_n], d6[K, :shared_n // 256]), GEMV_TOL)
            x_ref = (x_ref + eout[K]).astype(np.float32)  # src/infer.cpp:900-903
        flips = {k: v for k, v in A.flips.items() if k.startswith("hb[")}
        for k in flips:
            del A.flips[k]
        A.flips["hb"] = (sum(v[0] for v in flips.values()), 0.0)
        A.chk("x_out", x_out, x_ref, 2e-6)
        return A, x_out


def audit_head(orc, c, T, dev, x_in, rows=None):
    """final norm -> Q8_K -> classifier (src/infer.cpp:1292-1316) on sampled rows of the vocabulary"""
    A = Audit()
    Q = {"q2_k": Q2K, "q3_k": Q3K}[c.quant]
    logits = dev.run_head(x_in)
    q, d = dev.stage_q8("x_final", c.dim)
    audit_codes(A, orc, "x_final", orc.rmsnorm(x_in, T["model.norm.weight"].data, c.norm_eps), q, d, False)
    cls = T["model.output.weight"].data if "model.output.weight" in T else T["model.embed.weight"].data
    if rows is None:
        rows = np.arange(c.vocab_size)
    ref = orc.gemv_q8(Q, np.ascontiguousarray(cls[rows]), len(rows), c.dim, q, d)
    A.chk("logits", logits[rows], ref, GEMV_TOL)
    return A, logits


class HydrateDevice:
    """Token `index` of the LAST batched chunk of dsk_hydrate as a `Device` of the audit above (include/dsk.h
    dsk_hydrate_get_buffer with option "hydrate_tap_layer" = layer, dsk_hydrate_get_trace_x, dsk_model_get_cache_rows).

    The batched prompt path (src/main.cpp:312-319 run as GEMMs) has no per-block entry point: the chunk runs every block on
    its own residual stream.  The audit needs nothing else - it proves each stage on the DEVICE'S OWN inputs - so run_block
    here only hands back what the chunk computed: x_in must be the chunk's stream in front of the block (the caller takes
    it from the trace, or the embedding row for block 0) and the return value is the stream after it."""

    def __init__(self, M, c, layer, index):
        self.M, self.c, self.layer, self.i = M, c, layer, index
        self.moe = c.n_routed_experts > 0 and layer >= c.first_k_dense_replace

    def _buf(self, name, width, dtype=np.float32):
        return self.M.hydrate_buffer(name, self.i, 1, width, dtype)[0]

    def run_block(self, l, x_in, pos):
        assert l == self.layer
        return self.M.hydrate_trace_x(l, self.i)

    def stage(self, name, n, dtype=np.float32):
        c, M, l = self.c, self.M, self.layer
        H = c.n_heads
        widths = {"k_cache": H * c.head_dim, "v_cache": H * c.v_head_dim, "nope_cache": c.kv_lora_rank, "rope_cache": c.qk_rope_head_dim}
        if name in widths:
            w = widths[name]
            assert n % w == 0
            return M.get_cache_rows(l, name, 0, n // w, w).reshape(-1)
        if self.moe and name in ("hb", "eout"):
            K, mi = c.n_active_routed, c.moe_intermediate_size
            shn = c.n_shared_experts * mi
            slots = K + (1 if shn > 0 else 0)
            if name == "eout":
                out = np.zeros((slots, c.dim), np.float32)
                out[:K] = self._buf("eout", K * c.dim).reshape(K, c.dim)
                if shn > 0:
                    out[K] = self._buf("eout_sh", c.dim)
            else:
                stride = max(mi, shn, 1)
                out = np.zeros((slots, stride), np.float32)
                out[:K, :mi] = self._buf("hb", K * mi).reshape(K, mi)
                if shn > 0:
                    out[K, :shn] = self._buf("hb_sh", shn)
            assert out.size == n, (name, out.size, n)
            return out.reshape(-1)
        return self._buf(name, n, dtype)

    def stage_q8(self, point, n):
        c = self.c
        if point == "x_ffn_shared":
            raise KeyError(point)  # (the chunk quantises the normed x once: there is no second copy to compare)
        if point == "x_ffn_tap":
            point = "x_ffn"
        if self.moe and point == "hb":
            K, mi = c.n_active_routed, c.moe_intermediate_size
            shn = c.n_shared_experts * mi
            slots, stride = K + (1 if shn > 0 else 0), max(mi, shn, 1)
            q = np.zeros((slots, stride), np.int8)
            d = np.zeros((slots, stride // 256), np.float32)
            q[:K, :mi] = self._buf("q8.hb.qs", K * mi, np.int8).reshape(K, mi)
            d[:K, :mi // 256] = self._buf("q8.hb.d", K * mi // 256).reshape(K, mi // 256)
            if shn > 0:
                q[K, :shn] = self._buf("q8.hb_sh.qs", shn, np.int8)
                d[K, :shn // 256] = self._buf("q8.hb_sh.d", shn // 256)
            assert q.size == n
            return q.reshape(-1), d.reshape(-1)
        return self._buf(f"q8.{point}.qs", n, np.int8), self._buf(f"q8.{point}.d", n // 256)
