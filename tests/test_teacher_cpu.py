"""The teacher-forced block audit (tests/teacher.py) driven by a device made of ORACLE ops: checks the harness itself
(stage names, shapes, indexing, the tie proof) on the CPU, so that the GPU run of tests/test_teacher_forced_gpu.py
spends its minutes on the engine and not on the test code."""
import numpy as np
import pytest

from tests import teacher
from tools import synth


class OracleDevice:
    """run_block / stage / stage_q8 like deepseek.cpp_amd/dsk.py Model, computed with the oracle's op-level functions
    (Block::_block_cpu, src/infer.cpp:810-932, one quantize_row_q8_K_ref per staging point)."""

    def __init__(self, orc, c, T, perturb=0.0):
        self.orc, self.c, self.T = orc, c, T
        self.Q = {"q2_k": 3, "q3_k": 4}[c.quant]
        self.kc, self.vc, self.nc, self.rc = {}, {}, {}, {}
        self.perturb = perturb  # relative noise on the normed vectors: provokes int8 near-tie flips like a re-associated sum
        self.rng = np.random.default_rng(0)

    def w(self, l, name):
        return self.T[f"model.layers.{l}.{name}.weight"].data

    def q8(self, point, y, normed):
        y = np.ascontiguousarray(y, np.float32)
        if normed and self.perturb:
            y = (y * (1.0 + self.perturb * self.rng.standard_normal(y.size))).astype(np.float32)
        qs, d, _ = self.orc.q8k_quantize(y)
        self.q[point] = (qs, d)
        return qs, d

    def gemv(self, w, d, n, q):
        return self.orc.gemv_q8(self.Q, np.ascontiguousarray(w), d, n, q[0], q[1])

    def run_block(self, l, x_in, pos):
        c, orc = self.c, self.orc
        H, hd, nope, rope, vd, lora = c.n_heads, c.head_dim, c.qk_nope_head_dim, c.qk_rope_head_dim, c.v_head_dim, c.kv_lora_rank
        v3 = c.has_moegate_bias
        self.s, self.q = {}, {}
        S = self.s
        x = np.ascontiguousarray(x_in, np.float32)
        q1 = self.q8("x_attn", orc.rmsnorm(x, self.w(l, "attn.norm"), c.norm_eps), True)
        S["kv_a"] = self.gemv(self.w(l, "attn.wkv_a"), lora + rope, c.dim, q1)
        kv_len = pos + 1
        if c.q_lora_rank > 0:
            S["q_a"] = self.gemv(self.w(l, "attn.wq_a"), c.q_lora_rank, c.dim, q1)
            q2 = self.q8("q_a", orc.rmsnorm(S["q_a"], self.w(l, "attn.q_a_norm"), c.norm_eps), True)
        k_rope = orc.rope(S["kv_a"][lora:], rope, pos, c.rope_theta, v3)
        lat = orc.rmsnorm(S["kv_a"][:lora], self.w(l, "attn.kv_a_norm"), c.norm_eps)
        if c.use_mla:
            S["q_rope"] = self.gemv(self.w(l, "attn.wq_rope_b"), H * rope, c.q_lora_rank, q2)
            S["q_c"] = self.gemv(self.w(l, "attn.wc"), H * lora, c.q_lora_rank, q2)
            nc = self.nc.setdefault(l, np.zeros((c.max_seq_len, lora), np.uint16))
            rc = self.rc.setdefault(l, np.zeros((c.max_seq_len, rope), np.uint16))
            nc[pos] = lat.astype(np.float16).view(np.uint16)
            rc[pos] = k_rope.astype(np.float16).view(np.uint16)
            S["nope_cache"], S["rope_cache"] = nc[:kv_len].reshape(-1), rc[:kv_len].reshape(-1)
            qr = S["q_rope"].reshape(H, rope).copy()
            for h in range(H):
                qr[h] = orc.rope(qr[h], rope, pos, c.rope_theta, v3)
            S["latent_out"] = orc.attn_mla(S["q_c"], qr.reshape(-1), nc[:kv_len], rc[:kv_len], H, hd, lora, rope, kv_len)
            qL = np.zeros((H, lora), np.int8)
            dL = np.zeros((H, lora // 256), np.float32)
            vb = np.zeros(H * vd, np.float32)
            wv = self.w(l, "attn.wv_b")
            for h in range(H):
                qL[h], dL[h], _ = orc.q8k_quantize(S["latent_out"][h * lora:(h + 1) * lora])
                vb[h * vd:(h + 1) * vd] = orc.gemv_q8(self.Q, np.ascontiguousarray(wv[h * vd:(h + 1) * vd]), vd, lora, qL[h], dL[h])
            self.q["latent"] = (qL.reshape(-1), dL.reshape(-1))
            S["vb_out"] = vb
            att = vb
        else:
            q3 = self.q8("kv_a", lat, False)  # (perturbing here would also move the cache row the audit checks exactly)
            if c.q_lora_rank > 0:
                q = self.gemv(self.w(l, "attn.wq_b"), H * hd, c.q_lora_rank, q2)
            else:
                q = self.gemv(self.w(l, "attn.wq"), H * hd, c.dim, q1)
            kv_b = self.gemv(self.w(l, "attn.wkv_b"), H * (nope + vd), lora, q3).reshape(H, nope + vd)
            q = q.reshape(H, hd).copy()
            for h in range(H):
                q[h, nope:] = orc.rope(q[h, nope:], rope, pos, c.rope_theta, v3)
            k = np.concatenate([kv_b[:, :nope], np.broadcast_to(k_rope, (H, rope))], axis=1).astype(np.float32)
            kc = self.kc.setdefault(l, np.zeros((c.max_seq_len, H * hd), np.uint16))
            vc = self.vc.setdefault(l, np.zeros((c.max_seq_len, H * vd), np.uint16))
            kc[pos] = k.reshape(-1).astype(np.float16).view(np.uint16)
            vc[pos] = kv_b[:, nope:].reshape(-1).astype(np.float16).view(np.uint16)
            S["k_cache"], S["v_cache"] = kc[:kv_len].reshape(-1), vc[:kv_len].reshape(-1)
            S["att_out"] = orc.attn_mha(q.reshape(-1), kc[:kv_len], vc[:kv_len], H, hd, vd, kv_len)
            att = S["att_out"]
        q4 = self.q8("att", att, False)
        S["x_mid"] = (x + self.gemv(self.w(l, "attn.wo"), c.dim, H * vd, q4)).astype(np.float32)
        y5 = orc.rmsnorm(S["x_mid"], self.w(l, "mlp.norm"), c.norm_eps)
        act = teacher.silu if c.act == "silu" else teacher.gelu
        if not (c.n_routed_experts > 0 and l >= c.first_k_dense_replace):
            q5 = self.q8("x_ffn_tap", y5, True)
            S["hb"] = act(self.gemv(self.w(l, "mlp.w1"), c.hidden_dim, c.dim, q5)) * self.gemv(self.w(l, "mlp.w3"), c.hidden_dim, c.dim, q5)
            q6 = self.q8("hb", S["hb"], False)
            return (S["x_mid"] + self.gemv(self.w(l, "mlp.w2"), c.dim, c.hidden_dim, q6)).astype(np.float32)
        K, E, mi = c.n_active_routed, c.n_routed_experts, c.moe_intermediate_size
        shared_n = c.n_shared_experts * mi
        stride = max(mi, shared_n, 1)
        slots = K + (1 if c.n_shared_experts else 0)
        q5 = self.q8("x_ffn", y5, True)
        S["router_logits"] = orc.gemv(0, self.w(l, "moegate"), E, c.dim, y5)
        bias = self.T.get(f"model.layers.{l}.moegate.bias")
        e, w, _ = orc.moe_gate(S["router_logits"], None if bias is None else bias.data, K, c.norm_topk_prob, c.routed_scaling_factor,
                               1 if c.scoring_func == "sigmoid" else 0, 1 if c.topk_method == "group_limited_greedy" else 0,
                               c.n_group, c.topk_group)
        S["route_e"], S["route_w"] = e, w
        hb = np.zeros((slots, stride), np.float32)
        q6 = np.zeros((slots, stride), np.int8)
        d6 = np.zeros((slots, stride // 256), np.float32)
        eout = np.zeros((slots, c.dim), np.float32)
        xo = S["x_mid"].copy()
        for k in range(K):
            ek = int(e[k])
            hb[k, :mi] = act(self.gemv(self.w(l, "mlp.w1")[ek], mi, c.dim, q5)) * self.gemv(self.w(l, "mlp.w3")[ek], mi, c.dim, q5)
            q6[k, :mi], d6[k, :mi // 256], _ = orc.q8k_quantize(hb[k, :mi])
            eout[k] = orc.gemv_q8(self.Q, np.ascontiguousarray(self.w(l, "mlp.w2")[ek]), c.dim, mi, q6[k, :mi], d6[k, :mi // 256])
            xo = (xo + eout[k] * np.float32(w[k])).astype(np.float32)
        if c.n_shared_experts:
            hb[K, :shared_n] = act(self.gemv(self.w(l, "shared_mlp.w1"), shared_n, c.dim, q5)) * self.gemv(self.w(l, "shared_mlp.w3"), shared_n, c.dim, q5)
            q6[K, :shared_n], d6[K, :shared_n // 256], _ = orc.q8k_quantize(hb[K, :shared_n])
            eout[K] = orc.gemv_q8(self.Q, np.ascontiguousarray(self.w(l, "shared_mlp.w2")), c.dim, shared_n, q6[K, :shared_n], d6[K, :shared_n // 256])
            xo = (xo + eout[K]).astype(np.float32)
            self.q["x_ffn_shared"] = q5
        S["hb"], S["eout"] = hb.reshape(-1), eout.reshape(-1)
        self.q["hb"] = (q6.reshape(-1), d6.reshape(-1))
        return xo

    def run_head(self, x_in):
        c, orc = self.c, self.orc
        self.q = {}
        q = self.q8("x_final", orc.rmsnorm(x_in, self.T["model.norm.weight"].data, c.norm_eps), True)
        cls = self.T["model.output.weight"].data if "model.output.weight" in self.T else self.T["model.embed.weight"].data
        return orc.gemv_q8(self.Q, np.ascontiguousarray(cls), c.vocab_size, c.dim, q[0], q[1])

    def stage(self, name, n, dtype=np.float32):
        a = np.asarray(self.s[name]).reshape(-1)
        assert a.size >= n, (name, a.size, n)
        return a[:n].astype(dtype)

    def stage_q8(self, point, n):
        qs, d = self.q[point]
        return np.asarray(qs[:n], np.int8), np.asarray(d[:n // 256], np.float32)


CASES = [("tiny_v3", "q2_k", False), ("tiny_v3", "q2_k", True), ("tiny_v3", "q3_k", True), ("tiny_v2lite", "q2_k", False)]


@pytest.mark.parametrize("preset,quant,mla", CASES, ids=[f"{p}-{q}-{'mla' if m else 'mha'}" for p, q, m in CASES])
def test_audit_is_consistent_on_an_oracle_made_device(oracle, preset, quant, mla):
    c = synth.preset(preset, quant, mla)
    T = synth.synth_model(c, seed=21)
    dev = OracleDevice(oracle, c, T)
    aud = teacher.BlockAuditor(oracle, c, T)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(c.dim).astype(np.float32)
    for pos in range(3):  # the cache grows: attention over several positions
        for l in range(c.n_layers):
            A, x_out = aud.run(dev, l, x, pos)
            assert A.total_flips() == 0 and max(A.errs.values()) < 1e-6, A.summary()
            x = x_out if l + 1 < c.n_layers else rng.standard_normal(c.dim).astype(np.float32)
    A, _ = teacher.audit_head(oracle, c, T, dev, x)
    assert max(A.errs.values()) < 1e-6


def test_audit_counts_near_tie_flips_and_rejects_real_errors(oracle):
    """A device whose normed vectors carry ~1e-7 of relative noise (a re-associated rmsnorm) flips a few int8 roundings:
    every one must be proven a tie and the stage outputs still agree (the device's codes are injected).  A device
    with a WRONG code that is not a tie must be rejected."""
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=22)
    aud = teacher.BlockAuditor(oracle, c, T)
    rng = np.random.default_rng(2)
    class TieDevice(OracleDevice):
        """rounds the element closest to a rounding tie the OTHER way when it is within 1e-4 of the tie -- what a
        re-associated rmsnorm does to such an element"""
        forced = 0

        def q8(self, point, y, normed):
            qs, d = super().q8(point, y, normed)
            if not normed:
                return qs, d
            y = np.ascontiguousarray(y, np.float32)
            yb = y.reshape(-1, 256)
            mx = yb[np.arange(yb.shape[0]), np.argmax(np.abs(yb), axis=1)]
            v = (np.float32(-127.0) / mx).astype(np.float32)[np.arange(y.size) // 256] * y
            dist = np.abs(np.abs(v - np.floor(v)) - 0.5)
            i = int(np.argmin(dist))
            if dist[i] < 1e-4 and abs(int(qs[i])) < 126:
                lo = int(np.floor(v[i]))
                qs = qs.copy()
                qs[i] = lo + 1 if int(qs[i]) == lo else lo
                self.q[point] = (qs, d)
                TieDevice.forced += 1
            return qs, d

    flips = 0
    dev = TieDevice(oracle, c, T)
    for trial in range(40):
        x = (rng.standard_normal(c.dim) * rng.uniform(0.5, 4)).astype(np.float32)
        for l in range(c.n_layers):
            A, _ = aud.run(dev, l, x, 0)
            flips += A.total_flips()
        if flips >= 2:
            break
    assert flips > 0 and flips == TieDevice.forced, (flips, TieDevice.forced)

    class Broken(OracleDevice):
        def q8(self, point, y, normed):
            qs, d = super().q8(point, y, normed)
            if point == "x_ffn":
                i = int(np.argmin(np.abs(qs.astype(np.int32) - 40)))
                qs = qs.copy()
                qs[i] += 1  # not a tie
                self.q[point] = (qs, d)
            return qs, d

    with pytest.raises(AssertionError):
        aud.run(Broken(oracle, c, T), 1, rng.standard_normal(c.dim).astype(np.float32), 0)
