"""ctypes bindings for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

  Oracle : oracle/liborc.so            (plain-C restatement, oracle/dsk_oracle.c)
  Ref    : oracle/_ref/libdskref.so    (the unmodified reference behind oracle/ref_shim.cpp)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORC_SO = os.path.join(HERE, "liborc.so")
REF_SO = os.path.join(HERE, "_ref", "libdskref.so")

c_f = C.POINTER(C.c_float)
c_i32 = C.POINTER(C.c_int32)
c_i16 = C.POINTER(C.c_int16)
c_i8 = C.POINTER(C.c_int8)
c_u16 = C.POINTER(C.c_uint16)


def build(ref: bool = True):
    """make -C oracle (the reference part only where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all" if ref else "orc"])


def fp(a: np.ndarray, t=c_f):
    return a.ctypes.data_as(t)


def vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DskConfig(C.Structure):
    """dsk_config, include/dsk.h"""
    _fields_ = [
        ("dim", C.c_int32), ("hidden_dim", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
        ("vocab_size", C.c_int32), ("max_seq_len", C.c_int32), ("rope_theta", C.c_float),
        ("norm_eps", C.c_float), ("act", C.c_int32), ("first_k_dense_replace", C.c_int32),
        ("n_shared_experts", C.c_int32), ("n_routed_experts", C.c_int32), ("n_active_routed", C.c_int32),
        ("moe_intermediate_size", C.c_int32), ("routed_scaling_factor", C.c_float), ("n_group", C.c_int32),
        ("norm_topk_prob", C.c_int32), ("scoring_func", C.c_int32), ("topk_group", C.c_int32),
        ("topk_method", C.c_int32), ("has_moegate_bias", C.c_int32), ("use_mla", C.c_int32),
        ("kv_lora_rank", C.c_int32), ("q_lora_rank", C.c_int32), ("qk_nope_head_dim", C.c_int32),
        ("qk_rope_head_dim", C.c_int32), ("v_head_dim", C.c_int32), ("weight_quant", C.c_int32),
        ("block_size", C.c_int32 * 2), ("rs_original_max_position_embeddings", C.c_int32),
    ]


QUANT_IDS = {"fp32": 0, "fp16": 1, "f8e5m2": 2, "q2_k": 3, "q3_k": 4}


def to_dsk_config(c) -> DskConfig:
    """tools.synth.Cfg -> dsk_config"""
    d = DskConfig()
    d.dim, d.hidden_dim, d.n_layers, d.n_heads = c.dim, c.hidden_dim, c.n_layers, c.n_heads
    d.vocab_size, d.max_seq_len = c.vocab_size, c.max_seq_len
    d.rope_theta, d.norm_eps = c.rope_theta, c.norm_eps
    d.act = 1 if c.act == "silu" else 0
    d.first_k_dense_replace = c.first_k_dense_replace
    d.n_shared_experts, d.n_routed_experts = c.n_shared_experts, c.n_routed_experts
    d.n_active_routed, d.moe_intermediate_size = c.n_active_routed, c.moe_intermediate_size
    d.routed_scaling_factor, d.n_group = c.routed_scaling_factor, c.n_group
    d.norm_topk_prob = int(c.norm_topk_prob)
    d.scoring_func = 1 if c.scoring_func == "sigmoid" else 0
    d.topk_group = c.topk_group
    d.topk_method = 1 if c.topk_method == "group_limited_greedy" else 0
    d.has_moegate_bias = int(c.has_moegate_bias)
    d.use_mla = int(c.use_mla)
    d.kv_lora_rank, d.q_lora_rank = c.kv_lora_rank, c.q_lora_rank
    d.qk_nope_head_dim, d.qk_rope_head_dim, d.v_head_dim = c.qk_nope_head_dim, c.qk_rope_head_dim, c.v_head_dim
    d.weight_quant = QUANT_IDS[c.quant]
    d.block_size[0], d.block_size[1] = c.block_size
    d.rs_original_max_position_embeddings = c.rs_original_max_position_embeddings
    return d


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)


class Oracle(_Lib):
    def __init__(self):
        super().__init__(ORC_SO)
        L = self.lib
        L.orc_half_to_float.restype = C.c_float
        L.orc_half_to_float.argtypes = [C.c_uint16]
        L.orc_float_to_half.restype = C.c_uint16
        L.orc_float_to_half.argtypes = [C.c_float]
        L.orc_f8e5m2_to_float.restype = C.c_float
        L.orc_f8e5m2_to_float.argtypes = [C.c_uint8]
        L.orc_float_to_f8e5m2.restype = C.c_uint8
        L.orc_float_to_f8e5m2.argtypes = [C.c_float]
        L.orc_last_error.restype = C.c_char_p
        L.orc_moe_gate.argtypes = [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                   C.c_int, c_i32, c_f, c_f]
        L.orc_rope.argtypes = [c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.orc_rope_f16.argtypes = [c_u16, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.orc_rmsnorm.argtypes = [c_f, c_f, c_f, C.c_int, C.c_float]
        L.orc_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f]
        L.orc_model_bind.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_i32, C.c_void_p, C.c_size_t]

    def err(self):
        return self.lib.orc_last_error().decode()

    # ---- ops
    def q8k_quantize(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n = x.size
        qs, d, bs = np.zeros(n, np.int8), np.zeros(n // 256, np.float32), np.zeros(n // 16, np.int16)
        self.lib.orc_q8k_quantize(fp(x), n, fp(qs, c_i8), fp(d), fp(bs, c_i16))
        return qs, d, bs

    def gemv(self, quant, w, d, n, x, scale=None, block_size=(0, 0)):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(d, np.float32)
        bsz = (C.c_int32 * 2)(*block_size)
        sc = None if scale is None else fp(np.ascontiguousarray(scale, np.float32))
        r = self.lib.orc_gemv(quant, vp(w), sc, bsz, d, n, fp(x), fp(out))
        if r:
            raise RuntimeError(self.err())
        return out

    def gemv_q8(self, quant, w, d, n, qs, yd):
        """K-quant GEMV on a GIVEN Q8_K vector (int8 codes + per-block scales): teacher forcing at a staging point."""
        qs = np.ascontiguousarray(qs, np.int8)
        yd = np.ascontiguousarray(yd, np.float32)
        assert qs.size == n and yd.size == n // 256
        out = np.zeros(d, np.float32)
        f = self.lib.orc_gemv_q8
        f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, c_i8, c_f, c_f]
        if f(quant, vp(w), d, n, fp(qs, c_i8), fp(yd), fp(out)):
            raise RuntimeError(self.err())
        return out

    def gemv_q2k_tiles(self, w, d, n, qs, yd):
        """Q2_K GEMV on a GIVEN Q8_K vector with the f32 association of the device's tiled kernels (orc_gemv_q2k_tiles)."""
        qs = np.ascontiguousarray(qs, np.int8)
        yd = np.ascontiguousarray(yd, np.float32)
        assert qs.size == n and yd.size == n // 256
        out = np.zeros(d, np.float32)
        f = self.lib.orc_gemv_q2k_tiles
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, c_i8, c_f, c_f]
        if f(vp(w), d, n, fp(qs, c_i8), fp(yd), fp(out)):
            raise RuntimeError(self.err())
        return out

    def gemv_expert(self, quant, w, expert, d, n, x, scale=None, block_size=(0, 0)):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(d, np.float32)
        bsz = (C.c_int32 * 2)(*block_size)
        sc = None if scale is None else fp(np.ascontiguousarray(scale, np.float32))
        r = self.lib.orc_gemv_expert(quant, vp(w), sc, bsz, expert, d, n, fp(x), fp(out))
        if r:
            raise RuntimeError(self.err())
        return out

    def embed_row(self, quant, w, dim, token, scale=None, block_size=(0, 0)):
        out = np.zeros(dim, np.float32)
        bsz = (C.c_int32 * 2)(*block_size)
        sc = None if scale is None else fp(np.ascontiguousarray(scale, np.float32))
        if self.lib.orc_embed_row(quant, vp(w), sc, bsz, dim, token, fp(out)):
            raise RuntimeError(self.err())
        return out

    def dequant_row(self, quant, row_bytes, n):
        y = np.zeros(n, np.float32)
        self.lib.orc_dequant_row(quant, vp(row_bytes), n, fp(y))
        return y

    def rmsnorm(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        o = np.zeros_like(x)
        self.lib.orc_rmsnorm(fp(o), fp(x), fp(w), x.size, eps)
        return o

    def sample_prob(self, logits, index):
        """Sampler::sample_prob restated (src/sampler.cpp:12-26)."""
        l = np.ascontiguousarray(logits, np.float32)
        f = self.lib.orc_sample_prob
        f.argtypes, f.restype = [c_f, C.c_int, C.c_int], C.c_float
        return float(f(fp(l), l.size, int(index)))

    def sample(self, logits, temperature, top_p, coin):
        """Sampler::sample / sample_argmax restated (src/sampler.cpp:28-75)."""
        l = np.ascontiguousarray(logits, np.float32)
        f = self.lib.orc_sample
        f.argtypes, f.restype = [c_f, C.c_int, C.c_float, C.c_float, C.c_float], C.c_int
        return int(f(fp(l), l.size, temperature, top_p, coin))

    def moe_gate(self, scores, bias, n_active, norm_topk_prob, scaling, scoring_func, topk_method, n_group, topk_group):
        s = np.ascontiguousarray(scores, np.float32)
        E = s.size
        b = None if bias is None else fp(np.ascontiguousarray(bias, np.float32))
        ae, aw, so = np.zeros(n_active, np.int32), np.zeros(n_active, np.float32), np.zeros(E, np.float32)
        self.lib.orc_moe_gate(fp(s), b, E, n_active, int(norm_topk_prob), scaling, scoring_func, topk_method,
                              n_group, topk_group, fp(ae, c_i32), fp(aw), fp(so))
        return ae, aw, so

    def rope(self, vec, d, pos, theta, is_v3):
        v = np.array(vec, np.float32).copy()
        self.lib.orc_rope(fp(v), d, d, pos, theta, int(is_v3))
        return v

    def attn_mha(self, q, kb, vb, n_heads, head_dim, v_head_dim, kv_len):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros(n_heads * v_head_dim, np.float32)
        att = np.zeros(kv_len, np.float32)
        for h in range(n_heads):
            self.lib.orc_attn(
                C.cast(out.ctypes.data + 4 * h * v_head_dim, c_f), fp(att),
                C.cast(q.ctypes.data + 4 * h * head_dim, c_f),
                C.cast(kb.ctypes.data + 2 * h * head_dim, c_u16), C.cast(vb.ctypes.data + 2 * h * v_head_dim, c_u16),
                head_dim, v_head_dim, n_heads, kv_len)
        return out

    def attn_mla(self, q_c, q_rope, ckv, krope, n_heads, head_dim, lora, rope, kv_len):
        q_c = np.ascontiguousarray(q_c, np.float32)
        q_rope = np.ascontiguousarray(q_rope, np.float32)
        out = np.zeros(n_heads * lora, np.float32)
        att = np.zeros(kv_len, np.float32)
        for h in range(n_heads):
            self.lib.orc_attn_mla(
                C.cast(out.ctypes.data + 4 * h * lora, c_f), fp(att), C.cast(q_c.ctypes.data + 4 * h * lora, c_f),
                C.cast(q_rope.ctypes.data + 4 * h * rope, c_f), fp(ckv, c_u16), fp(krope, c_u16), head_dim, lora,
                rope, kv_len)
        return out

    def model(self, cfg, tensors):
        return OracleModel(self, cfg, tensors)


class OracleModel:
    """orc_model_* life-cycle; keeps the numpy tensors alive (the oracle borrows the pointers)."""

    def __init__(self, orc: Oracle, cfg, tensors):
        from tools import synth
        self.orc, self.cfg, self.tensors = orc, cfg, tensors
        self.dcfg = to_dsk_config(cfg)
        self.h = C.c_void_p()
        if orc.lib.orc_model_create(C.byref(self.dcfg), C.byref(self.h)):
            raise RuntimeError(orc.err())

        def bind(role, layer, quant, shape, arr):
            if orc.lib.orc_model_bind(self.h, role, layer, quant, fp(shape, c_i32), vp(arr), arr.nbytes):
                raise RuntimeError(orc.err())

        synth.bind_all(tensors, bind)
        if orc.lib.orc_model_finalize(self.h):
            raise RuntimeError(orc.err())

    def forward(self, token, pos, mode=1):
        logits = np.zeros(self.cfg.vocab_size, np.float32)
        if self.orc.lib.orc_forward(self.h, token, pos, mode, fp(logits)):
            raise RuntimeError(self.orc.err())
        return logits

    def routing(self):
        K = max(1, self.cfg.n_active_routed)
        e = np.zeros(self.cfg.n_layers * K, np.int32)
        w = np.zeros(self.cfg.n_layers * K, np.float32)
        self.orc.lib.orc_model_get_routing(self.h, fp(e, c_i32), fp(w))
        return e.reshape(self.cfg.n_layers, K), w.reshape(self.cfg.n_layers, K)

    def trace_x(self, layer):
        x = np.zeros(self.cfg.dim, np.float32)
        self.orc.lib.orc_model_get_trace_x(self.h, layer, fp(x))
        return x

    def router_logits(self, layer):
        x = np.zeros(self.cfg.n_routed_experts, np.float32)
        self.orc.lib.orc_model_get_router_logits(self.h, layer, fp(x))
        return x

    def close(self):
        if self.h:
            self.orc.lib.orc_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ref(_Lib):
    """The unmodified reference (oracle/_ref/libdskref.so)."""

    def __init__(self):
        super().__init__(REF_SO)
        L = self.lib
        L.ref_session_create.restype = C.c_void_p
        L.ref_session_create.argtypes = [C.c_char_p, C.c_int]
        L.ref_session_destroy.argtypes = [C.c_void_p]
        L.ref_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f]
        L.ref_forward_traced.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f]
        L.ref_get_routing.argtypes = [C.c_void_p, c_i32, c_f]
        L.ref_get_trace_x.argtypes = [C.c_void_p, C.c_int, c_f]
        L.ref_get_gate_scores.argtypes = [C.c_void_p, C.c_int, c_f]
        L.ref_active_bytes.restype = C.c_double
        L.ref_active_bytes.argtypes = [C.c_void_p, C.c_int]
        L.ref_moe_gate.argtypes = [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                   C.c_int, c_i32, c_f]
        L.ref_rope.argtypes = [c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.ref_rope_f16.argtypes = [c_u16, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.ref_rmsnorm.argtypes = [c_f, c_f, c_f, C.c_int, C.c_float]
        L.ref_float_to_half.restype = C.c_uint16
        L.ref_float_to_half.argtypes = [C.c_float]
        L.ref_half_to_float.restype = C.c_float
        L.ref_half_to_float.argtypes = [C.c_uint16]
        L.ref_float_to_f8e5m2.restype = C.c_uint8
        L.ref_float_to_f8e5m2.argtypes = [C.c_float]
        L.ref_f8e5m2_to_float.restype = C.c_float
        L.ref_f8e5m2_to_float.argtypes = [C.c_uint8]
        L.ref_quantize_row.argtypes = [C.c_int, c_f, C.c_void_p, C.c_int64]

    def set_threads(self, n):
        self.lib.ref_set_threads(n)

    def q8k_quantize(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n = x.size
        qs, d, bs = np.zeros(n, np.int8), np.zeros(n // 256, np.float32), np.zeros(n // 16, np.int16)
        self.lib.ref_q8k_quantize(fp(x), n, fp(qs, c_i8), fp(d), fp(bs, c_i16))
        return qs, d, bs

    def gemv(self, quant, w, d, n, x, scale=None, block_size=(0, 0)):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(d, np.float32)
        bsz = (C.c_int * 2)(*block_size)
        sc = None if scale is None else fp(np.ascontiguousarray(scale, np.float32))
        self.lib.ref_gemv(quant, vp(w), sc, bsz, d, n, fp(x), fp(out))
        return out

    def gemv_expert(self, quant, w, n_experts, expert, d, n, x, scale=None, block_size=(0, 0)):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(d, np.float32)
        bsz = (C.c_int * 2)(*block_size)
        sc = None if scale is None else fp(np.ascontiguousarray(scale, np.float32))
        self.lib.ref_gemv_expert(quant, vp(w), sc, bsz, n_experts, expert, d, n, fp(x), fp(out))
        return out

    def dequant_row(self, quant, row_bytes, n):
        y = np.zeros(n, np.float32)
        self.lib.ref_dequant_row(quant, vp(row_bytes), n, fp(y))
        return y

    def quantize_rows(self, quant, w):
        """reference offline quantizer (src/quant.cpp:147,308) on a (rows, n) matrix."""
        w = np.ascontiguousarray(w, np.float32)
        rows, n = w.shape
        bpb = 84 if quant == 3 else 110
        out = np.zeros((rows, n // 256 * bpb), np.uint8)
        self.lib.ref_quantize_row(quant, fp(w), vp(out), rows * n)
        return out

    def rmsnorm(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32).copy()
        w = np.ascontiguousarray(w, np.float32).copy()
        o = np.zeros_like(x)
        self.lib.ref_rmsnorm(fp(o), fp(x), fp(w), x.size, eps)
        return o

    def moe_gate(self, scores, bias, n_active, norm_topk_prob, scaling, scoring_func, topk_method, n_group, topk_group):
        s = np.array(scores, np.float32).copy()
        E = s.size
        b = None if bias is None else fp(np.ascontiguousarray(bias, np.float32))
        ae, aw = np.zeros(n_active, np.int32), np.zeros(n_active, np.float32)
        self.lib.ref_moe_gate(fp(s), b, E, n_active, int(norm_topk_prob), scaling, scoring_func, topk_method,
                              n_group, topk_group, fp(ae, c_i32), fp(aw))
        return ae, aw, s

    def rope(self, vec, d, pos, theta, is_v3):
        v = np.array(vec, np.float32).copy()
        self.lib.ref_rope(fp(v), d, d, pos, theta, int(is_v3))
        return v

    def attn_mha(self, q, kb, vb, n_heads, head_dim, v_head_dim, kv_len):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros(n_heads * v_head_dim, np.float32)
        att = np.zeros(kv_len, np.float32)
        for h in range(n_heads):
            self.lib.ref_attn(
                C.cast(out.ctypes.data + 4 * h * v_head_dim, c_f), fp(att),
                C.cast(q.ctypes.data + 4 * h * head_dim, c_f),
                C.cast(kb.ctypes.data + 2 * h * head_dim, c_u16), C.cast(vb.ctypes.data + 2 * h * v_head_dim, c_u16),
                head_dim, v_head_dim, n_heads, kv_len)
        return out

    def attn_mla(self, q_c, q_rope, ckv, krope, n_heads, head_dim, lora, rope, kv_len):
        q_c = np.ascontiguousarray(q_c, np.float32)
        q_rope = np.ascontiguousarray(q_rope, np.float32)
        out = np.zeros(n_heads * lora, np.float32)
        att = np.zeros(kv_len, np.float32)
        for h in range(n_heads):
            self.lib.ref_attn_mla(
                C.cast(out.ctypes.data + 4 * h * lora, c_f), fp(att), C.cast(q_c.ctypes.data + 4 * h * lora, c_f),
                C.cast(q_rope.ctypes.data + 4 * h * rope, c_f), fp(ckv, c_u16), fp(krope, c_u16), head_dim, lora,
                rope, kv_len)
        return out

    def session(self, dirname, cfg, context=0):
        return RefSession(self, dirname, cfg, context)


class RefSession:
    def __init__(self, ref: Ref, dirname, cfg, context=0):
        self.ref, self.cfg = ref, cfg
        self.h = ref.lib.ref_session_create(dirname.encode(), context)

    def forward(self, token, pos, mode=1, traced=True):
        logits = np.zeros(self.cfg.vocab_size, np.float32)
        (self.ref.lib.ref_forward_traced if traced else self.ref.lib.ref_forward)(self.h, token, pos, mode, fp(logits))
        return logits

    def routing(self):
        K = max(1, self.cfg.n_active_routed)
        e = np.zeros(self.cfg.n_layers * K, np.int32)
        w = np.zeros(self.cfg.n_layers * K, np.float32)
        self.ref.lib.ref_get_routing(self.h, fp(e, c_i32), fp(w))
        return e.reshape(self.cfg.n_layers, K), w.reshape(self.cfg.n_layers, K)

    def trace_x(self, layer):
        x = np.zeros(self.cfg.dim, np.float32)
        self.ref.lib.ref_get_trace_x(self.h, layer, fp(x))
        return x

    def active_bytes(self, pos):
        return self.ref.lib.ref_active_bytes(self.h, pos)

    def sample_prob(self, logits, index):
        """The reference's own Sampler::sample_prob on `logits`."""
        l = np.ascontiguousarray(logits, np.float32)
        assert l.size == self.cfg.vocab_size
        f = self.ref.lib.ref_sample_prob
        f.argtypes, f.restype = [C.c_void_p, c_f, C.c_int], C.c_float
        return float(f(self.h, fp(l), int(index)))

    def sample(self, logits, temperature, top_p, seed):
        """The reference's own Sampler::sample on `logits` (vocab_size floats), seeded with `seed`;
        returns (token, coin) with coin = the rand() / RAND_MAX it drew."""
        l = np.ascontiguousarray(logits, np.float32)
        assert l.size == self.cfg.vocab_size
        f = self.ref.lib.ref_sample
        f.argtypes, f.restype = [C.c_void_p, c_f, C.c_float, C.c_float, C.c_uint, C.POINTER(C.c_float)], C.c_int
        coin = C.c_float()
        tok = f(self.h, fp(l), temperature, top_p, seed, C.byref(coin))
        return int(tok), float(coin.value)

    def close(self):
        if self.h:
            self.ref.lib.ref_session_destroy(self.h)
            self.h = None


if __name__ == "__main__":
    build()
    print("built", ORC_SO, os.path.exists(REF_SO) and REF_SO)
