"""ctypes binding of the gfx950 decode engine's C ABI (include/dsk.h).

This is plumbing only: every call goes straight into deepseek.cpp_amd/libdsk_hip.so (hand-written
HIP kernels).  There is NO CPU or PyTorch fallback: if the shared library is missing or no GPU is
visible the import / context creation raises.

Object model mirrors the reference (src/model.h): `Model.forward(token, pos, mode)` is
`Model::forward(InferenceState&, token, pos, mode)` and returns what `InferenceState::logits()`
would hold.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSK_LIB") or os.path.join(HERE, "libdsk_hip.so")  # DSK_LIB: A/B builds of the same library

MODE_HYDRATE_KV_CACHE, MODE_OUTPUT_LOGITS = 0, 1
QUANT_IDS = {"fp32": 0, "fp16": 1, "f8e5m2": 2, "q2_k": 3, "q3_k": 4}

c_f = C.POINTER(C.c_float)
c_i32 = C.POINTER(C.c_int32)


class DskConfig(C.Structure):
    """dsk_config (include/dsk.h), POD mirror of reference Config (src/model.h:47-96)."""
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


class LoadStats(C.Structure):
    """dsk_load_stats (include/dsk.h)."""
    _fields_ = [("file_bytes", C.c_uint64), ("staged_bytes", C.c_uint64), ("seconds", C.c_double),
                ("read_seconds", C.c_double), ("n_files", C.c_int32), ("n_tensors", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int32), ("total_ms", C.c_float), ("algo_bytes", C.c_double)]


def make_config(c) -> DskConfig:
    """Any object with the reference Config's field names (e.g. tools.synth.Cfg) -> dsk_config."""
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


class DskError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libdsk_hip.so; fail loudly (the product path has no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DskError(f"{LIB_PATH} is missing: build it with `make -C deepseek.cpp_amd/csrc` "
                           f"(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.dsk_last_error.restype = C.c_char_p
        L.dsk_model_active_bytes.restype = C.c_double
        L.dsk_model_active_bytes.argtypes = [C.c_void_p, C.c_int]
        L.dsk_model_device_bytes.restype = C.c_double
        L.dsk_model_device_bytes.argtypes = [C.c_void_p]
        L.dsk_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.dsk_ctx_destroy.argtypes = [C.c_void_p]
        L.dsk_comm_unique_id.argtypes = [C.c_void_p]
        L.dsk_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.dsk_model_create.argtypes = [C.c_void_p, C.POINTER(DskConfig), C.POINTER(C.c_void_p)]
        L.dsk_model_bind.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_i32, C.c_void_p, C.c_size_t]
        L.dsk_model_synthesize.argtypes = [C.c_void_p, C.c_uint64]
        L.dsk_dseek_read_config.argtypes = [C.c_char_p, C.c_int, C.POINTER(DskConfig), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_uint64)]
        L.dsk_model_load_dseek.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(LoadStats)]
        L.dsk_model_finalize.argtypes = [C.c_void_p]
        L.dsk_model_destroy.argtypes = [C.c_void_p]
        L.dsk_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f]
        L.dsk_model_set_graph.argtypes = [C.c_void_p, C.c_int]
        L.dsk_model_set_trace.argtypes = [C.c_void_p, C.c_int]
        L.dsk_model_get_trace_x.argtypes = [C.c_void_p, C.c_int, c_f]
        L.dsk_model_host_logits.argtypes = [C.c_void_p]
        L.dsk_model_host_logits.restype = C.POINTER(C.c_float)
        L.dsk_forward_argmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.dsk_model_get_routing.argtypes = [C.c_void_p, c_i32, c_f]
        L.dsk_model_get_slot_outputs.argtypes = [C.c_void_p, c_f]
        L.dsk_profile_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(KernelTime), C.c_int, c_i32]
        L.dsk_q8k_quantize.argtypes = [C.c_void_p, c_f, C.c_int, C.c_void_p, c_f, C.c_void_p]
        L.dsk_gemv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, c_f, c_i32, C.c_int, C.c_int, c_f, c_f]
        L.dsk_gemv_expert.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, c_f, c_i32, C.c_int, C.c_int,
                                      C.c_int, C.c_int, c_f, c_f]
        L.dsk_embed_row.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, c_f, c_i32, C.c_int, C.c_int,
                                    C.c_int, c_f]
        L.dsk_rmsnorm.argtypes = [C.c_void_p, c_f, c_f, C.c_int, C.c_float, c_f]
        L.dsk_moe_gate.argtypes = [C.c_void_p, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                   C.c_int, C.c_int, c_i32, c_f]
        L.dsk_rope.argtypes = [C.c_void_p, c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.dsk_attn_mha.argtypes = [C.c_void_p, c_f, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_f]
        L.dsk_attn_mla.argtypes = [C.c_void_p, c_f, c_f, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, c_f]
        L.dsk_measure_read_bw.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.dsk_time_kernel_class.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.dsk_expert_shard.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.dsk_model_run_block.argtypes = [C.c_void_p, C.c_int, c_f, C.c_int, c_f]
        L.dsk_model_run_head.argtypes = [C.c_void_p, c_f, c_f]
        L.dsk_model_get_stage.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        L.dsk_router_logits.argtypes = [C.c_void_p, c_f, c_f, c_f, C.c_float, C.c_int, C.c_int, c_f]
        L.dsk_model_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.dsk_model_get_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.dsk_ctx_live_models.argtypes = [C.c_void_p]
        L.dsk_bench_router.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.POINTER(C.c_double)]
        L.dsk_bench_gemv.argtypes = [C.c_void_p] + [C.c_int] * 11 + [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def check(r):
    if r != 0:
        raise DskError(lib().dsk_last_error().decode())


def plan_gemv(quant: int, rows: int, n: int, n_tasks: int = 1, kind: int = 0, act_mode: int = 2, target_wgs: int = 0) -> dict:
    """The launch planner's geometry for n_tasks equal (rows x n) matrices (include/dsk.h dsk_plan_gemv; host only)."""
    out = (C.c_int * 8)()
    f = lib().dsk_plan_gemv
    f.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)]
    check(f(quant, rows, n, n_tasks, kind, act_mode, target_wgs, out))
    keys = ("lanes_per_row", "R", "U", "waves", "grid", "lds_bytes", "groups", "rows_per_step")
    return dict(zip(keys, list(out)))


def plan_gemv_ahead(quant: int, rows: int, n: int, n_tasks: int = 1, kind: int = 0, act_mode: int = 2, kvwrite: bool = False) -> int:
    """Which 'weights ahead of the staging' kernel the engine runs for that launch (include/dsk.h dsk_plan_gemv_ahead; host only):
    0 none, 1 gemv_ahead_kernel, 2 gemv_ahead_q8_kernel, 3 gemv_kvwrite_ahead_kernel."""
    out = C.c_int(0)
    f = lib().dsk_plan_gemv_ahead
    f.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)]
    check(f(quant, rows, n, n_tasks, kind, act_mode, int(kvwrite), C.byref(out)))
    return out.value



def _f(a):
    return a.ctypes.data_as(c_f)


def _fa(a):
    return np.ascontiguousarray(a, np.float32)


class Ctx:
    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        check(lib().dsk_ctx_create(device, C.byref(self.h)))
        self.rank, self.world = 0, 1

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        check(lib().dsk_comm_unique_id(buf))
        return buf.raw

    def comm_init_dry(self, rank: int, world: int):
        """shard `rank` of `world` without a communicator (single-GPU validation of the sharded path)"""
        check(lib().dsk_comm_init(self.h, None, rank, world))

    def comm_init(self, uid: bytes, rank: int, world: int):
        check(lib().dsk_comm_init(self.h, C.c_char_p(uid), rank, world))
        self.rank, self.world = rank, world

    def live_models(self) -> int:
        return lib().dsk_ctx_live_models(self.h)

    def close(self):
        if self.h:
            lib().dsk_ctx_destroy(self.h)
            self.h = C.c_void_p()

    # ---- op-level entry points (host buffers) ----
    def sample(self, logits, temperature: float, top_p: float, coin: float) -> int:
        l = _fa(logits)
        tok = C.c_int32()
        f = lib().dsk_sample
        f.argtypes = [C.c_void_p, c_f, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]
        check(f(self.h, _f(l), l.size, temperature, top_p, coin, C.byref(tok)))
        return tok.value

    def q8k_quantize(self, x):
        x = _fa(x)
        n = x.size
        qs, d, bs = np.zeros(n, np.int8), np.zeros(n // 256, np.float32), np.zeros(n // 16, np.int16)
        check(lib().dsk_q8k_quantize(self.h, _f(x), n, qs.ctypes.data, _f(d), bs.ctypes.data))
        return qs, d, bs

    def gemv(self, quant, w, d, n, x, scale=None, block_size=(0, 0)):
        x, out = _fa(x), np.zeros(d, np.float32)
        w = np.ascontiguousarray(w)
        bsz = (C.c_int32 * 2)(*block_size)
        sc = None if scale is None else _f(_fa(scale))
        check(lib().dsk_gemv(self.h, quant, w.ctypes.data, w.nbytes, sc, bsz, d, n, _f(x), _f(out)))
        return out

    def gemv_expert(self, quant, w, n_experts, expert, d, n, x, scale=None, block_size=(0, 0)):
        x, out = _fa(x), np.zeros(d, np.float32)
        w = np.ascontiguousarray(w)
        bsz = (C.c_int32 * 2)(*block_size)
        sc = None if scale is None else _f(_fa(scale))
        check(lib().dsk_gemv_expert(self.h, quant, w.ctypes.data, w.nbytes, sc, bsz, n_experts, expert, d, n, _f(x), _f(out)))
        return out

    def embed_row(self, quant, w, vocab, dim, token, scale=None, block_size=(0, 0)):
        out = np.zeros(dim, np.float32)
        w = np.ascontiguousarray(w)
        bsz = (C.c_int32 * 2)(*block_size)
        sc = None if scale is None else _f(_fa(scale))
        check(lib().dsk_embed_row(self.h, quant, w.ctypes.data, w.nbytes, sc, bsz, vocab, dim, token, _f(out)))
        return out

    def rmsnorm(self, x, w, eps):
        x, w = _fa(x), _fa(w)
        out = np.zeros_like(x)
        check(lib().dsk_rmsnorm(self.h, _f(x), _f(w), x.size, eps, _f(out)))
        return out

    def moe_gate(self, scores, bias, n_active, norm_topk_prob, scaling, scoring_func, topk_method, n_group, topk_group):
        s = _fa(scores)
        b = None if bias is None else _f(_fa(bias))
        ae, aw = np.zeros(n_active, np.int32), np.zeros(n_active, np.float32)
        check(lib().dsk_moe_gate(self.h, _f(s), b, s.size, n_active, int(norm_topk_prob), scaling, scoring_func,
                                 topk_method, n_group, topk_group, ae.ctypes.data_as(c_i32), _f(aw)))
        return ae, aw

    def rope(self, vec, n_heads, d, pos, theta, is_v3):
        v = np.array(vec, np.float32).copy()
        check(lib().dsk_rope(self.h, _f(v), n_heads, d, pos, theta, int(is_v3)))
        return v

    def attn_mha(self, q, kb, vb, n_heads, head_dim, v_head_dim, kv_len):
        q = _fa(q)
        out = np.zeros(n_heads * v_head_dim, np.float32)
        kb, vb = np.ascontiguousarray(kb), np.ascontiguousarray(vb)
        check(lib().dsk_attn_mha(self.h, _f(q), kb.ctypes.data, vb.ctypes.data, n_heads, head_dim, v_head_dim, kv_len, _f(out)))
        return out

    def attn_mla(self, q_c, q_rope, ckv, krope, n_heads, head_dim, lora, rope, kv_len):
        q_c, q_rope = _fa(q_c), _fa(q_rope)
        out = np.zeros(n_heads * lora, np.float32)
        ckv, krope = np.ascontiguousarray(ckv), np.ascontiguousarray(krope)
        check(lib().dsk_attn_mla(self.h, _f(q_c), _f(q_rope), ckv.ctypes.data, krope.ctypes.data, n_heads, head_dim,
                                 lora, rope, kv_len, _f(out)))
        return out

    def router_logits(self, w, x, norm_w=None, eps=1e-6):
        """raw router logits W(E, dim) . rmsnorm(x, norm_w) in the model's router kernel (src/infer.cpp:839,847)"""
        w, x = _fa(w), _fa(x)
        E, dim = w.shape
        out = np.zeros(E, np.float32)
        nw = None if norm_w is None else _f(_fa(norm_w))
        check(lib().dsk_router_logits(self.h, _f(w), _f(x), nw, eps, E, dim, _f(out)))
        return out

    def bench_gemv(self, quant, rows, n, n_tasks=1, kind=0, act_mode=0, lpr=0, R=0, U=0, target_wgs=0, iters=50):
        """-> (us per launch, weight bytes per launch)"""
        us, nb = C.c_double(), C.c_double()
        check(lib().dsk_bench_gemv(self.h, quant, rows, n, n_tasks, kind, act_mode, lpr, R, U, target_wgs, iters,
                                   C.byref(us), C.byref(nb)))
        return us.value, nb.value

    def bench_router(self, n_routed=256, dim=7168, ksplit=4, flags=0, iters=50):
        us = C.c_double()
        check(lib().dsk_bench_router(self.h, n_routed, dim, ksplit, flags, iters, C.byref(us)))
        return us.value

    def measure_read_bw(self, nbytes=8 << 30, iters=5) -> float:
        out = C.c_double()
        check(lib().dsk_measure_read_bw(self.h, nbytes, iters, C.byref(out)))
        return out.value


def read_dseek_config(dirname: str, context: int = 0):
    """dsk_dseek_read_config: (DskConfig, n_files, n_tensors, tensor_bytes) of a .dseek directory; needs no GPU."""
    d, nf, nt, nb = DskConfig(), C.c_int32(), C.c_int32(), C.c_uint64()
    check(lib().dsk_dseek_read_config(dirname.encode(), context, C.byref(d), C.byref(nf), C.byref(nt), C.byref(nb)))
    return d, nf.value, nt.value, nb.value


class Model:
    """dsk_model_* life-cycle.  `tensors`: name -> object with .data/.shape/.quant/.scale (tools.synth.Tens),
    named like the reference's .dseek tensors; or None + synth_seed to generate weights in HBM."""

    def __init__(self, ctx: Ctx, cfg, tensors=None, synth_seed=None, options=None):
        self.ctx, self.cfg = ctx, cfg
        self.dcfg = make_config(cfg)
        self.h = C.c_void_p()
        check(lib().dsk_model_create(ctx.h, C.byref(self.dcfg), C.byref(self.h)))
        for k, v in (options or {}).items():  # include/dsk.h dsk_model_set_option: between create and finalize
            check(lib().dsk_model_set_option(self.h, k.encode(), int(v)))
        if tensors is not None:
            from tools import synth  # name -> role mapping shared with the test generators

            def bind(role, layer, quant, shape, arr):
                arr = np.ascontiguousarray(arr)
                check(lib().dsk_model_bind(self.h, role, layer, quant, shape.ctypes.data_as(c_i32), arr.ctypes.data, arr.nbytes))

            synth.bind_all(tensors, bind)
        if synth_seed is not None:
            check(lib().dsk_model_synthesize(self.h, synth_seed))
        check(lib().dsk_model_finalize(self.h))
        self._logits = np.zeros(cfg.vocab_size, np.float32)
        self._pinned = None
        self._pinned_ptr = None

    @classmethod
    def from_dseek(cls, ctx: "Ctx", dirname: str, context: int = 0) -> "Model":
        """dsk_model_load_dseek: the checkpoint directory goes straight to HBM (loader.cpp); .load_stats has the rates."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.h = C.c_void_p()
        self.load_stats = LoadStats()
        check(lib().dsk_model_load_dseek(ctx.h, dirname.encode(), context, C.byref(self.h), C.byref(self.load_stats)))
        self.dcfg = read_dseek_config(dirname, context)[0]
        self.cfg = self.dcfg  # same field names as the reference Config
        self._logits = np.zeros(self.cfg.vocab_size, np.float32)
        self._pinned = None
        self._pinned_ptr = None
        return self

    def forward(self, token: int, pos: int, mode: int = MODE_OUTPUT_LOGITS):
        check(lib().dsk_forward(self.h, token, pos, mode, _f(self._logits)))
        return self._logits.copy() if mode == MODE_OUTPUT_LOGITS else None

    def forward_nocopy(self, token: int, pos: int, mode: int = MODE_OUTPUT_LOGITS):
        """logits land in the engine's pinned host buffer; returns a numpy view of it (valid until the next step)"""
        if self._pinned is None:
            ptr = lib().dsk_model_host_logits(self.h)
            self._pinned = np.ctypeslib.as_array(ptr, shape=(self.cfg.vocab_size,))
            self._pinned_ptr = ptr
        check(lib().dsk_forward(self.h, token, pos, mode, self._pinned_ptr))
        return self._pinned

    def set_graph(self, on: bool):
        check(lib().dsk_model_set_graph(self.h, int(on)))

    def set_trace(self, on: bool):
        check(lib().dsk_model_set_trace(self.h, int(on)))

    def trace_x(self, layer: int):
        x = np.zeros(self.cfg.dim, np.float32)
        check(lib().dsk_model_get_trace_x(self.h, layer, _f(x)))
        return x

    def forward_sample(self, token: int, pos: int, temperature: float, top_p: float, coin: float) -> int:
        nxt = C.c_int32()
        f = lib().dsk_forward_sample
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]
        check(f(self.h, token, pos, temperature, top_p, coin, C.byref(nxt)))
        return nxt.value

    def forward_prob(self, token: int, pos: int, index: int) -> float:
        pr = C.c_float()
        f = lib().dsk_forward_prob
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        check(f(self.h, token, pos, index, C.byref(pr)))
        return float(pr.value)

    def forward_argmax(self, token: int, pos: int) -> int:
        nxt = C.c_int32()
        check(lib().dsk_forward_argmax(self.h, token, pos, C.byref(nxt)))
        return nxt.value

    def run_block(self, layer: int, x_in, pos: int):
        """dsk_model_run_block: block `layer` on the residual stream x_in at `pos`, every Q8_K staging point tapped"""
        x_in = _fa(x_in)
        out = np.zeros(self.cfg.dim, np.float32)
        check(lib().dsk_model_run_block(self.h, layer, _f(x_in), pos, _f(out)))
        return out

    def run_head(self, x_in):
        x_in = _fa(x_in)
        out = np.zeros(self.cfg.vocab_size, np.float32)
        check(lib().dsk_model_run_head(self.h, _f(x_in), _f(out)))
        return out

    def stage(self, name: str, n: int, dtype=np.float32):
        """n elements of a named stage buffer of the last run_block / run_head (include/dsk.h dsk_model_get_stage)"""
        out = np.zeros(n, dtype)
        check(lib().dsk_model_get_stage(self.h, name.encode(), out.ctypes.data, out.nbytes))
        return out

    def set_cache_rows(self, layer: int, cache: str, row0: int, rows):
        """dsk_model_set_cache_rows: rows = (n, width) uint16 f16 bits written at cache rows [row0, row0 + n)"""
        rows = np.ascontiguousarray(rows, np.uint16)
        f = lib().dsk_model_set_cache_rows
        f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        check(f(self.h, layer, cache.encode(), row0, rows.shape[0], rows.ctypes.data))

    def get_cache_rows(self, layer: int, cache: str, row0: int, nrows: int, width: int):
        """dsk_model_get_cache_rows: (nrows, width) uint16 f16 bits of cache rows [row0, row0 + nrows)"""
        out = np.zeros((nrows, width), np.uint16)
        f = lib().dsk_model_get_cache_rows
        f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        check(f(self.h, layer, cache.encode(), row0, nrows, out.ctypes.data))
        return out

    def hydrate(self, tokens, pos0: int = 0, mode: int = MODE_HYDRATE_KV_CACHE):
        """dsk_hydrate: the prompt loop of src/main.cpp:312-319 (one forward per token), run as batched launches when the
        model qualifies; returns the last token's logits when mode == MODE_OUTPUT_LOGITS"""
        toks = np.ascontiguousarray(tokens, np.int32)
        f = lib().dsk_hydrate
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        check(f(self.h, toks.ctypes.data, int(toks.size), pos0, mode, self._logits.ctypes.data))
        return self._logits.copy() if mode == MODE_OUTPUT_LOGITS else None

    def hydrate_why_not(self) -> str:
        f = lib().dsk_hydrate_why_not
        f.argtypes = [C.c_void_p]
        f.restype = C.c_char_p
        return f(self.h).decode()

    def hydrate_buffer(self, name: str, row0: int, rows: int, width: int, dtype=np.float32):
        """dsk_hydrate_get_buffer: (rows, width) of a named intermediate of the last batched chunk (one row per token)"""
        out = np.zeros((rows, width), dtype)
        f = lib().dsk_hydrate_get_buffer
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        check(f(self.h, name.encode(), row0, rows, out.ctypes.data, out.nbytes))
        return out

    def hydrate_trace_x(self, layer: int, index: int):
        x = np.zeros(self.cfg.dim, np.float32)
        f = lib().dsk_hydrate_get_trace_x
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        check(f(self.h, layer, index, x.ctypes.data))
        return x

    def timeline(self, kind: int, n_wgs: int = 1024):
        """(n_wgs, 8) wall-clock stamps (100 MHz ticks) of the LAST launch of one kind in a token; the model must have been
        created with options={"timeline": 1} (include/dsk.h dsk_model_get_timeline; rows of unused workgroups are 0)"""
        out = np.zeros((n_wgs, 8), np.uint64)
        f = lib().dsk_model_get_timeline
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        check(f(self.h, kind, out.ctypes.data, n_wgs))
        return out

    def stage_q8(self, point: str, n: int):
        """(int8 codes, block scales) the device staged at a Q8_K point"""
        return self.stage(f"q8.{point}.qs", n, np.int8), self.stage(f"q8.{point}.d", n // 256, np.float32)

    def routing(self):
        K = max(1, self.cfg.n_active_routed)
        e = np.zeros(self.cfg.n_layers * K, np.int32)
        w = np.zeros(self.cfg.n_layers * K, np.float32)
        check(lib().dsk_model_get_routing(self.h, e.ctypes.data_as(c_i32), _f(w)))
        return e.reshape(self.cfg.n_layers, K), w.reshape(self.cfg.n_layers, K)

    def slot_outputs(self):
        n = self.cfg.n_active_routed + (1 if self.cfg.n_shared_experts > 0 else 0)
        out = np.zeros((n, self.cfg.dim), np.float32)
        check(lib().dsk_model_get_slot_outputs(self.h, _f(out)))
        return out

    def profile_forward(self, token: int, pos: int):
        arr = (KernelTime * 64)()
        n = C.c_int32()
        check(lib().dsk_profile_forward(self.h, token, pos, arr, 64, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     algo_bytes=arr[i].algo_bytes) for i in range(n.value)]

    def time_kernel_class(self, name: str, pos: int, reps: int = 8):
        """-> (us per launch, algorithmic bytes per launch, launches per token)"""
        us, nb, n = C.c_double(), C.c_double(), C.c_int()
        check(lib().dsk_time_kernel_class(self.h, name.encode(), pos, reps, C.byref(us), C.byref(nb), C.byref(n)))
        return us.value, nb.value, n.value

    def info(self, key: str) -> int:
        v = C.c_int()
        check(lib().dsk_model_get_info(self.h, key.encode(), C.byref(v)))
        return v.value

    def active_bytes(self, pos: int) -> float:
        return lib().dsk_model_active_bytes(self.h, pos)

    def device_bytes(self) -> float:
        return lib().dsk_model_device_bytes(self.h)

    def close(self):
        if self.h:
            lib().dsk_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
