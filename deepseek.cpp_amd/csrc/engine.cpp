// engine.cpp -- host side of the gfx950 decode engine behind the C ABI (include/dsk.h).
//
// Mirrors what the reference does around its hot path, for one GPU per process:
//   Model::Model / Block ctors (src/model.cpp:149-285, 354-461, 518-620, 756-872) -> dsk_model_bind
//   InferenceState scratch (src/model.cpp:677-726)                                -> dsk_model_finalize
//   Model::_forward_cpu / Block::_block_cpu (src/infer.cpp:1265-1317, 810-932)    -> dsk_forward
// All kernels of a token are enqueued on one HIP stream without host synchronisation; routing
// decisions stay in HBM; the whole step is captured once into a hipGraph and replayed.
#include "dsk_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include <rccl/rccl.h>

// ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dsk_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  int n = snprintf(g_err, sizeof g_err, "[dsk %d] ", code);
  vsnprintf(g_err + n, sizeof g_err - n, fmt, ap);
  va_end(ap);
}
extern "C" const char* dsk_last_error(void) { return g_err; }
extern "C" int dsk_abi_version(void) { return DSK_ABI_VERSION; }

struct dsk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // scratch for op-level entry points
  void* op_buf[8] = {nullptr};
  size_t op_cap[8] = {0};
};

extern "C" int dsk_ctx_create(int device_ordinal, dsk_ctx** out) {
  if (!out) DSK_FAIL(DSK_ERR_INVALID, "ctx_create: null out");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) DSK_FAIL(DSK_ERR_HIP, "no HIP device visible: this engine has no CPU fallback");
  if (device_ordinal < 0 || device_ordinal >= ndev) DSK_FAIL(DSK_ERR_INVALID, "device %d of %d", device_ordinal, ndev);
  HIP_TRY(hipSetDevice(device_ordinal));
  dsk_ctx* c = new dsk_ctx();
  c->device = device_ordinal;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return DSK_OK;
}
extern "C" int dsk_ctx_destroy(dsk_ctx* c) {
  if (!c) return DSK_OK;
  hipSetDevice(c->device);
  for (int i = 0; i < 8; ++i)
    if (c->op_buf[i]) hipFree(c->op_buf[i]);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return DSK_OK;
}
int ctx_scratch(dsk_ctx* c, int slot, size_t bytes, void** out) {
  if (c->op_cap[slot] < bytes) {
    if (c->op_buf[slot]) HIP_TRY(hipFree(c->op_buf[slot]));
    c->op_buf[slot] = nullptr;
    c->op_cap[slot] = 0;
    HIP_TRY(hipMalloc(&c->op_buf[slot], bytes));
    c->op_cap[slot] = bytes;
  }
  *out = c->op_buf[slot];
  return DSK_OK;
}
hipStream_t ctx_stream(dsk_ctx* c) { return c->stream; }
int ctx_device(dsk_ctx* c) { return c->device; }

extern "C" int dsk_comm_unique_id(void* uid128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) DSK_FAIL(DSK_ERR_COMM, "ncclGetUniqueId failed");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  memcpy(uid128, &id, 128);
  return DSK_OK;
}
extern "C" int dsk_comm_init(dsk_ctx* c, const void* uid128, int rank, int world) {
  if (!c || world < 1 || rank < 0 || rank >= world) DSK_FAIL(DSK_ERR_INVALID, "comm_init: rank %d / world %d", rank, world);
  c->rank = rank;
  c->world = world;
  if (world == 1) return DSK_OK;
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, uid128, 128);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) DSK_FAIL(DSK_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
  return DSK_OK;
}

// ---------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------
static const int NROLES = 32;
static const int KV_SINKS_GUARD = 2;  // KV_SINKS, src/model.h:14

struct Layer {
  DTensor t[NROLES];
  uint16_t *key_cache = nullptr, *value_cache = nullptr;    // MHA (src/model.cpp:459-460)
  uint16_t *nope_cache = nullptr, *rope_cache = nullptr;    // MLA (src/model.cpp:618-619)
  bool is_moe = false;
};

struct Q8Buf {  // Q8_K activation vector in HBM (struct-of-arrays of block_q8_K, src/quant.h:104-109)
  int8_t* qs = nullptr;
  float* d = nullptr;
  int16_t* bsums = nullptr;
  int cap = 0;
};

struct KTime {
  const char* name;
  int launches = 0;
  double algo_bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};

struct dsk_model {
  dsk_ctx* ctx = nullptr;
  dsk_config c{};
  int head_dim = 0;
  DTensor g[4];  // EMBED, FINAL_NORM, OUTPUT
  std::vector<Layer> L;
  bool finalized = false;
  bool tied = false;
  double weight_bytes = 0, cache_bytes = 0, scratch_bytes = 0;
  // activations (src/model.h:151-178)
  float *x = nullptr, *xb = nullptr, *q_a = nullptr, *q = nullptr, *kv_a = nullptr, *kv_b = nullptr, *att_out = nullptr,
        *hb = nullptr, *eout = nullptr, *q_c = nullptr, *q_rope = nullptr, *vb_out = nullptr, *router_partial = nullptr,
        *gate_scores = nullptr, *logits = nullptr, *trace_x = nullptr;
  int* route_e = nullptr;    // [n_layers][K] : every layer keeps its own routing decision
  float* route_w = nullptr;  // [n_layers][K]
  Q8Buf a_xb, a_qa, a_kva, a_att, a_hb;
  StepParams* sp_dev = nullptr;
  StepParams* sp_host = nullptr;  // pinned
  float* logits_host = nullptr;   // pinned
  int router_ksplit = 1;
  int n_slots = 0;  // routed slots (K) + 1 if shared experts
  // graphs
  bool use_graph = true, trace = false;
  hipGraphExec_t graph[2] = {nullptr, nullptr};
  // profiling
  bool profiling = false;
  std::vector<KTime> ktimes;
  std::map<std::string, int> kindex;
};

bool is_kq(int q) { return q == DSK_QUANT_Q2_K || q == DSK_QUANT_Q3_K; }
static int cdiv(int a, int b) { return (a + b - 1) / b; }
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bytes of a (rows, n) matrix in the reference's storage (src/codec.cpp:166-234, src/quant.h)
size_t mat_bytes(int quant, size_t rows, size_t n) {
  switch (quant) {
    case DSK_QUANT_F32: return rows * n * 4;
    case DSK_QUANT_F16: return rows * n * 2;
    case DSK_QUANT_F8E5M2: return rows * n;
    case DSK_QUANT_Q2_K: return rows * n / 256 * 84;
    case DSK_QUANT_Q3_K: return rows * n / 256 * 110;
  }
  return 0;
}

extern "C" int dsk_model_create(dsk_ctx* ctx, const dsk_config* cfg, dsk_model** out) {
  if (!ctx || !cfg || !out) DSK_FAIL(DSK_ERR_INVALID, "model_create: null argument");
  const dsk_config& c = *cfg;
  if (c.dim <= 0 || c.n_layers <= 0 || c.n_heads <= 0 || c.vocab_size <= 0 || c.max_seq_len <= 0)
    DSK_FAIL(DSK_ERR_INVALID, "model_create: non-positive dimension");
  if (c.weight_quant < DSK_QUANT_F32 || c.weight_quant > DSK_QUANT_Q3_K) DSK_FAIL(DSK_ERR_INVALID, "unsupported quant %d", c.weight_quant);
  if (c.weight_quant == DSK_QUANT_F8E5M2 && (c.block_size[0] <= 0 || c.block_size[1] <= 0 || c.block_size[1] % 16))
    DSK_FAIL(DSK_ERR_INVALID, "f8e5m2 needs block scales with block_size[1] %% 16 == 0 (src/infer.cpp:247)");
  if (c.use_mla && c.q_lora_rank <= 0) DSK_FAIL(DSK_ERR_INVALID, "MLA requires q_lora_rank > 0 (src/infer.cpp:1057)");
  if (c.n_routed_experts > 256) DSK_FAIL(DSK_ERR_UNSUPPORTED, "more than 256 routed experts (src/infer.cpp:527)");
  if (c.rs_original_max_position_embeddings <= KV_SINKS_GUARD) DSK_FAIL(DSK_ERR_INVALID, "rs_original_max_position_embeddings too small");
  if (c.qk_rope_head_dim > 128 || (c.qk_rope_head_dim & 1)) DSK_FAIL(DSK_ERR_UNSUPPORTED, "qk_rope_head_dim %d", c.qk_rope_head_dim);
  dsk_model* m = new dsk_model();
  m->ctx = ctx;
  m->c = c;
  m->head_dim = c.qk_nope_head_dim + c.qk_rope_head_dim;
  m->L.resize(c.n_layers);
  for (int l = 0; l < c.n_layers; ++l) m->L[l].is_moe = c.n_routed_experts > 0 && l >= c.first_k_dense_replace;
  *out = m;
  return DSK_OK;
}

// expected logical shape of a role (src/model.cpp:184-285, 393-457, 557-616, 766-871)
struct RoleShape {
  int quant;      // expected quant
  int e, rows, n; // e = 0: 2-D
  bool ok;
};
static RoleShape role_shape(const dsk_model* m, int role, int layer) {
  const dsk_config& c = m->c;
  const int wq = c.weight_quant, H = c.n_heads, hd = m->head_dim;
  const bool moe = layer >= 0 && m->L[layer].is_moe;
  const int mi = c.moe_intermediate_size;
  switch (role) {
    case DSK_ROLE_EMBED:
    case DSK_ROLE_OUTPUT: return {wq, 0, c.vocab_size, c.dim, true};
    case DSK_ROLE_FINAL_NORM:
    case DSK_ROLE_ATTN_NORM:
    case DSK_ROLE_FFN_NORM: return {DSK_QUANT_F32, 0, 1, c.dim, true};
    case DSK_ROLE_Q_A_NORM: return {DSK_QUANT_F32, 0, 1, c.q_lora_rank, c.q_lora_rank > 0};
    case DSK_ROLE_KV_A_NORM: return {DSK_QUANT_F32, 0, 1, c.kv_lora_rank, true};
    case DSK_ROLE_WQ: return {wq, 0, H * hd, c.dim, !c.use_mla && c.q_lora_rank == 0};
    case DSK_ROLE_WQ_A: return {wq, 0, c.q_lora_rank, c.dim, c.q_lora_rank > 0};
    case DSK_ROLE_WQ_B: return {wq, 0, H * hd, c.q_lora_rank, !c.use_mla && c.q_lora_rank > 0};
    case DSK_ROLE_WKV_A: return {wq, 0, c.kv_lora_rank + c.qk_rope_head_dim, c.dim, true};
    case DSK_ROLE_WKV_B: return {wq, 0, H * (c.qk_nope_head_dim + c.v_head_dim), c.kv_lora_rank, !c.use_mla};
    case DSK_ROLE_WO: return {wq, 0, c.dim, H * c.v_head_dim, true};
    case DSK_ROLE_WC: return {wq, 0, H * c.kv_lora_rank, c.q_lora_rank, (bool)c.use_mla};
    case DSK_ROLE_WQ_ROPE_B: return {wq, 0, H * c.qk_rope_head_dim, c.q_lora_rank, (bool)c.use_mla};
    case DSK_ROLE_WV_B: return {wq, 0, H * c.v_head_dim, c.kv_lora_rank, (bool)c.use_mla};
    case DSK_ROLE_W1:
    case DSK_ROLE_W3: return moe ? RoleShape{wq, c.n_routed_experts, mi, c.dim, true} : RoleShape{wq, 0, c.hidden_dim, c.dim, true};
    case DSK_ROLE_W2: return moe ? RoleShape{wq, c.n_routed_experts, c.dim, mi, true} : RoleShape{wq, 0, c.dim, c.hidden_dim, true};
    case DSK_ROLE_SHARED_W1:
    case DSK_ROLE_SHARED_W3: return {wq, 0, c.n_shared_experts * mi, c.dim, moe && c.n_shared_experts > 0};
    case DSK_ROLE_SHARED_W2: return {wq, 0, c.dim, c.n_shared_experts * mi, moe && c.n_shared_experts > 0};
    case DSK_ROLE_MOEGATE: return {DSK_QUANT_F32, 0, c.n_routed_experts, c.dim, moe};
    case DSK_ROLE_MOEGATE_BIAS: return {DSK_QUANT_F32, 0, 1, c.n_routed_experts, moe && c.has_moegate_bias};
  }
  return {0, 0, 0, 0, false};
}

static const int ALL_LAYER_ROLES[] = {DSK_ROLE_ATTN_NORM, DSK_ROLE_Q_A_NORM, DSK_ROLE_KV_A_NORM, DSK_ROLE_FFN_NORM, DSK_ROLE_WQ,
                                      DSK_ROLE_WQ_A, DSK_ROLE_WQ_B, DSK_ROLE_WKV_A, DSK_ROLE_WKV_B, DSK_ROLE_WO, DSK_ROLE_WC,
                                      DSK_ROLE_WQ_ROPE_B, DSK_ROLE_WV_B, DSK_ROLE_W1, DSK_ROLE_W2, DSK_ROLE_W3,
                                      DSK_ROLE_SHARED_W1, DSK_ROLE_SHARED_W2, DSK_ROLE_SHARED_W3, DSK_ROLE_MOEGATE,
                                      DSK_ROLE_MOEGATE_BIAS};

static bool is_routed_role(int role) { return role == DSK_ROLE_W1 || role == DSK_ROLE_W2 || role == DSK_ROLE_W3; }

// allocate the device planes of a tensor; local = experts kept on this rank
int alloc_tensor(int b0, int b1, DTensor& t, int quant, int e, int rows, int n, int local, int base) {
  t.quant = quant;
  t.n_experts = e;
  t.local_experts = e > 0 ? local : 0;
  t.expert_base = e > 0 ? base : 0;
  t.rows = rows;
  t.n = n;
  const size_t mats = e > 0 ? (size_t)local : 1;
  const size_t per = (size_t)rows * n;
  size_t o_qs = 0, o_sc = 0, o_hm = 0, o_dm = 0, o_scale = 0, total = 0;
  if (quant == DSK_QUANT_Q2_K) {
    const size_t nblk = per / 256;
    t.e_qs = nblk * 64; t.e_sc = nblk * 16; t.e_dm = nblk * 4;
    o_qs = 0; o_sc = align_up(o_qs + mats * t.e_qs, 256); o_dm = align_up(o_sc + mats * t.e_sc, 256);
    total = o_dm + mats * t.e_dm;
  } else if (quant == DSK_QUANT_Q3_K) {
    const size_t nblk = per / 256;
    t.e_qs = nblk * 64; t.e_hm = nblk * 32; t.e_sc = nblk * 12; t.e_dm = nblk * 2;
    o_qs = 0; o_hm = align_up(o_qs + mats * t.e_qs, 256); o_sc = align_up(o_hm + mats * t.e_hm, 256);
    o_dm = align_up(o_sc + mats * t.e_sc, 256);
    total = o_dm + mats * t.e_dm;
  } else {
    t.e_qs = mat_bytes(quant, rows, n);
    total = mats * t.e_qs;
    if (quant == DSK_QUANT_F8E5M2) {
      t.e_scale = (size_t)cdiv(rows, b0) * cdiv(n, b1);
      o_scale = align_up(total, 256);
      total = o_scale + mats * t.e_scale * 4;
    }
  }
  total = align_up(total ? total : 256, 256);
  HIP_TRY(hipMalloc((void**)&t.base, total));
  t.bytes = total;
  t.qs = t.base + o_qs;
  if (quant == DSK_QUANT_Q2_K || quant == DSK_QUANT_Q3_K) {
    t.sc = t.base + o_sc;
    t.dm = t.base + o_dm;
    if (quant == DSK_QUANT_Q3_K) t.hm = t.base + o_hm;
  }
  if (quant == DSK_QUANT_F8E5M2) t.scale = reinterpret_cast<float*>(t.base + o_scale);
  if (e == 0) t.e_qs = t.e_sc = t.e_hm = t.e_dm = 0;  // strides only meaningful for stacks (e_scale kept: scale count)
  return DSK_OK;
}

// copy the local part of a tensor from the reference's host layout into the device planes
int upload_tensor(dsk_ctx* ctx, DTensor& t, const void* host_ptr) {
  hipStream_t st = ctx->stream;
  const size_t per_bytes = mat_bytes(t.quant, t.rows, t.n);
  const size_t lm = t.n_experts > 0 ? (size_t)t.local_experts : 1;
  const char* src = (const char*)host_ptr + (size_t)t.expert_base * per_bytes;
  if (lm == 0) return DSK_OK;
  if (!is_kq(t.quant)) {
    HIP_TRY(hipMemcpyAsync(t.qs, src, lm * per_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSK_OK;
  }
  // K-quants: stage AoS blocks in chunks, re-lay-out into planes on the GPU
  const size_t bsz = t.quant == DSK_QUANT_Q2_K ? 84 : 110;
  const size_t total_blocks = lm * per_bytes / bsz;
  const size_t chunk_blocks = std::min<size_t>(total_blocks, (size_t)(192u << 20) / bsz / 64 * 64);
  void* stage;
  DSK_TRY(ctx_scratch(ctx, 7, chunk_blocks * bsz + 256, &stage));
  for (size_t b0 = 0; b0 < total_blocks; b0 += chunk_blocks) {
    const size_t nb = std::min(chunk_blocks, total_blocks - b0);
    HIP_TRY(hipMemcpyAsync(stage, src + b0 * bsz, nb * bsz, hipMemcpyHostToDevice, st));
    if (t.quant == DSK_QUANT_Q2_K)
      DSK_TRY(launch_repack_q2k(st, (const uint8_t*)stage, nb, t.qs + b0 * 64, t.sc + b0 * 16, t.dm + b0 * 4));
    else
      DSK_TRY(launch_repack_q3k(st, (const uint8_t*)stage, nb, t.qs + b0 * 64, t.hm + b0 * 32, t.sc + b0 * 12, t.dm + b0 * 2));
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipGetLastError());
  return DSK_OK;
}

static int tensor_slot(dsk_model* m, int role, int layer, DTensor** out) {
  if (role < 10) {
    if (role > 2 || layer != -1) DSK_FAIL(DSK_ERR_INVALID, "bind: model-level role %d needs layer -1", role);
    *out = &m->g[role];
  } else {
    if (role >= NROLES || layer < 0 || layer >= m->c.n_layers) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d layer %d", role, layer);
    *out = &m->L[layer].t[role];
  }
  return DSK_OK;
}

static void shard_range(const dsk_model* m, int role, int e, int* base, int* local) {
  *base = 0;
  *local = e;
  if (e > 0 && is_routed_role(role) && m->ctx->world > 1) {
    const int per = cdiv(e, m->ctx->world);
    *base = std::min(e, m->ctx->rank * per);
    *local = std::max(0, std::min(per, e - *base));
  }
}

extern "C" int dsk_model_bind(dsk_model* m, int role, int layer, int quant, const int32_t shape[4], const void* host_ptr, size_t bytes) {
  if (!m || !host_ptr || !shape) DSK_FAIL(DSK_ERR_INVALID, "bind: null argument");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "bind after finalize");
  HIP_TRY(hipSetDevice(m->ctx->device));
  const bool is_scale = role >= DSK_ROLE_SCALE;
  const int r = is_scale ? role - DSK_ROLE_SCALE : role;
  DTensor* t;
  DSK_TRY(tensor_slot(m, r, layer, &t));
  const RoleShape rs = role_shape(m, r, layer);
  if (!rs.ok) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d is not part of this configuration (layer %d)", r, layer);
  const dsk_config& c = m->c;
  hipStream_t st = m->ctx->stream;
  int base, local;
  shard_range(m, r, rs.e, &base, &local);

  if (is_scale) {
    if (rs.quant != DSK_QUANT_F8E5M2) DSK_FAIL(DSK_ERR_INVALID, "bind: scale tensor for a non-f8e5m2 weight");
    if (!t->bound()) DSK_FAIL(DSK_ERR_STATE, "bind: scale before its weight (role %d layer %d)", r, layer);
    const size_t per = (size_t)cdiv(rs.rows, c.block_size[0]) * cdiv(rs.n, c.block_size[1]);
    const size_t mats = rs.e > 0 ? rs.e : 1;
    if (bytes != per * mats * 4) DSK_FAIL(DSK_ERR_INVALID, "bind: scale bytes %zu, expected %zu", bytes, per * mats * 4);
    const size_t lm = rs.e > 0 ? (size_t)local : 1;
    if (lm) HIP_TRY(hipMemcpyAsync(t->scale, (const char*)host_ptr + (size_t)base * per * 4, lm * per * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSK_OK;
  }

  if (quant != rs.quant) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d layer %d has quant %d, expected %d", r, layer, quant, rs.quant);
  // logical shape check (QTensor::from_codec_tensor, src/codec.cpp:166-234)
  int exp_shape[4] = {0, 0, 0, 0};
  if (rs.e > 0) { exp_shape[0] = rs.e; exp_shape[1] = rs.rows; exp_shape[2] = rs.n; }
  else if (rs.rows == 1 && rs.quant == DSK_QUANT_F32 && r != DSK_ROLE_MOEGATE) { exp_shape[0] = rs.n; }
  else { exp_shape[0] = rs.rows; exp_shape[1] = rs.n; }
  for (int i = 0; i < 4; ++i)
    if (shape[i] != exp_shape[i])
      DSK_FAIL(DSK_ERR_INVALID, "bind: role %d layer %d shape [%d,%d,%d,%d], expected [%d,%d,%d,%d]", r, layer, shape[0], shape[1],
               shape[2], shape[3], exp_shape[0], exp_shape[1], exp_shape[2], exp_shape[3]);
  if (is_kq(quant) && rs.n % 256) DSK_FAIL(DSK_ERR_INVALID, "bind: k-quant row length %d is not a multiple of 256", rs.n);
  const size_t mats = rs.e > 0 ? rs.e : 1;
  const size_t per_bytes = mat_bytes(quant, rs.rows, rs.n);
  if (bytes != per_bytes * mats) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d layer %d has %zu bytes, expected %zu", r, layer, bytes, per_bytes * mats);
  if (t->bound()) DSK_FAIL(DSK_ERR_STATE, "bind: role %d layer %d bound twice", r, layer);
  DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], *t, quant, rs.e, rs.rows, rs.n, local, base));
  m->weight_bytes += (double)t->bytes;
  return upload_tensor(m->ctx, *t, host_ptr);
}

extern "C" int dsk_model_synthesize(dsk_model* m, uint64_t seed) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "synthesize: null model");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "synthesize after finalize");
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const dsk_config& c = m->c;
  auto one = [&](int role, int layer) -> int {
    const RoleShape rs = role_shape(m, role, layer);
    if (!rs.ok) return DSK_OK;
    DTensor* t;
    DSK_TRY(tensor_slot(m, role, layer, &t));
    if (t->bound()) return DSK_OK;
    if (is_kq(rs.quant) && rs.n % 256) DSK_FAIL(DSK_ERR_INVALID, "synthesize: k-quant row length %d", rs.n);
    int base, local;
    shard_range(m, role, rs.e, &base, &local);
    DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], *t, rs.quant, rs.e, rs.rows, rs.n, local, base));
    m->weight_bytes += (double)t->bytes;
    // the seed depends on (role, layer, first local expert) only: every rank generates the same weights
    const uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(layer + 1) * 1000003ull + (uint64_t)role * 7919ull;
    const bool norm = role == DSK_ROLE_FINAL_NORM || role == DSK_ROLE_ATTN_NORM || role == DSK_ROLE_FFN_NORM ||
                      role == DSK_ROLE_Q_A_NORM || role == DSK_ROLE_KV_A_NORM;
    if (norm) return launch_fill_f32(st, reinterpret_cast<float*>(t->qs), (size_t)rs.n, s, 1.0f, 0.1f);
    if (role == DSK_ROLE_MOEGATE_BIAS) return launch_fill_f32(st, reinterpret_cast<float*>(t->qs), (size_t)rs.n, s, 0.0f, 0.1f);
    float wscale = 1.0f / sqrtf((float)rs.n);
    if (role == DSK_ROLE_EMBED) wscale = 1.0f;  // residual stream starts at unit scale
    if (rs.e > 0 && local < rs.e) {
      // sharded stack: fill expert by expert so that expert e gets the same bytes on any world size
      DTensor one_e = *t;
      for (int le = 0; le < local; ++le) {
        one_e.local_experts = 1;
        one_e.qs = t->qs + (size_t)le * t->e_qs;
        one_e.sc = t->sc ? t->sc + (size_t)le * t->e_sc : nullptr;
        one_e.hm = t->hm ? t->hm + (size_t)le * t->e_hm : nullptr;
        one_e.dm = t->dm ? t->dm + (size_t)le * t->e_dm : nullptr;
        one_e.scale = t->scale ? t->scale + (size_t)le * t->e_scale : nullptr;
        DSK_TRY(launch_fill_tensor(st, one_e, s + (uint64_t)(base + le) * 0x100000001B3ull, wscale));
      }
      return DSK_OK;
    }
    if (rs.e > 0) {
      DTensor one_e = *t;
      for (int le = 0; le < rs.e; ++le) {
        one_e.local_experts = 1;
        one_e.qs = t->qs + (size_t)le * t->e_qs;
        one_e.sc = t->sc ? t->sc + (size_t)le * t->e_sc : nullptr;
        one_e.hm = t->hm ? t->hm + (size_t)le * t->e_hm : nullptr;
        one_e.dm = t->dm ? t->dm + (size_t)le * t->e_dm : nullptr;
        one_e.scale = t->scale ? t->scale + (size_t)le * t->e_scale : nullptr;
        DSK_TRY(launch_fill_tensor(st, one_e, s + (uint64_t)le * 0x100000001B3ull, wscale));
      }
      return DSK_OK;
    }
    return launch_fill_tensor(st, *t, s, wscale);
  };
  DSK_TRY(one(DSK_ROLE_EMBED, -1));
  DSK_TRY(one(DSK_ROLE_FINAL_NORM, -1));
  DSK_TRY(one(DSK_ROLE_OUTPUT, -1));
  for (int l = 0; l < c.n_layers; ++l)
    for (int role : ALL_LAYER_ROLES) DSK_TRY(one(role, l));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return DSK_OK;
}

static int alloc_f(dsk_model* m, float** p, size_t n) {
  HIP_TRY(hipMalloc((void**)p, std::max<size_t>(n, 1) * 4));
  HIP_TRY(hipMemset(*p, 0, std::max<size_t>(n, 1) * 4));
  m->scratch_bytes += (double)n * 4;
  return DSK_OK;
}
static int alloc_q8(dsk_model* m, Q8Buf& b, int n) {
  n = std::max(256, (n + 255) / 256 * 256);
  b.cap = n;
  HIP_TRY(hipMalloc((void**)&b.qs, n));
  HIP_TRY(hipMalloc((void**)&b.d, n / 256 * 4));
  HIP_TRY(hipMalloc((void**)&b.bsums, n / 16 * 2));
  HIP_TRY(hipMemset(b.qs, 0, n));
  HIP_TRY(hipMemset(b.d, 0, n / 256 * 4));
  HIP_TRY(hipMemset(b.bsums, 0, n / 16 * 2));
  m->scratch_bytes += n * 1.15;
  return DSK_OK;
}

extern "C" int dsk_model_finalize(dsk_model* m) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "finalize: null model");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "finalize twice");
  HIP_TRY(hipSetDevice(m->ctx->device));
  const dsk_config& c = m->c;
  const int H = c.n_heads, hd = m->head_dim;
  if (!m->g[DSK_ROLE_EMBED].bound() || !m->g[DSK_ROLE_FINAL_NORM].bound()) DSK_FAIL(DSK_ERR_STATE, "finalize: embed / final norm not bound");
  m->tied = !m->g[DSK_ROLE_OUTPUT].bound();  // src/model.cpp:852-856
  for (int l = 0; l < c.n_layers; ++l)
    for (int role : ALL_LAYER_ROLES) {
      const RoleShape rs = role_shape(m, role, l);
      if (rs.ok && !m->L[l].t[role].bound()) DSK_FAIL(DSK_ERR_STATE, "finalize: layer %d role %d not bound", l, role);
      if (rs.ok && rs.quant == DSK_QUANT_F8E5M2 && !m->L[l].t[role].scale) DSK_FAIL(DSK_ERR_STATE, "finalize: layer %d role %d has no scale", l, role);
    }
  if (is_kq(c.weight_quant)) {
    const int lens[] = {c.dim, c.q_lora_rank, c.kv_lora_rank, H * c.v_head_dim, c.hidden_dim,
                        c.n_routed_experts > 0 ? c.moe_intermediate_size : 0,
                        c.n_routed_experts > 0 ? c.n_shared_experts * c.moe_intermediate_size : 0};
    for (int v : lens)
      if (v % 256) DSK_FAIL(DSK_ERR_INVALID, "finalize: k-quant model with a GEMV input length %d not divisible by 256", v);
  }
  const size_t seq = (size_t)c.max_seq_len;
  for (int l = 0; l < c.n_layers; ++l) {
    Layer& L = m->L[l];
    if (c.use_mla) {
      HIP_TRY(hipMalloc((void**)&L.nope_cache, seq * c.kv_lora_rank * 2));
      HIP_TRY(hipMalloc((void**)&L.rope_cache, seq * c.qk_rope_head_dim * 2));
      HIP_TRY(hipMemset(L.nope_cache, 0, seq * c.kv_lora_rank * 2));
      HIP_TRY(hipMemset(L.rope_cache, 0, seq * c.qk_rope_head_dim * 2));
      m->cache_bytes += (double)seq * (c.kv_lora_rank + c.qk_rope_head_dim) * 2;
    } else {
      HIP_TRY(hipMalloc((void**)&L.key_cache, seq * H * hd * 2));
      HIP_TRY(hipMalloc((void**)&L.value_cache, seq * H * c.v_head_dim * 2));
      HIP_TRY(hipMemset(L.key_cache, 0, seq * H * hd * 2));
      HIP_TRY(hipMemset(L.value_cache, 0, seq * H * c.v_head_dim * 2));
      m->cache_bytes += (double)seq * H * (hd + c.v_head_dim) * 2;
    }
  }
  const int K = std::max(1, c.n_active_routed);
  m->n_slots = c.n_routed_experts > 0 ? c.n_active_routed + (c.n_shared_experts > 0 ? 1 : 0) : 0;
  const int slots = std::max(1, m->n_slots);
  const int shared_n = c.n_shared_experts * c.moe_intermediate_size;
  const int hb_stride = std::max(std::max(c.moe_intermediate_size, shared_n), 1);
  const size_t hb_n = std::max<size_t>((size_t)slots * hb_stride, (size_t)c.hidden_dim);
  DSK_TRY(alloc_f(m, &m->x, c.dim));
  DSK_TRY(alloc_f(m, &m->xb, c.dim));
  DSK_TRY(alloc_f(m, &m->q_a, std::max(1, c.q_lora_rank)));
  DSK_TRY(alloc_f(m, &m->q, (size_t)H * hd));
  DSK_TRY(alloc_f(m, &m->kv_a, c.kv_lora_rank + c.qk_rope_head_dim));
  DSK_TRY(alloc_f(m, &m->kv_b, (size_t)H * (c.qk_nope_head_dim + c.v_head_dim)));
  DSK_TRY(alloc_f(m, &m->att_out, (size_t)H * std::max(c.v_head_dim, c.use_mla ? c.kv_lora_rank : 0)));
  DSK_TRY(alloc_f(m, &m->hb, hb_n));
  DSK_TRY(alloc_f(m, &m->eout, (size_t)slots * c.dim));
  DSK_TRY(alloc_f(m, &m->q_c, (size_t)H * std::max(1, c.kv_lora_rank)));
  DSK_TRY(alloc_f(m, &m->q_rope, (size_t)H * std::max(1, c.qk_rope_head_dim)));
  DSK_TRY(alloc_f(m, &m->vb_out, (size_t)H * c.v_head_dim));
  m->router_ksplit = c.n_routed_experts > 0 ? std::max(1, std::min(8, 1024 / std::max(1, c.n_routed_experts))) : 1;
  while (m->router_ksplit > 1 && c.dim / m->router_ksplit < 256) m->router_ksplit /= 2;
  DSK_TRY(alloc_f(m, &m->router_partial, (size_t)m->router_ksplit * std::max(1, c.n_routed_experts)));
  DSK_TRY(alloc_f(m, &m->gate_scores, (size_t)c.n_layers * std::max(1, c.n_routed_experts)));
  DSK_TRY(alloc_f(m, &m->logits, c.vocab_size));
  DSK_TRY(alloc_f(m, &m->trace_x, (size_t)c.n_layers * c.dim));
  DSK_TRY(alloc_f(m, &m->route_w, (size_t)c.n_layers * K));
  HIP_TRY(hipMalloc((void**)&m->route_e, (size_t)c.n_layers * K * 4));
  HIP_TRY(hipMemset(m->route_e, 0xff, (size_t)c.n_layers * K * 4));
  if (is_kq(c.weight_quant)) {
    DSK_TRY(alloc_q8(m, m->a_xb, c.dim));
    DSK_TRY(alloc_q8(m, m->a_qa, std::max(256, c.q_lora_rank)));
    DSK_TRY(alloc_q8(m, m->a_kva, c.kv_lora_rank));
    DSK_TRY(alloc_q8(m, m->a_att, H * std::max(c.v_head_dim, c.use_mla ? c.kv_lora_rank : 0)));
    DSK_TRY(alloc_q8(m, m->a_hb, (int)hb_n));
  }
  HIP_TRY(hipMalloc((void**)&m->sp_dev, sizeof(StepParams)));
  HIP_TRY(hipHostMalloc((void**)&m->sp_host, sizeof(StepParams), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&m->logits_host, (size_t)c.vocab_size * 4, hipHostMallocDefault));
  memset(m->sp_host, 0, sizeof(StepParams));
  HIP_TRY(hipDeviceSynchronize());
  m->finalized = true;
  return DSK_OK;
}

static void free_tensor(DTensor& t) {
  if (t.base) hipFree(t.base);
  t = DTensor();
}
static void free_q8(Q8Buf& b) {
  if (b.qs) hipFree(b.qs);
  if (b.d) hipFree(b.d);
  if (b.bsums) hipFree(b.bsums);
}
extern "C" int dsk_model_destroy(dsk_model* m) {
  if (!m) return DSK_OK;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  for (int i = 0; i < 2; ++i)
    if (m->graph[i]) hipGraphExecDestroy(m->graph[i]);
  for (int i = 0; i < 3; ++i)
    if (!(i == DSK_ROLE_OUTPUT && m->tied && m->finalized)) free_tensor(m->g[i]);
  for (auto& L : m->L) {
    for (auto& t : L.t) free_tensor(t);
    for (void* p : {(void*)L.key_cache, (void*)L.value_cache, (void*)L.nope_cache, (void*)L.rope_cache})
      if (p) hipFree(p);
  }
  for (void* p : {(void*)m->x, (void*)m->xb, (void*)m->q_a, (void*)m->q, (void*)m->kv_a, (void*)m->kv_b, (void*)m->att_out,
                  (void*)m->hb, (void*)m->eout, (void*)m->q_c, (void*)m->q_rope, (void*)m->vb_out, (void*)m->router_partial,
                  (void*)m->gate_scores, (void*)m->logits, (void*)m->trace_x, (void*)m->route_e, (void*)m->route_w,
                  (void*)m->sp_dev})
    if (p) hipFree(p);
  free_q8(m->a_xb); free_q8(m->a_qa); free_q8(m->a_kva); free_q8(m->a_att); free_q8(m->a_hb);
  if (m->sp_host) hipHostFree(m->sp_host);
  if (m->logits_host) hipHostFree(m->logits_host);
  for (auto& k : m->ktimes)
    for (auto& e : k.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  delete m;
  return DSK_OK;
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
struct Prof {  // RAII-free bracket: records events around a launch when profiling
  dsk_model* m;
  int idx = -1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
static int prof_begin(dsk_model* m, const char* name, double bytes, Prof* p) {
  p->m = m;
  if (!m->profiling) return DSK_OK;
  auto it = m->kindex.find(name);
  if (it == m->kindex.end()) {
    m->kindex[name] = (int)m->ktimes.size();
    KTime k;
    k.name = name;
    m->ktimes.push_back(k);
    it = m->kindex.find(name);
  }
  p->idx = it->second;
  KTime& k = m->ktimes[p->idx];
  k.launches++;
  k.algo_bytes += bytes;
  HIP_TRY(hipEventCreate(&p->e0));
  HIP_TRY(hipEventCreate(&p->e1));
  HIP_TRY(hipEventRecord(p->e0, m->ctx->stream));
  return DSK_OK;
}
static int prof_end(Prof* p) {
  if (!p->m->profiling) return DSK_OK;
  HIP_TRY(hipEventRecord(p->e1, p->m->ctx->stream));
  p->m->ktimes[p->idx].ev.push_back({p->e0, p->e1});
  return DSK_OK;
}
#define PROFILED(name, bytes, call) \
  do {                              \
    Prof _p;                        \
    DSK_TRY(prof_begin(m, name, (double)(bytes), &_p)); \
    DSK_TRY(call);                  \
    DSK_TRY(prof_end(&_p));         \
  } while (0)

// activation bytes of a GEMV input/outputs (SURVEY 8d per-GEMV unit)
static double act_bytes(int quant, int n, int rows) {
  return (is_kq(quant) ? (double)n / 256 * 292 : (double)n * 4) + 4.0 * rows;
}
static double tensor_bytes_2d(const DTensor& t) {
  double b = (double)mat_bytes(t.quant, t.rows, t.n);
  if (t.quant == DSK_QUANT_F8E5M2) b += 4.0 * t.e_scale;
  return b;
}

static void seg_from(const dsk_model* m, GemvSeg& s, const DTensor& t) {
  memset(&s, 0, sizeof s);
  s.qs = t.qs; s.sc = t.sc; s.hm = t.hm; s.dm = t.dm; s.scale = t.scale;
  s.rows = t.rows; s.n = t.n; s.n_slots = 1;
  s.b0 = std::max(1, m->c.block_size[0]); s.b1 = std::max(1, m->c.block_size[1]);
  s.sc_cols = cdiv(t.n, s.b1);
  s.local_experts = 1;
}
static void seg_act(GemvSeg& s, int quant, const Q8Buf& q8, const float* f32) {
  if (is_kq(quant)) { s.a_qs = q8.qs; s.a_d = q8.d; s.a_bsums = q8.bsums; }
  else s.a_f32 = f32;
}

// plain W(d,n).x -> out (matmul, src/infer.cpp:381-417)
static int run_gemv(dsk_model* m, const char* name, const DTensor& t, const Q8Buf& q8, const float* f32, float* out, int epilogue) {
  GemvSeg s;
  seg_from(m, s, t);
  seg_act(s, t.quant, q8, f32);
  s.out = out;
  s.epilogue = epilogue;
  PROFILED(name, tensor_bytes_2d(t) + act_bytes(t.quant, t.n, t.rows), launch_gemv(m->ctx->stream, t.quant, s));
  return DSK_OK;
}
// fused pair silu(W1 x) * W3 x (src/infer.cpp:882-897, 910-925)
static int run_glu(dsk_model* m, const char* name, const DTensor& t1, const DTensor& t3, const Q8Buf& q8, const float* f32, float* out) {
  GemvSeg s;
  seg_from(m, s, t1);
  s.qs2 = t3.qs; s.sc2 = t3.sc; s.hm2 = t3.hm; s.dm2 = t3.dm; s.scale2 = t3.scale;
  seg_act(s, t1.quant, q8, f32);
  s.out = out;
  s.epilogue = EPI_GLU;
  s.act = m->c.act;
  PROFILED(name, 2 * tensor_bytes_2d(t1) + act_bytes(t1.quant, t1.n, t1.rows), launch_gemv(m->ctx->stream, t1.quant, s));
  return DSK_OK;
}
// quantise an f32 vector to Q8_K when the model is K-quantised (src/infer.cpp:325-337)
static int run_quant(dsk_model* m, const float* x, int n, Q8Buf& q8) {
  if (!is_kq(m->c.weight_quant)) return DSK_OK;
  PROFILED("quantize_q8k", (double)n * 5.2, launch_quantize_q8k(m->ctx->stream, x, n, q8.qs, q8.d, q8.bsums));
  return DSK_OK;
}

static int fill_step_params(dsk_model* m, int token, int pos) {
  const dsk_config& c = m->c;
  StepParams* sp = m->sp_host;
  const int W = c.rs_original_max_position_embeddings;
  sp->token = token;
  sp->pos = pos;
  sp->kv_sink = pos >= W ? 2 : 0;  // KV_SINKS, src/model.h:14; ring arithmetic src/infer.cpp:1274-1277
  sp->kv_pos = sp->kv_sink + (pos - sp->kv_sink) % (W - sp->kv_sink);
  sp->kv_len = pos >= W ? W : pos + 1;
  if (sp->kv_pos >= c.max_seq_len || sp->kv_len > c.max_seq_len)
    DSK_FAIL(DSK_ERR_INVALID, "forward: pos %d exceeds the max_seq_len=%d allocation (the reference overruns its cache here)", pos, c.max_seq_len);
  const int rd = c.qk_rope_head_dim;
  for (int j = 0; j < rd / 2; ++j) {  // same libm calls as src/infer.cpp:655-658
    // 1/powf(theta, j/d) as the reference's -ffast-math build evaluates it (see oracle/dsk_oracle.c ref_rope_freq)
    const float freq = powf(c.rope_theta, -((float)(2 * j) * (1.0f / (float)rd)));
    const float v = pos * freq, v1 = 1 * freq;
    sp->rope_cs[2 * j] = cosf(v);
    sp->rope_cs[2 * j + 1] = sinf(v);
    sp->rope_cs1[2 * j] = cosf(v1);
    sp->rope_cs1[2 * j + 1] = sinf(v1);
  }
  return DSK_OK;
}

static int attention_mha(dsk_model* m, Layer& L, int max_kv) {
  const dsk_config& c = m->c;
  hipStream_t st = m->ctx->stream;
  const int H = c.n_heads, hd = m->head_dim;
  const bool kq = is_kq(c.weight_quant);
  // q / kv_a projections
  if (c.q_lora_rank > 0) {
    DSK_TRY(run_gemv(m, "gemv_wq_a", L.t[DSK_ROLE_WQ_A], m->a_xb, m->xb, m->q_a, EPI_STORE));
  } else {
    DSK_TRY(run_gemv(m, "gemv_wq", L.t[DSK_ROLE_WQ], m->a_xb, m->xb, m->q, EPI_STORE));
  }
  DSK_TRY(run_gemv(m, "gemv_wkv_a", L.t[DSK_ROLE_WKV_A], m->a_xb, m->xb, m->kv_a, EPI_STORE));
  // norms (+Q8) of q_a and the latent part of kv_a (src/infer.cpp:946,974)
  NormJob jobs[2];
  int nj = 0;
  memset(jobs, 0, sizeof jobs);
  if (c.q_lora_rank > 0) {
    NormJob& j = jobs[nj++];
    j.x = m->q_a; j.weight = reinterpret_cast<const float*>(L.t[DSK_ROLE_Q_A_NORM].qs); j.n = c.q_lora_rank; j.eps = c.norm_eps;
    j.y_f32 = m->q_a;
    if (kq) { j.q_qs = m->a_qa.qs; j.q_d = m->a_qa.d; j.q_bsums = m->a_qa.bsums; }
  }
  {
    NormJob& j = jobs[nj++];
    j.x = m->kv_a; j.weight = reinterpret_cast<const float*>(L.t[DSK_ROLE_KV_A_NORM].qs); j.n = c.kv_lora_rank; j.eps = c.norm_eps;
    j.y_f32 = m->kv_a;
    if (kq) { j.q_qs = m->a_kva.qs; j.q_d = m->a_kva.d; j.q_bsums = m->a_kva.bsums; }
  }
  PROFILED("norm_q8", (double)(c.q_lora_rank + c.kv_lora_rank) * 9, launch_norm_jobs(st, jobs, nj, m->sp_dev));
  if (c.q_lora_rank > 0) DSK_TRY(run_gemv(m, "gemv_wq_b", L.t[DSK_ROLE_WQ_B], m->a_qa, m->q_a, m->q, EPI_STORE));
  DSK_TRY(run_gemv(m, "gemv_wkv_b", L.t[DSK_ROLE_WKV_B], m->a_kva, m->kv_a, m->kv_b, EPI_STORE));
  AttnMhaArgs a;
  a.q = m->q; a.kv_b = m->kv_b; a.kv_a = m->kv_a; a.key_cache = L.key_cache; a.value_cache = L.value_cache; a.out = m->att_out;
  a.n_heads = H; a.head_dim = hd; a.nope = c.qk_nope_head_dim; a.rope = c.qk_rope_head_dim; a.v_dim = c.v_head_dim;
  a.lora = c.kv_lora_rank; a.is_v3 = c.has_moegate_bias;
  PROFILED("rope_kv", (double)H * (hd * 6 + c.v_head_dim * 6), launch_rope_kv_mha(st, a, m->sp_dev));
  PROFILED("attn_mha", (double)m->sp_host->kv_len * H * (hd + c.v_head_dim) * 2, launch_attn_mha(st, a, m->sp_dev, 0, max_kv));
  DSK_TRY(run_quant(m, m->att_out, H * c.v_head_dim, m->a_att));
  DSK_TRY(run_gemv(m, "gemv_wo", L.t[DSK_ROLE_WO], m->a_att, m->att_out, m->x, EPI_ADD));  // residual: src/infer.cpp:832-834
  return DSK_OK;
}

static int attention_mla(dsk_model* m, Layer& L, int max_kv) {
  const dsk_config& c = m->c;
  hipStream_t st = m->ctx->stream;
  const int H = c.n_heads;
  const bool kq = is_kq(c.weight_quant);
  DSK_TRY(run_gemv(m, "gemv_wq_a", L.t[DSK_ROLE_WQ_A], m->a_xb, m->xb, m->q_a, EPI_STORE));
  DSK_TRY(run_gemv(m, "gemv_wkv_a", L.t[DSK_ROLE_WKV_A], m->a_xb, m->xb, m->kv_a, EPI_STORE));
  NormJob jobs[2];
  memset(jobs, 0, sizeof jobs);
  {
    NormJob& j = jobs[0];
    j.x = m->q_a; j.weight = reinterpret_cast<const float*>(L.t[DSK_ROLE_Q_A_NORM].qs); j.n = c.q_lora_rank; j.eps = c.norm_eps;
    j.y_f32 = m->q_a;
    if (kq) { j.q_qs = m->a_qa.qs; j.q_d = m->a_qa.d; j.q_bsums = m->a_qa.bsums; }
  }
  {
    NormJob& j = jobs[1];
    j.x = m->kv_a; j.weight = reinterpret_cast<const float*>(L.t[DSK_ROLE_KV_A_NORM].qs); j.n = c.kv_lora_rank; j.eps = c.norm_eps;
    j.y_f32 = m->kv_a;
  }
  PROFILED("norm_q8", (double)(c.q_lora_rank + c.kv_lora_rank) * 9, launch_norm_jobs(st, jobs, 2, m->sp_dev));
  DSK_TRY(run_gemv(m, "gemv_wq_rope_b", L.t[DSK_ROLE_WQ_ROPE_B], m->a_qa, m->q_a, m->q_rope, EPI_STORE));
  DSK_TRY(run_gemv(m, "gemv_wc", L.t[DSK_ROLE_WC], m->a_qa, m->q_a, m->q_c, EPI_STORE));
  AttnMlaArgs a;
  a.q_rope = m->q_rope; a.q_c = m->q_c; a.kv_a = m->kv_a; a.nope_cache = L.nope_cache; a.rope_cache = L.rope_cache; a.out = m->att_out;
  a.n_heads = H; a.head_dim = m->head_dim; a.rope = c.qk_rope_head_dim; a.lora = c.kv_lora_rank; a.is_v3 = c.has_moegate_bias;
  PROFILED("rope_kv", (double)H * c.qk_rope_head_dim * 8 + c.kv_lora_rank * 6, launch_rope_kv_mla(st, a, m->sp_dev));
  PROFILED("attn_mla", (double)m->sp_host->kv_len * (c.kv_lora_rank + c.qk_rope_head_dim) * 2, launch_attn_mla(st, a, m->sp_dev, 0, max_kv));
  // per-head wv_b on the per-head latent outputs (src/infer.cpp:1134-1137): a block-diagonal GEMV
  DSK_TRY(run_quant(m, m->att_out, H * c.kv_lora_rank, m->a_att));
  {
    const DTensor& t = L.t[DSK_ROLE_WV_B];
    GemvSeg s;
    seg_from(m, s, t);
    s.rows = c.v_head_dim; s.n = c.kv_lora_rank; s.n_slots = H;
    // view (H*v, lora) as H stacked (v, lora) matrices
    const size_t per = (size_t)c.v_head_dim * c.kv_lora_rank;
    if (t.quant == DSK_QUANT_Q2_K) { s.e_qs = per / 256 * 64; s.e_sc = per / 256 * 16; s.e_dm = per / 256 * 4; }
    else if (t.quant == DSK_QUANT_Q3_K) { s.e_qs = per / 256 * 64; s.e_hm = per / 256 * 32; s.e_sc = per / 256 * 12; s.e_dm = per / 256 * 2; }
    else {
      s.e_qs = mat_bytes(t.quant, c.v_head_dim, c.kv_lora_rank);
      // reference indexing: expert_index * cdiv(d,b0)*cdiv(n,b1) (src/infer.cpp:437-438)
      s.e_scale = (size_t)cdiv(c.v_head_dim, s.b0) * cdiv(c.kv_lora_rank, s.b1);
    }
    s.expert_base = 0; s.local_experts = H;
    seg_act(s, t.quant, m->a_att, m->att_out);
    s.a_slot_stride = c.kv_lora_rank;
    s.out = m->vb_out; s.out_slot_stride = c.v_head_dim;
    s.sc_cols = cdiv(c.kv_lora_rank, s.b1);
    PROFILED("gemv_wv_b", tensor_bytes_2d(t) + act_bytes(t.quant, H * c.kv_lora_rank, H * c.v_head_dim), launch_gemv(st, t.quant, s));
  }
  DSK_TRY(run_quant(m, m->vb_out, H * c.v_head_dim, m->a_att));
  DSK_TRY(run_gemv(m, "gemv_wo", L.t[DSK_ROLE_WO], m->a_att, m->vb_out, m->x, EPI_ADD));
  return DSK_OK;
}

// MoE / dense FFN of one block (src/infer.cpp:844-931).  Leaves the routed + shared expert
// outputs in m->eout for the NEXT norm kernel to fold into x (pending_combine).
static int ffn(dsk_model* m, int l, bool* pending_combine) {
  const dsk_config& c = m->c;
  Layer& L = m->L[l];
  hipStream_t st = m->ctx->stream;
  *pending_combine = false;
  if (!L.is_moe) {
    DSK_TRY(run_glu(m, "gemv_dense_w13", L.t[DSK_ROLE_W1], L.t[DSK_ROLE_W3], m->a_xb, m->xb, m->hb));
    DSK_TRY(run_quant(m, m->hb, c.hidden_dim, m->a_hb));
    DSK_TRY(run_gemv(m, "gemv_dense_w2", L.t[DSK_ROLE_W2], m->a_hb, m->hb, m->x, EPI_ADD));
    return DSK_OK;
  }
  const int K = c.n_active_routed, E = c.n_routed_experts, mi = c.moe_intermediate_size;
  const int shared_n = c.n_shared_experts * mi;
  const int hb_stride = std::max(mi, shared_n);
  int* ae = m->route_e + (size_t)l * K;
  float* aw = m->route_w + (size_t)l * K;
  // router (always F32, src/model.cpp:196-198) + gate
  PROFILED("router_gemv", (double)E * c.dim * 4, launch_router(st, reinterpret_cast<const float*>(L.t[DSK_ROLE_MOEGATE].qs), m->xb, E, c.dim,
                                                               m->router_partial, m->router_ksplit));
  const float* bias = L.t[DSK_ROLE_MOEGATE_BIAS].bound() ? reinterpret_cast<const float*>(L.t[DSK_ROLE_MOEGATE_BIAS].qs) : nullptr;
  PROFILED("moe_gate", (double)E * 8, launch_gate(st, m->router_partial, m->router_ksplit, bias, E, K, c.norm_topk_prob, c.routed_scaling_factor,
                                                  c.scoring_func, c.topk_method, c.n_group, c.topk_group, ae, aw,
                                                  m->gate_scores + (size_t)l * E));
  const DTensor &w1 = L.t[DSK_ROLE_W1], &w2 = L.t[DSK_ROLE_W2], &w3 = L.t[DSK_ROLE_W3];
  const double e_bytes = (double)mat_bytes(w1.quant, mi, c.dim) + (w1.quant == DSK_QUANT_F8E5M2 ? 4.0 * w1.e_scale : 0.0);
  if (m->ctx->world > 1) HIP_TRY(hipMemsetAsync(m->eout, 0, (size_t)K * c.dim * 4, st));
  {  // routed W1/W3 + activation: K slots, expert ids read on the device
    GemvSeg s;
    seg_from(m, s, w1);
    s.qs2 = w3.qs; s.sc2 = w3.sc; s.hm2 = w3.hm; s.dm2 = w3.dm; s.scale2 = w3.scale;
    s.e_qs = w1.e_qs; s.e_sc = w1.e_sc; s.e_hm = w1.e_hm; s.e_dm = w1.e_dm; s.e_scale = w1.e_scale;
    s.expert_ids = ae; s.expert_base = w1.expert_base; s.local_experts = w1.local_experts; s.n_slots = K;
    seg_act(s, w1.quant, m->a_xb, m->xb);
    s.out = m->hb; s.out_slot_stride = hb_stride; s.epilogue = EPI_GLU; s.act = c.act;
    PROFILED("gemv_experts_w13", 2.0 * K * e_bytes + act_bytes(w1.quant, c.dim, 0) + 4.0 * K * mi, launch_gemv(st, w1.quant, s));
  }
  if (c.n_shared_experts > 0)
    DSK_TRY(run_glu(m, "gemv_shared_w13", L.t[DSK_ROLE_SHARED_W1], L.t[DSK_ROLE_SHARED_W3], m->a_xb, m->xb, m->hb + (size_t)K * hb_stride));
  // Q8 of every slot's hidden vector (routed: mi each; shared: shared_n)
  if (is_kq(c.weight_quant)) {
    const int total = K * hb_stride + (c.n_shared_experts > 0 ? shared_n : 0);
    // the slots are laid out back to back with stride hb_stride (a multiple of 256), so one pass covers them
    DSK_TRY(run_quant(m, m->hb, (total + 255) / 256 * 256, m->a_hb));
  }
  {  // routed W2 -> eout[k]
    GemvSeg s;
    seg_from(m, s, w2);
    s.e_qs = w2.e_qs; s.e_sc = w2.e_sc; s.e_hm = w2.e_hm; s.e_dm = w2.e_dm; s.e_scale = w2.e_scale;
    s.expert_ids = ae; s.expert_base = w2.expert_base; s.local_experts = w2.local_experts; s.n_slots = K;
    seg_act(s, w2.quant, m->a_hb, m->hb);
    s.a_slot_stride = hb_stride;
    s.out = m->eout; s.out_slot_stride = c.dim; s.epilogue = EPI_STORE;
    PROFILED("gemv_experts_w2", (double)K * e_bytes + K * act_bytes(w2.quant, mi, c.dim), launch_gemv(st, w2.quant, s));
  }
  if (c.n_shared_experts > 0) {
    const DTensor& t = L.t[DSK_ROLE_SHARED_W2];
    GemvSeg s;
    seg_from(m, s, t);
    if (is_kq(t.quant)) {
      const size_t off = (size_t)K * hb_stride;
      s.a_qs = m->a_hb.qs + off; s.a_d = m->a_hb.d + off / 256; s.a_bsums = m->a_hb.bsums + off / 16;
    } else {
      s.a_f32 = m->hb + (size_t)K * hb_stride;
    }
    s.out = m->eout + (size_t)K * c.dim; s.epilogue = EPI_STORE;
    PROFILED("gemv_shared_w2", tensor_bytes_2d(t) + act_bytes(t.quant, t.n, t.rows), launch_gemv(st, t.quant, s));
  }
  if (m->ctx->world > 1) {
    // every slot is non-zero on exactly one rank: a sum all-reduce is exact and order-independent
    ncclResult_t r = ncclAllReduce(m->eout, m->eout, (size_t)K * c.dim, ncclFloat, ncclSum, m->ctx->comm, st);
    if (r != ncclSuccess) DSK_FAIL(DSK_ERR_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
  }
  *pending_combine = true;
  return DSK_OK;
}

// enqueue one whole token on the stream (no host synchronisation inside)
static int enqueue_forward(dsk_model* m, int mode, int max_kv) {
  const dsk_config& c = m->c;
  hipStream_t st = m->ctx->stream;
  const bool kq = is_kq(c.weight_quant);
  HIP_TRY(hipMemcpyAsync(m->sp_dev, m->sp_host, sizeof(StepParams), hipMemcpyHostToDevice, st));
  PROFILED("embed", (double)mat_bytes(c.weight_quant, 1, c.dim), launch_embed(st, m->g[DSK_ROLE_EMBED], m->sp_dev, -1, std::max(1, c.block_size[0]),
                                                                              std::max(1, c.block_size[1]), m->x));
  bool pending = false;  // previous layer's expert outputs still to be folded into x
  const int K = c.n_active_routed;
  auto norm_job = [&](const float* weight, bool want_f32, bool want_q8) {
    NormJob j;
    memset(&j, 0, sizeof j);
    j.x = m->x; j.weight = weight; j.n = c.dim; j.eps = c.norm_eps;
    if (want_f32) j.y_f32 = m->xb;
    if (want_q8 && kq) { j.q_qs = m->a_xb.qs; j.q_d = m->a_xb.d; j.q_bsums = m->a_xb.bsums; }
    return j;
  };
  for (int l = 0; l < c.n_layers; ++l) {
    Layer& L = m->L[l];
    {  // attention pre-norm (src/infer.cpp:823), folding in the previous block's MoE combine
      NormJob j = norm_job(reinterpret_cast<const float*>(L.t[DSK_ROLE_ATTN_NORM].qs), !kq, true);
      if (pending) {
        j.eout = m->eout; j.eweights = m->route_w + (size_t)(l - 1) * K; j.n_routed_slots = K;
        j.add_shared = c.n_shared_experts > 0; j.x_store = m->x;
      }
      PROFILED("norm_q8", (double)c.dim * (pending ? (K + 2) * 4 + 9 : 9), launch_norm_jobs(st, &j, 1, m->sp_dev));
      if (m->trace && pending) HIP_TRY(hipMemcpyAsync(m->trace_x + (size_t)(l - 1) * c.dim, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToDevice, st));
      pending = false;
    }
    if (c.use_mla) DSK_TRY(attention_mla(m, L, max_kv));
    else DSK_TRY(attention_mha(m, L, max_kv));
    {  // FFN pre-norm (src/infer.cpp:839); the router needs xb in f32
      NormJob j = norm_job(reinterpret_cast<const float*>(L.t[DSK_ROLE_FFN_NORM].qs), !kq || L.is_moe, true);
      PROFILED("norm_q8", (double)c.dim * 13, launch_norm_jobs(st, &j, 1, m->sp_dev));
    }
    DSK_TRY(ffn(m, l, &pending));
    if (m->trace && !pending) HIP_TRY(hipMemcpyAsync(m->trace_x + (size_t)l * c.dim, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToDevice, st));
  }
  const int last = c.n_layers - 1;
  if (mode == DSK_MODE_HYDRATE_KV_CACHE) {
    // skip final norm + lm_head (src/infer.cpp:1284-1287); x is dead afterwards, but keep it exact for the taps
    if (pending) {
      NormJob j = norm_job(nullptr, false, false);
      j.eout = m->eout; j.eweights = m->route_w + (size_t)last * K; j.n_routed_slots = K; j.add_shared = c.n_shared_experts > 0; j.x_store = m->x;
      PROFILED("norm_q8", (double)c.dim * (K + 2) * 4, launch_norm_jobs(st, &j, 1, m->sp_dev));
      if (m->trace) HIP_TRY(hipMemcpyAsync(m->trace_x + (size_t)last * c.dim, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToDevice, st));
    }
    return DSK_OK;
  }
  {  // final norm (src/infer.cpp:1292) + classifier (src/infer.cpp:1297-1316)
    NormJob j = norm_job(reinterpret_cast<const float*>(m->g[DSK_ROLE_FINAL_NORM].qs), !kq, true);
    if (pending) {
      j.eout = m->eout; j.eweights = m->route_w + (size_t)last * K; j.n_routed_slots = K; j.add_shared = c.n_shared_experts > 0; j.x_store = m->x;
    }
    PROFILED("norm_q8", (double)c.dim * 13, launch_norm_jobs(st, &j, 1, m->sp_dev));
    if (m->trace && pending) HIP_TRY(hipMemcpyAsync(m->trace_x + (size_t)last * c.dim, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToDevice, st));
  }
  const DTensor& cls = m->tied ? m->g[DSK_ROLE_EMBED] : m->g[DSK_ROLE_OUTPUT];
  DSK_TRY(run_gemv(m, "gemv_lm_head", cls, m->a_xb, m->xb, m->logits, EPI_STORE));
  HIP_TRY(hipMemcpyAsync(m->logits_host, m->logits, (size_t)c.vocab_size * 4, hipMemcpyDeviceToHost, st));
  return DSK_OK;
}

static int check_forward_args(dsk_model* m, int token, int pos, int mode, float* host_logits) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "forward: null model");
  if (!m->finalized) DSK_FAIL(DSK_ERR_STATE, "forward before finalize");
  if (token < 0 || token >= m->c.vocab_size) DSK_FAIL(DSK_ERR_INVALID, "forward: token %d out of range", token);
  if (pos < 0) DSK_FAIL(DSK_ERR_INVALID, "forward: negative pos");
  if (mode != DSK_MODE_HYDRATE_KV_CACHE && mode != DSK_MODE_OUTPUT_LOGITS) DSK_FAIL(DSK_ERR_INVALID, "forward: bad mode %d", mode);
  if (mode == DSK_MODE_OUTPUT_LOGITS && !host_logits) DSK_FAIL(DSK_ERR_INVALID, "forward: OUTPUT_LOGITS needs a logits buffer");
  return DSK_OK;
}

extern "C" int dsk_forward(dsk_model* m, int token, int pos, int mode, float* host_logits) {
  DSK_TRY(check_forward_args(m, token, pos, mode, host_logits));
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  DSK_TRY(fill_step_params(m, token, pos));
  const int max_kv = m->c.max_seq_len;  // LDS for attention scores is sized for the allocation: graph-replay safe
  const bool graphable = m->use_graph && !m->trace && !m->profiling;
  if (graphable) {
    const int gi = mode == DSK_MODE_OUTPUT_LOGITS ? 1 : 0;
    if (!m->graph[gi]) {
      hipGraph_t g = nullptr;
      HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      int r = enqueue_forward(m, mode, max_kv);
      hipError_t e = hipStreamEndCapture(st, &g);
      if (r != DSK_OK) {
        if (g) hipGraphDestroy(g);
        return r;
      }
      if (e != hipSuccess) DSK_FAIL(DSK_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
      HIP_TRY(hipGraphInstantiate(&m->graph[gi], g, nullptr, nullptr, 0));
      HIP_TRY(hipGraphDestroy(g));
    }
    HIP_TRY(hipGraphLaunch(m->graph[gi], st));
  } else {
    DSK_TRY(enqueue_forward(m, mode, max_kv));
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  if (mode == DSK_MODE_OUTPUT_LOGITS) memcpy(host_logits, m->logits_host, (size_t)m->c.vocab_size * 4);
  return DSK_OK;
}

extern "C" int dsk_model_set_graph(dsk_model* m, int enable) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "null model");
  m->use_graph = enable != 0;
  return DSK_OK;
}
extern "C" int dsk_model_set_trace(dsk_model* m, int enable) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "null model");
  m->trace = enable != 0;
  return DSK_OK;
}
extern "C" int dsk_model_get_trace_x(dsk_model* m, int layer, float* x_out) {
  if (!m || !m->finalized || layer < 0 || layer >= m->c.n_layers || !x_out) DSK_FAIL(DSK_ERR_INVALID, "get_trace_x: bad argument");
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpy(x_out, m->trace_x + (size_t)layer * m->c.dim, (size_t)m->c.dim * 4, hipMemcpyDeviceToHost));
  return DSK_OK;
}
extern "C" int dsk_model_get_routing(dsk_model* m, int32_t* experts, float* weights) {
  if (!m || !m->finalized || !experts || !weights) DSK_FAIL(DSK_ERR_INVALID, "get_routing: bad argument");
  HIP_TRY(hipSetDevice(m->ctx->device));
  const int K = std::max(1, m->c.n_active_routed);
  const size_t n = (size_t)m->c.n_layers * K;
  HIP_TRY(hipMemcpy(experts, m->route_e, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(weights, m->route_w, n * 4, hipMemcpyDeviceToHost));
  for (int l = 0; l < m->c.n_layers; ++l)
    if (!m->L[l].is_moe)
      for (int k = 0; k < K; ++k) { experts[(size_t)l * K + k] = -1; weights[(size_t)l * K + k] = 0.f; }
  return DSK_OK;
}

extern "C" int dsk_profile_forward(dsk_model* m, int token, int pos, dsk_kernel_time* out, int max_classes, int* n_classes) {
  DSK_TRY(check_forward_args(m, token, pos, DSK_MODE_OUTPUT_LOGITS, m ? m->logits_host : nullptr));
  if (!out || !n_classes) DSK_FAIL(DSK_ERR_INVALID, "profile_forward: null output");
  HIP_TRY(hipSetDevice(m->ctx->device));
  DSK_TRY(fill_step_params(m, token, pos));
  for (auto& k : m->ktimes) { k.launches = 0; k.algo_bytes = 0; }
  m->profiling = true;
  int r = enqueue_forward(m, DSK_MODE_OUTPUT_LOGITS, m->c.max_seq_len);
  m->profiling = false;
  if (r != DSK_OK) return r;
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  int n = 0;
  for (auto& k : m->ktimes) {
    float total = 0.f;
    for (auto& e : k.ev) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
      total += ms;
      hipEventDestroy(e.first);
      hipEventDestroy(e.second);
    }
    k.ev.clear();
    if (k.launches == 0) continue;
    if (n < max_classes) {
      out[n].name = k.name;
      out[n].launches = k.launches;
      out[n].total_ms = total;
      out[n].algo_bytes = k.algo_bytes;
      ++n;
    }
  }
  *n_classes = n;
  return DSK_OK;
}

// algorithmic bytes of one forward (corrected analogue of Model::active_bytes, src/model.cpp:885-901)
extern "C" double dsk_model_active_bytes(const dsk_model* m, int pos) {
  if (!m) return 0;
  const dsk_config& c = m->c;
  double b = (double)mat_bytes(c.weight_quant, 1, c.dim);  // one embedding row
  const int kv_len = std::min(c.rs_original_max_position_embeddings, pos + 1);
  for (int l = 0; l < c.n_layers; ++l) {
    const Layer& L = m->L[l];
    for (int role : ALL_LAYER_ROLES) {
      const RoleShape rs = role_shape(m, role, l);
      if (!rs.ok) continue;
      double tb = (double)mat_bytes(rs.quant, rs.rows, rs.n);
      if (rs.quant == DSK_QUANT_F8E5M2) tb += 4.0 * cdiv(rs.rows, c.block_size[0]) * cdiv(rs.n, c.block_size[1]);
      if (rs.e > 0) tb *= c.n_active_routed;  // only the selected experts are touched
      b += tb;
    }
    (void)L;
    if (c.use_mla) b += (double)kv_len * (c.kv_lora_rank + c.qk_rope_head_dim) * 2;
    else b += (double)kv_len * c.n_heads * (m->head_dim + c.v_head_dim) * 2;
  }
  b += (double)c.dim * 4;  // final norm
  double cls = (double)mat_bytes(c.weight_quant, c.vocab_size, c.dim);
  if (c.weight_quant == DSK_QUANT_F8E5M2) cls += 4.0 * cdiv(c.vocab_size, c.block_size[0]) * cdiv(c.dim, c.block_size[1]);
  return b + cls;
}
extern "C" double dsk_model_device_bytes(const dsk_model* m) { return m ? m->weight_bytes + m->cache_bytes + m->scratch_bytes : 0; }
