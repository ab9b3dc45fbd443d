// engine.cpp -- host side of the gfx950 decode engine behind the C ABI (include/dsk.h).
//
// Mirrors what the reference does around its hot path, for one GPU per process:
//   Model::Model / Block ctors (src/model.cpp:149-285, 354-461, 518-620, 756-872) -> dsk_model_bind
//   InferenceState scratch (src/model.cpp:677-726)                                -> dsk_model_finalize
//   Model::_forward_cpu / Block::_block_cpu (src/infer.cpp:1265-1317, 810-932)    -> dsk_forward
// All kernels of a token are enqueued on one HIP stream without host synchronisation; routing
// decisions stay in HBM; the whole step is captured once into a hipGraph and replayed.
#include "engine.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <thread>
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
void dsk_clear_error() { g_err[0] = 0; }
extern "C" const char* dsk_last_error(void) { return g_err; }
extern "C" int dsk_abi_version(void) { return DSK_ABI_VERSION; }

// struct dsk_ctx / dsk_model: engine.h

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
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) c->n_cus = cus;
  *out = c;
  return DSK_OK;
}
// A context outlives its models: destroying it while models are alive only marks it; the last dsk_model_destroy
// frees it (a host binding may drop its handles in any order, e.g. at interpreter exit).
static int ctx_free(dsk_ctx* c);
extern "C" int dsk_ctx_destroy(dsk_ctx* c) {
  if (!c) return DSK_OK;
  if (c->live_models > 0) {
    c->closing = true;
    return DSK_OK;
  }
  return ctx_free(c);
}
void ctx_model_released(dsk_ctx* c) {
  if (--c->live_models == 0 && c->closing) ctx_free(c);
}
static int ctx_free(dsk_ctx* c) {
  hipSetDevice(c->device);
  for (int i = 0; i < 8; ++i)
    if (c->op_buf[i]) hipFree(c->op_buf[i]);
  for (int i = 0; i < dsk_ctx::STAGE_BUFS; ++i) {
    if (c->pin[i]) hipHostFree(c->pin[i]);
    if (c->pin_ev[i]) hipEventDestroy(c->pin_ev[i]);
  }
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
// ---------------------------------------------------------------------------------
// host -> HBM staging ring.  `bytes` of `src` (from src_off on) go to dev_dst, asynchronously on the context stream;
// the SOURCE is consumed when this returns (it has been copied into pinned memory), the device copy completes in
// stream order.  Pieces of <= STAGE_BYTES; each piece is filled by up to 8 threads (memcpy, or pread on the file).
// ---------------------------------------------------------------------------------
static int fill_piece(const HostSrc& src, uint64_t off, char* dst, size_t n) {
  const size_t min_slice = (size_t)4 << 20;
  int nt = (int)std::min<size_t>(8, (n + min_slice - 1) / min_slice);
  if (nt < 1) nt = 1;
  std::vector<int> rc(nt, 0);
  auto work = [&](int i) {
    const size_t a = n * i / nt, b = n * (i + 1) / nt;
    if (src.fd < 0) {
      memcpy(dst + a, (const char*)src.ptr + off + a, b - a);
      return;
    }
    size_t done = a;
    while (done < b) {
      const ssize_t r = pread(src.fd, dst + done, b - done, (off_t)(src.off + off + done));
      if (r <= 0) { rc[i] = -1; return; }
      done += (size_t)r;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(work, i);
    work(0);
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < nt; ++i)
    if (rc[i]) DSK_FAIL(DSK_ERR_INVALID, "loader: short read at offset %llu", (unsigned long long)(src.off + off));
  return DSK_OK;
}
int stage_copy(dsk_ctx* ctx, const HostSrc& src, uint64_t src_off, void* dev_dst, size_t bytes) {
  for (size_t done = 0; done < bytes;) {
    const size_t n = std::min(bytes - done, dsk_ctx::STAGE_BYTES);
    const int k = ctx->pin_next;
    ctx->pin_next = (k + 1) % dsk_ctx::STAGE_BUFS;
    if (!ctx->pin[k]) {
      HIP_TRY(hipHostMalloc(&ctx->pin[k], dsk_ctx::STAGE_BYTES, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming));
    }
    if (ctx->pin_busy[k]) HIP_TRY(hipEventSynchronize(ctx->pin_ev[k]));
    const auto t0 = std::chrono::steady_clock::now();
    DSK_TRY(fill_piece(src, src_off + done, (char*)ctx->pin[k], n));
    ctx->staged_fill_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipMemcpyAsync((char*)dev_dst + done, ctx->pin[k], n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->pin_ev[k], ctx->stream));
    ctx->pin_busy[k] = true;
    ctx->staged_bytes += (double)n;
    done += n;
  }
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
  // models already created on this context hold shards sized for the old world and (graph_with_comm) captured graphs whose
  // nodes reference the old communicator: replacing it under them would be a use-after-free inside RCCL
  if (c->live_models > 0 && (c->comm || c->world != world || c->rank != rank))
    DSK_FAIL(DSK_ERR_STATE, "comm_init: %d live model(s) on this context; destroy them before changing the communicator", c->live_models);
  c->rank = rank;
  c->world = world;
  if (!uid128) return DSK_OK;  // world > 1: dry run of one shard (no communicator, the all-reduce is skipped)
  // (world == 1 WITH a uid: a one-rank communicator - the exchange of a model created with the option
  //  "force_exchange" then really calls RCCL on the engine stream: tests/test_comm_gpu.py)
  HIP_TRY(hipSetDevice(c->device));
  if (c->comm) {  // a second call replaces the communicator
    ncclCommDestroy(c->comm);
    c->comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, uid128, 128);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) DSK_FAIL(DSK_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
  return DSK_OK;
}

// ---------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------

bool is_kq(int q) { return q == DSK_QUANT_Q2_K || q == DSK_QUANT_Q3_K; }
static int cdiv(int a, int b) { return (a + b - 1) / b; }
int cdiv_i(int a, int b) { return cdiv(a, b); }
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
  if (ctx->closing) DSK_FAIL(DSK_ERR_STATE, "model_create: the context has been destroyed");
  dsk_model* m = new dsk_model();
  m->ctx = ctx;
  ctx->live_models++;
  m->c = c;
  m->head_dim = c.qk_nope_head_dim + c.qk_rope_head_dim;
  m->L.resize(c.n_layers);
  for (int l = 0; l < c.n_layers; ++l) m->L[l].is_moe = c.n_routed_experts > 0 && l >= c.first_k_dense_replace;
  m->mha_split_min = 0;  // 0: MHA_SPLIT_MIN_KV (finalize)
#ifdef DSK_AB
  // A/B builds only (tools/ab_build.sh): the old environment knobs seed the options, so that one command line can flip a
  // kernel choice in bench.py / tools/*.py without touching their code.  The shipped library never reads the environment.
  auto env_i = [](const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; };
  if (getenv("DSK_NO_FUSE_MOE")) m->fuse_moe = false;
  if (getenv("DSK_NO_FUSE_SHARED")) m->ride_shared = false;
  if (getenv("DSK_NO_KVWRITE_RIDE")) m->ride_kvwrite = false;
  if (getenv("DSK_NO_COMPACT")) m->compact_absent = false;
  m->att_q8_in_wo = env_i("DSK_ATT_Q8_IN_WO", 0) != 0;
  m->rider_fill = env_i("DSK_RIDER_FILL", m->rider_fill);
  m->mla_flash_min_kv = std::max(32, env_i("DSK_MLA_FLASH_MIN", m->mla_flash_min_kv));
  m->mha_split_min = env_i("DSK_MHA_SPLIT_MIN", 0);
  if (getenv("DSK_MOE_TIMELINE") || getenv("DSK_TIMELINE")) m->want_timeline = true;
  if (getenv("DSK_NO_MOE_Q8_HANDOFF")) m->moe_q8_handoff = false;
  if (getenv("DSK_TAIL_PF")) m->tail_prefetch = atoi(getenv("DSK_TAIL_PF"));
  if (getenv("DSK_NO_FUSE_MOE_FLOAT")) m->fuse_moe_float = false;
  m->hydrate_route_seed = std::max(0, env_i("DSK_HYD_ROUTE_SEED", 0));  // measurement only: uniform routing in the batched prompt path
#endif
  *out = m;
  return DSK_OK;
}

// Per-model options (include/dsk.h).  Everything here selects kernels, launch shapes or diagnostics; all of it must be
// set between dsk_model_create and dsk_model_finalize (the launch plans are built there).
extern "C" int dsk_model_set_option(dsk_model* m, const char* key, int value) {
  if (!m || !key) DSK_FAIL(DSK_ERR_INVALID, "set_option: null argument");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "set_option('%s') after finalize", key);
  const std::string k(key);
  if (k == "fuse_moe") m->fuse_moe = value != 0;
  else if (k == "fuse_shared") m->ride_shared = value != 0;
  else if (k == "ride_kvwrite") m->ride_kvwrite = value != 0;
  else if (k == "att_q8_in_wo") m->att_q8_in_wo = value != 0;
  else if (k == "compact_absent") m->compact_absent = value != 0;
  else if (k == "rider_fill") { if (value < 1 || value > 16) DSK_FAIL(DSK_ERR_INVALID, "set_option: rider_fill %d", value); m->rider_fill = value; }
  else if (k == "mla_flash_min") m->mla_flash_min_kv = std::max(32, value);
  else if (k == "mha_split_min") m->mha_split_min = std::max(0, value);
  else if (k == "timeline") m->want_timeline = value != 0;
  else if (k == "moe_spin_limit") m->moe_spin_limit = value;
  else if (k == "moe_q8_handoff") m->moe_q8_handoff = value != 0;
  else if (k == "moe_pipe") m->moe_pipe = value;
  else if (k == "gemv_ahead") m->gemv_ahead = value & 3;
  else if (k == "tail_prefetch") m->tail_prefetch = value;
  else if (k == "fuse_moe_float") m->fuse_moe_float = value != 0;
  else if (k == "force_exchange") m->force_exchange = value != 0;
  else if (k == "graph_with_comm") m->graph_with_comm = value != 0;
  else if (k == "exchange_allgather") m->exchange_allgather = value != 0;
  else if (k == "hydrate_chunk") { if (value < 1 || value > 1024) DSK_FAIL(DSK_ERR_INVALID, "set_option: hydrate_chunk %d (1 .. 1024)", value); m->hydrate_chunk = value; }
  else if (k == "hydrate_batched") m->hydrate_batched = value != 0;
  else if (k == "hydrate_tile_copies") m->hydrate_tile_copies = value != 0;
  else if (k == "hydrate_tap_layer") { if (value < -1 || value >= m->c.n_layers) DSK_FAIL(DSK_ERR_INVALID, "set_option: hydrate_tap_layer %d", value); m->hydrate_tap_layer = value; }
  else if (k == "q2k_tiles") {
    if (m->any_bound) DSK_FAIL(DSK_ERR_STATE, "set_option: q2k_tiles must be set before the first tensor is bound");
    if (value < 0 || value > 2) DSK_FAIL(DSK_ERR_INVALID, "set_option: q2k_tiles %d (0 none, 1 experts, 2 every converted role)", value);
    m->q2k_tiles = value;
  }
  else DSK_FAIL(DSK_ERR_INVALID, "set_option: unknown option '%s'", key);
  return DSK_OK;
}

extern "C" int dsk_model_get_info(dsk_model* m, const char* key, int* value) {
  if (!m || !key || !value) DSK_FAIL(DSK_ERR_INVALID, "get_info: null argument");
  const std::string k(key);
  if (k == "handoff_fallbacks") *value = m->handoff_fallbacks;
  else if (k == "graph_capture_fallbacks") *value = m->graph_capture_fallbacks;
  else if (k == "hydrate_batched_tokens") *value = (int)std::min<long long>(m->hydrate_batched_tokens, 0x7fffffff);
  else if (k == "hydrate_looped_tokens") *value = (int)std::min<long long>(m->hydrate_looped_tokens, 0x7fffffff);
  else if (k == "gemv_ahead_plans") {  // launch plans that run a "weights ahead of the staging" kernel (option "gemv_ahead")
    int n = 0;
    for (size_t l = 0; l < m->lp_qkv_a.size(); ++l) {
      if (m->lp_qkv_a[l] >= 0) n += gemv_ahead_kind(m->plans[m->lp_qkv_a[l]], false) != 0;
      if (m->lp_wo[l] >= 0) n += gemv_ahead_kind(m->plans[m->lp_wo[l]], false) != 0;
      if (m->c.use_mla && m->ride_kvwrite && m->lp_qkv_b[l] >= 0) n += gemv_ahead_kind(m->plans[m->lp_qkv_b[l]], true) != 0;
    }
    *value = n;
  }
  else if (k == "hydrate_tile_copy_mb") *value = (int)(m->hydrate_tile_copy_bytes / 1048576.0);
  else if (k == "fused_moe_layers") { int n = 0; for (auto& a : m->moe_ffn) n += a.grid > 0; *value = n; }
  else if (k == "graph_captured") { int n = 0; for (auto g : m->graph) n += g != nullptr; *value = n; }
  else if (k == "exchange_calls") *value = m->exchange_calls;
  else if (k == "tiled_tensors") {  // weight tensors stored as tile records (option "q2k_tiles")
    int n = 0;
    for (auto& t : m->g) n += t.tiled;
    for (auto& L : m->L) for (auto& t : L.t) n += t.tiled;
    *value = n;
  }
  else DSK_FAIL(DSK_ERR_INVALID, "get_info: unknown key '%s'", key);
  return DSK_OK;
}

// expected logical shape of a role (src/model.cpp:184-285, 393-457, 557-616, 766-871)
RoleShape role_shape(const dsk_model* m, int role, int layer) {
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

const int N_LAYER_ROLES = 21;
const int ALL_LAYER_ROLES[] = {DSK_ROLE_ATTN_NORM, DSK_ROLE_Q_A_NORM, DSK_ROLE_KV_A_NORM, DSK_ROLE_FFN_NORM, DSK_ROLE_WQ,
                                      DSK_ROLE_WQ_A, DSK_ROLE_WQ_B, DSK_ROLE_WKV_A, DSK_ROLE_WKV_B, DSK_ROLE_WO, DSK_ROLE_WC,
                                      DSK_ROLE_WQ_ROPE_B, DSK_ROLE_WV_B, DSK_ROLE_W1, DSK_ROLE_W2, DSK_ROLE_W3,
                                      DSK_ROLE_SHARED_W1, DSK_ROLE_SHARED_W2, DSK_ROLE_SHARED_W3, DSK_ROLE_MOEGATE,
                                      DSK_ROLE_MOEGATE_BIAS};

static bool is_routed_role(int role) { return role == DSK_ROLE_W1 || role == DSK_ROLE_W2 || role == DSK_ROLE_W3; }

// Which Q2_K tensors live in the tiled layout (tile_device.h), by option "q2k_tiles" (set before the first bind):
//   0  none (the plane layout and the dot4 row products everywhere)
//   1  (default) the routed experts' and the shared expert's matrices, when the fused expert launch can run them
//      (kernels_moe_tile.hip: hidden vectors of <= 2048 values): same-box A/B on the full model, phase A of the fused launch
//      16.6 -> 14.0 us, the launch 34.3 -> 34.1 us; the other converted roles are at parity or behind in the model (wo 10.9 -> 11.9 us:
//      448 strips of 16 rows deal 2 : 1 over 256 CUs where 7168 rows deal evenly), so they stay on planes
//   2  every converted role (experts, shared expert, dense FFN, first-stage projections, wq_b / wkv_b or the MLA second stage and wv_b, wo,
//      embedding / classifier): tests, kbench, and what dsk_hydrate's batched path needs
#ifndef TILE_LEVEL_HEAD
#define TILE_LEVEL_HEAD 2  // the q2k_tiles level from which wq_b / wkv_b are tiled
#endif
static bool tile_experts_ok(const dsk_config& c) {
  const int mi = c.moe_intermediate_size, sn = c.n_shared_experts * mi;
  return c.n_routed_experts > 0 && c.dim % 256 == 0 && mi % 256 == 0 && mi / 256 <= 8 && sn % 256 == 0 && sn / 256 <= 8;
}
static bool role_tiled(const dsk_model* m, int role, int e, int quant) {
  if (quant != DSK_QUANT_Q2_K || m->q2k_tiles <= 0) return false;
  const bool all = m->q2k_tiles >= 2;
  switch (role) {
    case DSK_ROLE_EMBED: case DSK_ROLE_OUTPUT: case DSK_ROLE_WQ: case DSK_ROLE_WQ_A: case DSK_ROLE_WKV_A: case DSK_ROLE_WO: return all;
    case DSK_ROLE_W1: case DSK_ROLE_W2: case DSK_ROLE_W3: return e == 0 ? all : (all || tile_experts_ok(m->c));
    case DSK_ROLE_SHARED_W1: case DSK_ROLE_SHARED_W2: case DSK_ROLE_SHARED_W3: return all || tile_experts_ok(m->c);
    case DSK_ROLE_WQ_B: case DSK_ROLE_WKV_B: {  // the per-head attention launch's projections (kernels_gemv.hip head_attn_kernel)
      const dsk_config& c = m->c;
      const int hd = c.qk_nope_head_dim + c.qk_rope_head_dim, nv = c.qk_nope_head_dim + c.v_head_dim;
      const bool ok = !c.use_mla && c.q_lora_rank > 0 && c.q_lora_rank % 256 == 0 && c.kv_lora_rank % 256 == 0 && c.q_lora_rank / 256 <= 8 &&
                      c.kv_lora_rank / 256 <= 8 && hd % 16 == 0 && nv % 16 == 0;
      return ok && m->q2k_tiles >= TILE_LEVEL_HEAD;
    }
    case DSK_ROLE_WQ_ROPE_B: case DSK_ROLE_WC: {  // MLA second stage (gemv_kvwrite_tile_kernel): rows of q_lora_rank values
      const dsk_config& c = m->c;
      return all && c.use_mla && c.q_lora_rank > 0 && c.q_lora_rank % 256 == 0;
    }
    case DSK_ROLE_WV_B: {  // MLA per-head value projection inside mla_head_kernel: whole 16-row strips per head, rows of <= 8 blocks
      const dsk_config& c = m->c;
      return all && c.use_mla && c.v_head_dim % 16 == 0 && c.kv_lora_rank % 256 == 0 && c.kv_lora_rank / 256 <= 8 &&
             (c.v_head_dim / 16) * (c.kv_lora_rank / 256) <= 64;
    }
    default: return false;
  }
}

// allocate the device planes of a tensor (t.tiled set by the caller: Q2_K tile records instead); local = experts kept on this rank
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
  if (quant != DSK_QUANT_Q2_K) t.tiled = false;
  if (t.tiled) {  // tile records (tile_device.h): rows padded to 16 per matrix; padding rows stay zero (memset below)
    t.e_qs = tile_mat_bytes(rows, n);
    total = mats * t.e_qs;
  } else if (quant == DSK_QUANT_Q2_K) {
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
  if (t.tiled) {
    if (rows & 15) HIP_TRY(hipMemset(t.base, 0, total));
  } else if (quant == DSK_QUANT_Q2_K || quant == DSK_QUANT_Q3_K) {
    t.sc = t.base + o_sc;
    t.dm = t.base + o_dm;
    if (quant == DSK_QUANT_Q3_K) t.hm = t.base + o_hm;
  }
  if (quant == DSK_QUANT_F8E5M2) t.scale = reinterpret_cast<float*>(t.base + o_scale);
  if (e == 0) t.e_qs = t.e_sc = t.e_hm = t.e_dm = 0;  // strides only meaningful for stacks (e_scale kept: scale count)
  return DSK_OK;
}

// copy the local part of a tensor from the reference's host layout (memory or file range) into the device planes;
// asynchronous on the context stream (the caller synchronises), the source is consumed on return
int upload_tensor(dsk_ctx* ctx, DTensor& t, const HostSrc& src) {
  hipStream_t st = ctx->stream;
  const size_t per_bytes = mat_bytes(t.quant, t.rows, t.n);
  const size_t lm = t.n_experts > 0 ? (size_t)t.local_experts : 1;
  const uint64_t off0 = (uint64_t)t.expert_base * per_bytes;
  if (lm == 0) return DSK_OK;
  if (!is_kq(t.quant)) return stage_copy(ctx, src, off0, t.qs, lm * per_bytes);
  // K-quants: stage AoS blocks in chunks, re-lay-out into planes on the GPU (stream order: the repack of chunk i is
  // done before the copy of chunk i+1 overwrites the device staging buffer)
  const size_t bsz = t.quant == DSK_QUANT_Q2_K ? 84 : 110;
  const size_t total_blocks = lm * per_bytes / bsz;
  const size_t chunk_blocks = std::min<size_t>(total_blocks, dsk_ctx::STAGE_BYTES / bsz / 64 * 64);
  void* stage;
  DSK_TRY(ctx_scratch(ctx, 7, chunk_blocks * bsz + 256, &stage));
  for (size_t b0 = 0; b0 < total_blocks; b0 += chunk_blocks) {
    const size_t nb = std::min(chunk_blocks, total_blocks - b0);
    DSK_TRY(stage_copy(ctx, src, off0 + b0 * bsz, stage, nb * bsz));
    if (t.tiled)
      DSK_TRY(launch_repack_q2k_tiles(st, (const uint8_t*)stage, b0, nb, t.rows, t.n / 256, tile_mat_bytes(t.rows, t.n), t.qs));
    else if (t.quant == DSK_QUANT_Q2_K)
      DSK_TRY(launch_repack_q2k(st, (const uint8_t*)stage, nb, t.qs + b0 * 64, t.sc + b0 * 16, t.dm + b0 * 4));
    else
      DSK_TRY(launch_repack_q3k(st, (const uint8_t*)stage, nb, t.qs + b0 * 64, t.hm + b0 * 32, t.sc + b0 * 12, t.dm + b0 * 2));
  }
  HIP_TRY(hipGetLastError());
  return DSK_OK;
}

int upload_tensor(dsk_ctx* ctx, DTensor& t, const void* host_ptr) {
  HostSrc src;
  src.ptr = host_ptr;
  return upload_tensor(ctx, t, src);
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

// Expert e of a routed stack lives on rank e / ceil(E / world): contiguous blocks, so that with the
// reference's group-limited routing (<= topk_group survivors per group of E / n_group experts) the
// per-rank load is bounded.  Pure host arithmetic (callable without a GPU; tests/test_dist_cpu.py).
extern "C" int dsk_expert_shard(int n_experts, int world, int rank, int* base, int* count) {
  if (n_experts < 0 || world < 1 || rank < 0 || rank >= world || !base || !count) DSK_FAIL(DSK_ERR_INVALID, "expert_shard: bad argument");
  const int per = cdiv(n_experts, world);
  *base = std::min(n_experts, rank * per);
  *count = std::max(0, std::min(per, n_experts - *base));
  return DSK_OK;
}

// the rank that owns expert `expert` under dsk_expert_shard's partition - what the gathered combine evaluates on the device
// (kernels_misc.hip moe_combine_gathered_kernel: e / ceil(E / world))
extern "C" int dsk_expert_owner(int n_experts, int world, int expert, int* rank) {
  if (n_experts < 1 || world < 1 || expert < 0 || expert >= n_experts || !rank) DSK_FAIL(DSK_ERR_INVALID, "expert_owner: bad argument");
  *rank = expert / cdiv(n_experts, world);
  return DSK_OK;
}

// SURVEY 8 row f-4 (tensor parallelism, DESIGN.md 4.4: design + host arithmetic, not yet an engine mode): output rows of a
// replicated GEMV are dealt to the ranks in contiguous ranges that are multiples of `unit` rows (unit = 256 where the
// output is quantised to Q8_K blocks downstream, the head size for per-head projections, 1 otherwise); rows are independent
// dot products, so an all-gather of the ranks' ranges reproduces the one-GPU vector bit for bit.  The first
// (units % world) ranks take one unit more.  Pure host arithmetic (tests/test_dist_cpu.py).
extern "C" int dsk_tp_rows(int rows, int unit, int world, int rank, int* row0, int* count) {
  if (rows < 0 || unit < 1 || world < 1 || rank < 0 || rank >= world || !row0 || !count || rows % unit) DSK_FAIL(DSK_ERR_INVALID, "tp_rows: bad argument");
  const int units = rows / unit, q = units / world, r = units % world;
  const int u0 = rank * q + std::min(rank, r), n = q + (rank < r ? 1 : 0);
  *row0 = u0 * unit;
  *count = n * unit;
  return DSK_OK;
}
// attention heads of rank `rank`: [*head0, *head0 + *count); with them go the head's rows of wq_b / wkv_b (or wc, wq_rope_b,
// wv_b), its KV-cache columns (each rank keeps n_heads / world of the cache) and its 128 columns of wo's INPUT - which is
// why wo is split by output rows instead and preceded by an all-gather of the attention output
extern "C" int dsk_tp_heads(int n_heads, int world, int rank, int* head0, int* count) {
  return dsk_tp_rows(n_heads, 1, world, rank, head0, count);
}

static void shard_range(const dsk_model* m, int role, int e, int* base, int* local) {
  *base = 0;
  *local = e;
  if (e > 0 && is_routed_role(role) && m->ctx->world > 1) dsk_expert_shard(e, m->ctx->world, m->ctx->rank, base, local);
}

extern "C" int dsk_model_bind(dsk_model* m, int role, int layer, int quant, const int32_t shape[4], const void* host_ptr, size_t bytes) {
  if (!m || !host_ptr || !shape) DSK_FAIL(DSK_ERR_INVALID, "bind: null argument");
  HostSrc src;
  src.ptr = host_ptr;
  DSK_TRY(bind_src(m, role, layer, quant, shape, src, bytes));
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  return DSK_OK;
}

// the body of dsk_model_bind over a HostSrc; asynchronous (loader.cpp synchronises once per checkpoint)
int bind_src(dsk_model* m, int role, int layer, int quant, const int32_t shape[4], const HostSrc& src, size_t bytes) {
  if (!m || !shape) DSK_FAIL(DSK_ERR_INVALID, "bind: null argument");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "bind after finalize");
  HIP_TRY(hipSetDevice(m->ctx->device));
  const bool is_scale = role >= DSK_ROLE_SCALE;
  const int r = is_scale ? role - DSK_ROLE_SCALE : role;
  DTensor* t;
  DSK_TRY(tensor_slot(m, r, layer, &t));
  const RoleShape rs = role_shape(m, r, layer);
  if (!rs.ok) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d is not part of this configuration (layer %d)", r, layer);
  const dsk_config& c = m->c;
  int base, local;
  shard_range(m, r, rs.e, &base, &local);

  if (is_scale) {
    if (rs.quant != DSK_QUANT_F8E5M2) DSK_FAIL(DSK_ERR_INVALID, "bind: scale tensor for a non-f8e5m2 weight");
    if (!t->bound()) DSK_FAIL(DSK_ERR_STATE, "bind: scale before its weight (role %d layer %d)", r, layer);
    const size_t per = (size_t)cdiv(rs.rows, c.block_size[0]) * cdiv(rs.n, c.block_size[1]);
    const size_t mats = rs.e > 0 ? rs.e : 1;
    if (bytes != per * mats * 4) DSK_FAIL(DSK_ERR_INVALID, "bind: scale bytes %zu, expected %zu", bytes, per * mats * 4);
    const size_t lm = rs.e > 0 ? (size_t)local : 1;
    if (lm) DSK_TRY(stage_copy(m->ctx, src, (uint64_t)base * per * 4, t->scale, lm * per * 4));
    t->scale_bound = true;
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
  t->tiled = role_tiled(m, r, rs.e, quant);
  DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], *t, quant, rs.e, rs.rows, rs.n, local, base));
  m->weight_bytes += (double)t->bytes;
  m->any_bound = true;
  return upload_tensor(m->ctx, *t, src);
}

// the plane layout persisted offline (tools/repack.py).  No DEVICE staging buffer and no repack kernel:
// every plane of the rank's experts is one contiguous byte range of the file that travels through the pinned host ring
// (stage_copy) straight into its device plane.
int bind_planes(dsk_model* m, int role, int layer, int quant, const HostSrc planes[4], const size_t bytes[4]) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "bind: null argument");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "bind after finalize");
  if (!is_kq(quant)) DSK_FAIL(DSK_ERR_INVALID, "bind_planes: k-quants only");
  HIP_TRY(hipSetDevice(m->ctx->device));
  DTensor* t;
  DSK_TRY(tensor_slot(m, role, layer, &t));
  const RoleShape rs = role_shape(m, role, layer);
  if (!rs.ok) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d is not part of this configuration (layer %d)", role, layer);
  if (quant != rs.quant) DSK_FAIL(DSK_ERR_INVALID, "bind: role %d layer %d has quant %d, expected %d", role, layer, quant, rs.quant);
  if (rs.n % 256) DSK_FAIL(DSK_ERR_INVALID, "bind: k-quant row length %d is not a multiple of 256", rs.n);
  if (t->bound()) DSK_FAIL(DSK_ERR_STATE, "bind: role %d layer %d bound twice", role, layer);
  const dsk_config& c = m->c;
  int base, local;
  shard_range(m, role, rs.e, &base, &local);
  const size_t mats = rs.e > 0 ? rs.e : 1, lm = rs.e > 0 ? (size_t)local : 1;
  const size_t nblk = (size_t)rs.rows * rs.n / 256;
  const size_t per[4] = {nblk * 64, nblk * (quant == DSK_QUANT_Q2_K ? 16 : 12), quant == DSK_QUANT_Q3_K ? nblk * 32 : 0,
                         nblk * (quant == DSK_QUANT_Q2_K ? 4 : 2)};
  for (int i = 0; i < 4; ++i)
    if (bytes[i] != per[i] * mats) DSK_FAIL(DSK_ERR_INVALID, "bind: plane %d of role %d layer %d has %zu bytes, expected %zu", i, role, layer, bytes[i], per[i] * mats);
  m->any_bound = true;
  if (role_tiled(m, role, rs.e, quant)) {
    // a tiled tensor from a planes-v1 checkpoint: the planes land in a temporary plane tensor and are re-laid-out on the device
    DTensor tmp;
    DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], tmp, quant, rs.e, rs.rows, rs.n, local, base));
    struct TmpTensor { DTensor t; ~TmpTensor() { if (t.base) hipFree(t.base); } } guard;
    guard.t = tmp;
    uint8_t* dstp[4] = {tmp.qs, tmp.sc, tmp.hm, tmp.dm};
    for (int i = 0; i < 4 && lm; ++i)
      if (per[i]) DSK_TRY(stage_copy(m->ctx, planes[i], (uint64_t)base * per[i], dstp[i], lm * per[i]));
    t->tiled = true;
    DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], *t, quant, rs.e, rs.rows, rs.n, local, base));
    m->weight_bytes += (double)t->bytes;
    if (lm) DSK_TRY(launch_planes_to_tiles_q2k(m->ctx->stream, tmp.qs, tmp.sc, tmp.dm, 0, lm * nblk, rs.rows, rs.n / 256, tile_mat_bytes(rs.rows, rs.n), t->qs));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return DSK_OK;
  }
  DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], *t, quant, rs.e, rs.rows, rs.n, local, base));
  m->weight_bytes += (double)t->bytes;
  uint8_t* dst[4] = {t->qs, t->sc, t->hm, t->dm};
  for (int i = 0; i < 4 && lm; ++i)
    if (per[i]) DSK_TRY(stage_copy(m->ctx, planes[i], (uint64_t)base * per[i], dst[i], lm * per[i]));
  return DSK_OK;
}

extern "C" int dsk_model_synthesize(dsk_model* m, uint64_t seed) {
  if (!m) DSK_FAIL(DSK_ERR_INVALID, "synthesize: null model");
  if (m->finalized) DSK_FAIL(DSK_ERR_STATE, "synthesize after finalize");
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const dsk_config& c = m->c;
  // One matrix of `view` (a plain tensor, or one expert of a stack).  Tile records are filled THROUGH the plane layout - the
  // same random planes, then re-laid-out on the device - so that a seed gives the same logical weights at every "q2k_tiles"
  // level (the layouts used to draw from different random streams; A/B runs across levels could not double as a
  // correctness check, and per-expert routing load differed between the compared runs).
  auto fill_matrix = [&](const DTensor& view, uint64_t sd, float wscale) -> int {
    if (!view.tiled) return launch_fill_tensor(st, view, sd, wscale);
    const size_t nblk = (size_t)view.rows * (view.n / 256);
    void* scratch = nullptr;
    DSK_TRY(ctx_scratch(m->ctx, 7, nblk * 84, &scratch));
    DTensor tmp = view;
    tmp.tiled = false;
    tmp.qs = static_cast<uint8_t*>(scratch);
    tmp.sc = tmp.qs + nblk * 64;
    tmp.dm = tmp.sc + nblk * 16;
    tmp.hm = nullptr;
    tmp.n_experts = 0; tmp.local_experts = 0; tmp.expert_base = 0;
    tmp.e_qs = tmp.e_sc = tmp.e_hm = tmp.e_dm = 0;
    DSK_TRY(launch_fill_tensor(st, tmp, sd, wscale));
    return launch_planes_to_tiles_q2k(st, tmp.qs, tmp.sc, tmp.dm, 0, nblk, view.rows, view.n / 256, tile_mat_bytes(view.rows, view.n), view.qs);
  };
  auto one = [&](int role, int layer) -> int {
    const RoleShape rs = role_shape(m, role, layer);
    if (!rs.ok) return DSK_OK;
    DTensor* t;
    DSK_TRY(tensor_slot(m, role, layer, &t));
    if (t->bound()) return DSK_OK;
    if (is_kq(rs.quant) && rs.n % 256) DSK_FAIL(DSK_ERR_INVALID, "synthesize: k-quant row length %d", rs.n);
    int base, local;
    shard_range(m, role, rs.e, &base, &local);
    t->tiled = role_tiled(m, role, rs.e, rs.quant);
    m->any_bound = true;
    DSK_TRY(alloc_tensor(c.block_size[0], c.block_size[1], *t, rs.quant, rs.e, rs.rows, rs.n, local, base));
    m->weight_bytes += (double)t->bytes;
    t->scale_bound = true;  // launch_fill_tensor writes the block scales of an F8 tensor too
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
        DSK_TRY(fill_matrix(one_e, s + (uint64_t)(base + le) * 0x100000001B3ull, wscale));
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
        DSK_TRY(fill_matrix(one_e, s + (uint64_t)le * 0x100000001B3ull, wscale));
      }
      return DSK_OK;
    }
    return fill_matrix(*t, s, wscale);
  };
  DSK_TRY(one(DSK_ROLE_EMBED, -1));
  DSK_TRY(one(DSK_ROLE_FINAL_NORM, -1));
  DSK_TRY(one(DSK_ROLE_OUTPUT, -1));
  for (int l = 0; l < c.n_layers; ++l)
    for (int role : ALL_LAYER_ROLES) DSK_TRY(one(role, l));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  if (m->ctx->op_buf[7]) {  // the plane staging of fill_matrix (as large as the classifier): not kept
    hipFree(m->ctx->op_buf[7]);
    m->ctx->op_buf[7] = nullptr;
    m->ctx->op_cap[7] = 0;
  }
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
  // F8E5M2 weights need their block scales (the reference asserts on a missing ".scale", src/model.cpp:191,862):
  // alloc_tensor reserves the scale plane with the weight, so "allocated" says nothing -- it must have been bound
  if (c.weight_quant == DSK_QUANT_F8E5M2) {
    if (!m->g[DSK_ROLE_EMBED].scale_bound) DSK_FAIL(DSK_ERR_STATE, "finalize: model.embed has no scale");
    if (!m->tied && !m->g[DSK_ROLE_OUTPUT].scale_bound) DSK_FAIL(DSK_ERR_STATE, "finalize: model.output has no scale");
  }
  for (int l = 0; l < c.n_layers; ++l)
    for (int role : ALL_LAYER_ROLES) {
      const RoleShape rs = role_shape(m, role, l);
      if (rs.ok && !m->L[l].t[role].bound()) DSK_FAIL(DSK_ERR_STATE, "finalize: layer %d role %d not bound", l, role);
      if (rs.ok && rs.quant == DSK_QUANT_F8E5M2 && !m->L[l].t[role].scale_bound) DSK_FAIL(DSK_ERR_STATE, "finalize: layer %d role %d has no scale", l, role);
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
  if (m->exchange_allgather && m->sharded())  // the all-gather form of the exchange: every rank's routed slot rows side by side
    DSK_TRY(alloc_f(m, &m->egather, (size_t)std::max(1, m->ctx->world) * std::max(1, c.n_active_routed) * c.dim));
  DSK_TRY(alloc_f(m, &m->q_c, (size_t)H * std::max(1, c.kv_lora_rank)));
  DSK_TRY(alloc_f(m, &m->q_rope, (size_t)H * std::max(1, c.qk_rope_head_dim)));
  DSK_TRY(alloc_f(m, &m->vb_out, (size_t)H * c.v_head_dim));
  // column slices per router row: 8 (2 rows per 16-wave workgroup) unless the rows are short
  m->router_ksplit = c.n_routed_experts > 0 ? 8 : 1;
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
  if (c.use_mla && c.kv_lora_rank == 512 && c.qk_rope_head_dim == 64) {  // long-context MLA on the matrix cores
    HIP_TRY(hipMalloc((void**)&m->fl_part_o, (size_t)64 * c.n_heads * c.kv_lora_rank * 4));
    HIP_TRY(hipMalloc((void**)&m->fl_part_ml, (size_t)64 * c.n_heads * 8));
    m->scratch_bytes += (double)64 * c.n_heads * (c.kv_lora_rank * 4 + 8);
  }
  if (!c.use_mla) {  // long-context MHA: enough workgroups per head to occupy the 256 CUs
    int S = 256 / std::max(1, c.n_heads);
    S = std::max(1, std::min(S, MHA_SPLIT_MAX));
    m->mha_split = S;
    if (m->mha_split_min <= 0) m->mha_split_min = MHA_SPLIT_MIN_KV;  // option "mha_split_min" overrides
    if (m->mha_split_min < 16 * S) m->mha_split_min = 16 * S;  // every split keeps >= 16 positions (the sink rows stay in split 0)
    if (S > 1) {
      const size_t pf = (size_t)c.n_heads * S * (c.v_head_dim + 2) * 4;
      HIP_TRY(hipMalloc((void**)&m->mha_split_part, pf));
      HIP_TRY(hipMalloc((void**)&m->mha_split_counter, (size_t)c.n_heads * 4));
      HIP_TRY(hipMemset(m->mha_split_counter, 0, (size_t)c.n_heads * 4));
      m->scratch_bytes += (double)pf;
    }
  }
  HIP_TRY(hipMalloc((void**)&m->argmax_dev, 64));
  HIP_TRY(hipMalloc((void**)&m->sample_scratch, sample_scratch_bytes()));
  HIP_TRY(hipMemset(m->sample_scratch, 0, sample_scratch_bytes()));
  HIP_TRY(hipHostMalloc((void**)&m->argmax_host, 64, hipHostMallocDefault));
  HIP_TRY(hipMalloc((void**)&m->router_counter, 64));
  HIP_TRY(hipMemset(m->router_counter, 0, 64));
  HIP_TRY(hipMalloc((void**)&m->att_counter, ((size_t)c.n_heads * c.v_head_dim / 256 + 2) * 4));
  HIP_TRY(hipMemset(m->att_counter, 0, ((size_t)c.n_heads * c.v_head_dim / 256 + 2) * 4));
  HIP_TRY(hipMalloc((void**)&m->comb_counter, (size_t)c.dim * 4));
  HIP_TRY(hipMemset(m->comb_counter, 0, (size_t)c.dim * 4));
  HIP_TRY(hipMalloc((void**)&m->moe_ctr, MOE_CTR_WORDS * 4));
  HIP_TRY(hipMemset(m->moe_ctr, 0, MOE_CTR_WORDS * 4));
  HIP_TRY(hipMalloc((void**)&m->moe_blk_ctr, MOE_BLK_CTRS * 4));
  HIP_TRY(hipMemset(m->moe_blk_ctr, 0, MOE_BLK_CTRS * 4));
  HIP_TRY(hipMalloc((void**)&m->moe_cand, MOE_CAND_BYTES));
  HIP_TRY(hipMemset(m->moe_cand, 0, MOE_CAND_BYTES));
  HIP_TRY(hipHostMalloc((void**)&m->err_host, 64, hipHostMallocDefault));
  memset(m->err_host, 0, 64);
  if (m->want_timeline) {
    HIP_TRY(hipMalloc((void**)&m->moe_timeline, (size_t)8 * DSK_TL_WGS * 8 * 8));
    HIP_TRY(hipMemset(m->moe_timeline, 0, (size_t)8 * DSK_TL_WGS * 8 * 8));
  }
  DSK_TRY(build_plans(m));
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
  for (int i = 0; i < 8; ++i)
    if (m->graph[i]) hipGraphExecDestroy(m->graph[i]);
  if (m->fl_part_o) hipFree(m->fl_part_o);
  if (m->fl_part_ml) hipFree(m->fl_part_ml);
  if (m->argmax_dev) hipFree(m->argmax_dev);
  if (m->sample_scratch) hipFree(m->sample_scratch);
  if (m->argmax_host) hipHostFree(m->argmax_host);
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
  free_plans(m);
  hydrate_free(m);
  for (void* p : {(void*)m->tap_qs, (void*)m->tap_d, (void*)m->tap_latent, (void*)m->stage_x_mid})
    if (p) hipFree(p);
  if (m->egather) hipFree(m->egather);
  if (m->moe_ctr) hipFree(m->moe_ctr);
  if (m->moe_blk_ctr) hipFree(m->moe_blk_ctr);
  if (m->moe_cand) hipFree(m->moe_cand);
  if (m->moe_timeline) hipFree(m->moe_timeline);
  if (m->err_host) hipHostFree(m->err_host);
  if (m->router_counter) hipFree(m->router_counter);
  if (m->comb_counter) hipFree(m->comb_counter);
  if (m->att_counter) hipFree(m->att_counter);
  if (m->mha_split_part) hipFree(m->mha_split_part);
  if (m->mha_split_counter) hipFree(m->mha_split_counter);
  free_q8(m->a_xb); free_q8(m->a_qa); free_q8(m->a_kva); free_q8(m->a_att); free_q8(m->a_hb);
  if (m->sp_host) hipHostFree(m->sp_host);
  if (m->logits_host) hipHostFree(m->logits_host);
  for (auto& k : m->ktimes)
    for (auto& e : k.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  dsk_ctx* ctx = m->ctx;
  delete m;
  ctx_model_released(ctx);  // frees the context if dsk_ctx_destroy was called while this model was alive
  return DSK_OK;
}

// diagnostics for the life-cycle tests: models alive on a context
extern "C" int dsk_ctx_live_models(dsk_ctx* c) { return c ? c->live_models : -1; }

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
