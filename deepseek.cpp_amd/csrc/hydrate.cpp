// hydrate.cpp -- dsk_hydrate: the prompt phase as batched launches.
//
// The reference ingests a prompt one token per forward (src/main.cpp:312-319: Model::forward(token, pos,
// InferenceMode::HYDRATE_KV_CACHE) for every prompt token but the last; src/infer.cpp:1284-1287 skips the classifier in that
// mode).  dsk_hydrate(model, tokens, n, pos0, mode, logits) is DEFINED as that loop - dsk_forward(tokens[i], pos0 + i,
// HYDRATE_KV_CACHE) for i < n - 1, then dsk_forward(tokens[n - 1], pos0 + n - 1, mode) - and, when the model qualifies, runs it
// as chunks of up to `hydrate_chunk` tokens through every block with each weight matrix read once per chunk
// (kernels_hydrate.hip): KV-cache rows, residual stream, routing and logits are bit-identical to the loop.
//
// Per block and chunk of P tokens (DeepSeek-V3, MHA path; the numbers are the decode launches they replace P times):
//   rmsnorm + Q8_K of P rows                      (the prologue of launch 1)
//   GEMM wq_a, wkv_a                              (launch 1)
//   latent norms + Q8_K                           (prologue of the per-head launch 2)
//   GEMM wq_b, wkv_b                              (launch 2, projections)
//   K / V rows of the P positions -> cache; rope(q) + causal attention per (head, token); Q8_K of the outputs   (launch 2)
//   GEMM wo, x += .                               (launch 3)
//   router + gate per token (+ Q8_K of rmsnorm(x, ffn_norm)); tokens grouped by expert       (launch 4)
//   GEMM shared w1/w3 GLU -> Q8_K -> GEMM shared w2                                            (launches 4 / 5)
//   grouped GEMM experts w1/w3 GLU over the (token, slot) pairs of each expert -> Q8_K -> grouped GEMM w2    (launch 5)
//   x += sum_k w_k out_k (k order) + shared                                                    (launch 5's combine)
// A model that does not qualify (float weights, plane layout, MLA, expert-sharded) and positions at or past the ring wrap
// (the in-place sink rotation of src/infer.cpp:1008-1020 is sequential by nature) take the loop itself.
#include "engine.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>

int fill_step_params_at(dsk_model* m, int token, int pos, StepParams* sp);  // forward.cpp
int forward_head(dsk_model* m);                                           // forward.cpp: final norm + classifier on m->x, logits to the pinned host buffer

struct HydQ8 {
  int8_t* qs = nullptr;
  float* d = nullptr;
  int16_t* bsums = nullptr;
  unsigned* dig = nullptr;  // the sums as digit words (the inputs of the grouped expert GEMMs only)
};
struct HydState {
  int cap = 0;
  float *X = nullptr, *q_a = nullptr, *kv_a = nullptr, *q = nullptr, *kv_b = nullptr, *att = nullptr, *hbd = nullptr, *hb = nullptr,
        *hb_sh = nullptr, *eout = nullptr, *eout_sh = nullptr, *route_w = nullptr, *router_partial = nullptr, *trace = nullptr;
  HydQ8 a_x, a_qa, a_kva, a_att, a_hd, a_hb, a_hsh, a_lat;
  float *q_rope = nullptr, *q_c = nullptr, *latent = nullptr;  // MLA: (P, H * rope), (P, H * lora), (P, H * lora)
  int *head_list = nullptr, *head_count = nullptr;              // MLA: wv_b as one GEMM task per head
  // MLA, tokens whose context has reached the matrix-core regime (mla_flash_min_kv): chunk partials of HYD_FL_TOKENS tokens at a time
  float *fl_part_o = nullptr, *fl_part_ml = nullptr;
  StepParams *sp = nullptr, *sp_host = nullptr;  // the chunk's step rows; the pinned side is two halves used in turn (sp_done)
  hipEvent_t sp_done[2] = {nullptr, nullptr};     // the upload that last read half k
  int sp_turn = 0;
  unsigned* router_counter = nullptr;
  int *route_e = nullptr, *list = nullptr, *count = nullptr;
  int last_P = 0;                                 // tokens of the last batched chunk (what the accessors may hand out)
  // parity harness (option "hydrate_tap_layer"): copies of ONE block's intermediates, taken while the chunk runs on
  std::map<std::string, std::pair<void*, size_t>> taps;  // name -> (device copy of `cap` rows, bytes per row)
  // Q2_K matrices the decode launches keep as PLANES (option "q2k_tiles" = 1: every role but the experts'), re-laid-out as tile
  // records for the batched GEMMs when the first prompt arrives (hyd_tile_copies below)
  std::map<const DTensor*, const uint8_t*> tile_copy;
  double tile_copy_bytes = 0;
  std::vector<void*> allocs;
};

#define HYD_FL_TOKENS 16  // tokens per mla_flash launch of the batched path: 16 x 64 chunks x 128 heads x 512 floats = 268 MB of partials

static int hyd_alloc(HydState* h, void** p, size_t bytes, double* total) {
  if (bytes == 0) bytes = 16;
  HIP_TRY(hipMalloc(p, bytes));
  h->allocs.push_back(*p);
  *total += (double)bytes;
  return DSK_OK;
}
static int hyd_alloc_q8(HydState* h, HydQ8& q, size_t rows, size_t n, double* total) {
  DSK_TRY(hyd_alloc(h, (void**)&q.qs, rows * n, total));
  DSK_TRY(hyd_alloc(h, (void**)&q.d, rows * (n / 256) * 4, total));
  DSK_TRY(hyd_alloc(h, (void**)&q.bsums, rows * (n / 16) * 2, total));
  return DSK_OK;
}

void hydrate_free(dsk_model* m) {
  if (!m->hyd) return;
  for (void* p : m->hyd->allocs) hipFree(p);
  if (m->hyd->sp_host) hipHostFree(m->hyd->sp_host);
  for (hipEvent_t e : m->hyd->sp_done) if (e) hipEventDestroy(e);
  delete m->hyd;
  m->hyd = nullptr;
  m->hydrate_tile_copy_bytes = 0;
}

// The tile records of a matrix: the tensor itself when it is stored that way, else the batched path's copy.
static const uint8_t* tile_w(const HydState& h, const DTensor& t) {
  if (t.tiled) return t.qs;
  const auto it = h.tile_copy.find(&t);
  return it == h.tile_copy.end() ? nullptr : it->second;
}
// a plane-layout Q2_K matrix the batched path may copy into tile records (option "hydrate_tile_copies")
static bool tile_copyable(const dsk_model* m, const DTensor& t) {
  return m->hydrate_tile_copies && t.bound() && !t.tiled && t.quant == DSK_QUANT_Q2_K && t.n_experts == 0 && t.n % 256 == 0 && t.n / 256 <= 255 && t.qs && t.sc && t.dm;
}
// the matrices of block l the batched path multiplies as plain (one-matrix) GEMMs
static void hyd_plain_roles(const dsk_model* m, int l, std::vector<int>& roles) {
  const dsk_config& c = m->c;
  const Layer& L = m->L[l];
  roles = {DSK_ROLE_WQ_A, DSK_ROLE_WKV_A, DSK_ROLE_WO};
  if (c.use_mla) roles.insert(roles.end(), {DSK_ROLE_WQ_ROPE_B, DSK_ROLE_WC, DSK_ROLE_WV_B});
  else roles.insert(roles.end(), {DSK_ROLE_WQ_B, DSK_ROLE_WKV_B});
  if (!L.is_moe) roles.insert(roles.end(), {DSK_ROLE_W1, DSK_ROLE_W2, DSK_ROLE_W3});
  else if (c.n_shared_experts > 0) roles.insert(roles.end(), {DSK_ROLE_SHARED_W1, DSK_ROLE_SHARED_W2, DSK_ROLE_SHARED_W3});
}

// why this model takes the per-token loop (nullptr: the batched path applies)
static const char* hyd_why_not(const dsk_model* m) {
  const dsk_config& c = m->c;
  if (c.weight_quant != DSK_QUANT_Q2_K) return "weights are not Q2_K";
  if (c.q_lora_rank <= 0) return "no q latent";
  if (c.use_mla && (c.kv_lora_rank > 512 || c.kv_lora_rank % 64 || c.qk_rope_head_dim > 64 || c.qk_rope_head_dim % 4 || c.v_head_dim % 16))
    return "MLA head shapes";
  if (m->sharded()) return "expert-sharded model";
  if (c.dim % 256 || (c.n_heads * c.v_head_dim) % 256 || c.hidden_dim % 256) return "vector lengths";
  if (m->head_dim > 256 || c.v_head_dim > 256 || (c.v_head_dim & 3) || (m->head_dim & 3)) return "head dims";
  if (c.q_lora_rank % 256 || c.kv_lora_rank % 256 || c.q_lora_rank / 256 + c.kv_lora_rank / 256 > 16) return "latent ranks";
  if (c.n_routed_experts > 256 || c.n_active_routed > 64) return "expert counts";
  // tile records: stored that way (option "q2k_tiles"), or - the plain matrices only, never the expert stacks - copied into that
  // layout when the first prompt arrives (option "hydrate_tile_copies": ~4.4 GB next to DeepSeek-V3's 242)
  std::vector<int> roles;
  for (int l = 0; l < c.n_layers; ++l) {
    const Layer& L = m->L[l];
    hyd_plain_roles(m, l, roles);
    for (int role : roles)
      if (!(L.t[role].bound() && (L.t[role].tiled || tile_copyable(m, L.t[role]))))
        return "a Q2_K matrix is neither stored as tile records nor copyable into them (options q2k_tiles, hydrate_tile_copies)";
    if (L.is_moe)
      for (int role : {DSK_ROLE_W1, DSK_ROLE_W2, DSK_ROLE_W3})
        if (!(L.t[role].bound() && L.t[role].tiled)) return "the routed experts are not stored as tile records (option q2k_tiles >= 1 and shapes the tiled expert kernels take)";
    if (L.is_moe && c.moe_intermediate_size % 256) return "moe_intermediate_size";
    // the norm + Q8_K launches reproduce the sum-of-squares tree of the decode launch that consumes the vector: built for 4 / 8 / 16 waves
    for (int lp : {m->lp_qkv_a[l], c.use_mla ? m->lp_qkv_b[l] : -1, L.is_moe ? -1 : m->lp_w13[l]})
      if (lp >= 0 && m->plans[lp].NW != 4 && m->plans[lp].NW != 8 && m->plans[lp].NW != 16) return "a decode launch stages its vector with a workgroup size the batched path has no twin for";
    if (m->lp_qkv_a[l] < 0 || (c.use_mla && m->lp_qkv_b[l] < 0) || (!L.is_moe && m->lp_w13[l] < 0)) return "a decode launch plan is missing";
  }
  return nullptr;
}

static int hyd_ensure_alloc(dsk_model* m);
static int hyd_ensure(dsk_model* m) {
  if (m->hyd) return DSK_OK;
  const int r = hyd_ensure_alloc(m);
  if (r != DSK_OK) {  // (a half-allocated state must not survive: the next call would take it for complete)
    const std::string keep = dsk_last_error();
    hydrate_free(m);
    DSK_FAIL(r, "%s", keep.c_str());
  }
  return DSK_OK;
}
static int hyd_ensure_alloc(dsk_model* m) {
  const dsk_config& c = m->c;
  HydState* h = new HydState();
  m->hyd = h;
  const size_t P = (size_t)std::max(1, m->hydrate_chunk);
  h->cap = (int)P;
  const size_t dim = c.dim, H = c.n_heads, hd = m->head_dim, nv = c.qk_nope_head_dim + c.v_head_dim, vd = c.v_head_dim;
  const size_t K = std::max(1, c.n_active_routed), E = std::max(1, c.n_routed_experts), mi = std::max(256, c.moe_intermediate_size);
  const size_t shn = std::max<size_t>(256, (size_t)c.n_shared_experts * c.moe_intermediate_size);
  double tot = 0;
  DSK_TRY(hyd_alloc(h, (void**)&h->X, P * dim * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->q_a, P * c.q_lora_rank * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->kv_a, P * (c.kv_lora_rank + c.qk_rope_head_dim) * 4, &tot));
  if (c.use_mla) {
    const size_t lora = c.kv_lora_rank;
    DSK_TRY(hyd_alloc(h, (void**)&h->q_rope, P * H * c.qk_rope_head_dim * 4, &tot));
    DSK_TRY(hyd_alloc(h, (void**)&h->q_c, P * H * lora * 4, &tot));
    DSK_TRY(hyd_alloc(h, (void**)&h->latent, P * H * lora * 4, &tot));
    DSK_TRY(hyd_alloc_q8(h, h->a_lat, P * H, lora, &tot));
    DSK_TRY(hyd_alloc(h, (void**)&h->head_list, H * P * 4, &tot));
    DSK_TRY(hyd_alloc(h, (void**)&h->head_count, H * 4, &tot));
    if (m->fl_part_o) {
      DSK_TRY(hyd_alloc(h, (void**)&h->fl_part_o, (size_t)HYD_FL_TOKENS * 64 * H * lora * 4, &tot));
      DSK_TRY(hyd_alloc(h, (void**)&h->fl_part_ml, (size_t)HYD_FL_TOKENS * 64 * H * 8, &tot));
    }
  } else {
    DSK_TRY(hyd_alloc(h, (void**)&h->q, P * H * hd * 4, &tot));
    DSK_TRY(hyd_alloc(h, (void**)&h->kv_b, P * H * nv * 4, &tot));
  }
  DSK_TRY(hyd_alloc(h, (void**)&h->att, P * H * vd * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->hbd, P * std::max<size_t>(c.hidden_dim, dim) * 4, &tot));  // (also the router's normed vectors)
  DSK_TRY(hyd_alloc(h, (void**)&h->hb, P * K * mi * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->hb_sh, P * shn * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->eout, P * K * dim * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->eout_sh, P * dim * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->route_w, P * K * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->route_e, P * K * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->router_partial, P * E * 4 + 64, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->router_counter, P * 4, &tot));
  HIP_TRY(hipMemset(h->router_counter, 0, P * 4));
  DSK_TRY(hyd_alloc(h, (void**)&h->list, E * P * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->count, E * 4, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_x, P, dim, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->a_x.dig, P * (dim / 16) * 4, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->a_hb.dig, P * K * (mi / 16) * 4, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_qa, P, c.q_lora_rank, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_kva, P, c.kv_lora_rank, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_att, P, H * vd, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_hd, P, c.hidden_dim, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_hb, P * K, mi, &tot));
  DSK_TRY(hyd_alloc_q8(h, h->a_hsh, P, shn, &tot));
  DSK_TRY(hyd_alloc(h, (void**)&h->sp, P * sizeof(StepParams), &tot));
  HIP_TRY(hipHostMalloc((void**)&h->sp_host, 2 * P * sizeof(StepParams), hipHostMallocDefault));
  for (hipEvent_t& e : h->sp_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (m->trace) DSK_TRY(hyd_alloc(h, (void**)&h->trace, (size_t)c.n_layers * P * dim * 4, &tot));
  m->scratch_bytes += tot;
  // tile-record copies of the plane matrices (nothing to do at "q2k_tiles" = 2); the decode launches keep reading the planes
  std::vector<int> roles;
  for (int l = 0; l < c.n_layers; ++l) {
    hyd_plain_roles(m, l, roles);
    for (int role : roles) {
      const DTensor& t = m->L[l].t[role];
      if (t.tiled) continue;
      const size_t bytes = tile_mat_bytes(t.rows, t.n);
      void* p = nullptr;
      double cb = 0;
      DSK_TRY(hyd_alloc(h, &p, bytes, &cb));
      if (t.rows & 15) HIP_TRY(hipMemsetAsync(p, 0, bytes, m->ctx->stream));  // (padding rows of the last strip stay zero)
      DSK_TRY(launch_planes_to_tiles_q2k(m->ctx->stream, t.qs, t.sc, t.dm, 0, (size_t)t.rows * (t.n / 256), t.rows, t.n / 256, bytes, static_cast<uint8_t*>(p)));
      h->tile_copy[&t] = static_cast<const uint8_t*>(p);
      h->tile_copy_bytes += cb;
    }
  }
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  HIP_TRY(hipGetLastError());
  m->hydrate_tile_copy_bytes = h->tile_copy_bytes;
  return DSK_OK;
}

// Parity harness (option "hydrate_tap_layer" = l): while block l of a chunk runs, its intermediates are COPIED aside - the chunk
// itself is untouched (nothing is skipped, nothing is replaced), so a tapped call leaves the same caches and logits as any other.
// `rows` rows of `row_bytes` each; the copy holds `cap` rows (dsk_hydrate_get_buffer reads the last chunk's).
static int hyd_tap(dsk_model* m, int l, const char* name, const void* src, size_t rows, size_t row_bytes) {
  if (l != m->hydrate_tap_layer || !src) return DSK_OK;
  HydState& h = *m->hyd;
  auto it = h.taps.find(name);
  if (it == h.taps.end()) {
    void* p = nullptr;
    double tot = 0;
    DSK_TRY(hyd_alloc(&h, &p, (size_t)h.cap * row_bytes, &tot));
    m->scratch_bytes += tot;
    it = h.taps.emplace(name, std::make_pair(p, row_bytes)).first;
  }
  if (it->second.second != row_bytes) DSK_FAIL(DSK_ERR_STATE, "hydrate tap '%s': row size changed", name);
  HIP_TRY(hipMemcpyAsync(it->second.first, src, rows * row_bytes, hipMemcpyDeviceToDevice, m->ctx->stream));
  return DSK_OK;
}
static int hyd_tap_q8(dsk_model* m, int l, const std::string& point, const HydQ8& q, size_t rows, size_t n) {
  if (l != m->hydrate_tap_layer) return DSK_OK;
  DSK_TRY(hyd_tap(m, l, ("q8." + point + ".qs").c_str(), q.qs, rows, n));
  DSK_TRY(hyd_tap(m, l, ("q8." + point + ".d").c_str(), q.d, rows, (n / 256) * 4));
  return DSK_OK;
}

static int hyd_gemm(dsk_model* m, const DTensor& w, const DTensor* w3, const HydQ8& a, int a_rows, int P, float* out, int out_stride, int epilogue) {
  HydGemmArgs A;
  memset(&A, 0, sizeof A);
  A.W = tile_w(*m->hyd, w); A.W3 = w3 ? tile_w(*m->hyd, *w3) : nullptr;
  if (!A.W || (w3 && !A.W3)) DSK_FAIL(DSK_ERR_STATE, "hydrate: a matrix has no tile records");
  A.rows = w.rows; A.n = w.n;
  A.a_qs = a.qs; A.a_d = a.d; A.a_bsums = a.bsums; A.a_rows = a_rows; A.a_div = 1;
  A.m = P; A.out = out; A.out_stride = out_stride; A.epilogue = epilogue; A.act = m->c.act;
  // plain matrices: 8 tokens per wave and pass (168 / 236 VGPRs: 3 / 2 waves per SIMD; 16 tokens need 260 / 366 and leave one),
  // the four waves of a workgroup on consecutive chunks of the same strip
  return launch_hyd_gemm(m->ctx->stream, A, P <= 4 ? 1 : 2);
}

// n_short: the chunk's first tokens whose context is below the MLA matrix-core regime (all of them for MHA)
static int hyd_layer(dsk_model* m, int l, int P, int max_kv, int n_short) {
  const dsk_config& c = m->c;
  HydState& h = *m->hyd;
  Layer& L = m->L[l];
  hipStream_t st = m->ctx->stream;
  const int dim = c.dim, qlr = c.q_lora_rank, kvl = c.kv_lora_rank, rope = c.qk_rope_head_dim, H = c.n_heads, hd = m->head_dim;
  const int nv = c.qk_nope_head_dim + c.v_head_dim, vd = c.v_head_dim;
  auto f32w = [](const DTensor& t) { return reinterpret_cast<const float*>(t.qs); };
  // ---- attention half (src/infer.cpp:823-834, 934-1049) ----
  DSK_TRY(launch_hyd_norm_q8(st, m->plans[m->lp_qkv_a[l]].NW, h.X, P, dim, f32w(L.t[DSK_ROLE_ATTN_NORM]), c.norm_eps, h.a_x.qs, h.a_x.d, h.a_x.bsums));
  DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WQ_A], nullptr, h.a_x, P, P, h.q_a, qlr, EPI_STORE));
  DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WKV_A], nullptr, h.a_x, P, P, h.kv_a, kvl + rope, EPI_STORE));
  DSK_TRY(hyd_tap_q8(m, l, "x_attn", h.a_x, P, dim));
  DSK_TRY(hyd_tap(m, l, "q_a", h.q_a, P, (size_t)qlr * 4));
  DSK_TRY(hyd_tap(m, l, "kv_a", h.kv_a, P, (size_t)(kvl + rope) * 4));
  if (c.use_mla) {
    // second stage on norm(q_a): wq_rope_b, wc (the prologue of gemv_kvwrite_tile_kernel: stage_q8 at that plan's workgroup size)
    DSK_TRY(launch_hyd_norm_q8(st, m->plans[m->lp_qkv_b[l]].NW, h.q_a, P, qlr, f32w(L.t[DSK_ROLE_Q_A_NORM]), c.norm_eps, h.a_qa.qs, h.a_qa.d, h.a_qa.bsums));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WQ_ROPE_B], nullptr, h.a_qa, P, P, h.q_rope, H * rope, EPI_STORE));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WC], nullptr, h.a_qa, P, P, h.q_c, H * kvl, EPI_STORE));
    MlaKvArgs kv;
    kv.kv_a = h.kv_a; kv.norm_w = f32w(L.t[DSK_ROLE_KV_A_NORM]); kv.eps = c.norm_eps;
    kv.nope_cache = L.nope_cache; kv.rope_cache = L.rope_cache; kv.lora = kvl; kv.rope = rope; kv.is_v3 = c.has_moegate_bias;
    DSK_TRY(launch_hyd_mla_kv_write(st, kv, h.sp, P, kvl + rope));
    const AttnMlaArgs& am = m->mla_head[l].a;
    if (n_short > 0) DSK_TRY(launch_hyd_mla_attn(st, am, h.sp, n_short, max_kv - (P - n_short), h.q_c, H * kvl, h.q_rope, H * rope, h.latent, H * kvl));
    for (int t0 = n_short; t0 < P; t0 += HYD_FL_TOKENS) {  // decode's long-context regime, token by token in one grid (src/infer.cpp:766-804)
      const int nb = std::min(HYD_FL_TOKENS, P - t0);
      MlaFlashArgs F = m->mla_flash[l];
      F.q_c = h.q_c + (size_t)t0 * H * kvl; F.q_rope = h.q_rope + (size_t)t0 * H * rope; F.tok_qc_stride = H * kvl; F.tok_qr_stride = H * rope;
      F.part_o = h.fl_part_o; F.part_ml = h.fl_part_ml; F.timeline = nullptr; F.min_kv = 0;
      DSK_TRY(launch_mla_flash(st, F, h.sp + t0, 0, nb));
      DSK_TRY(launch_hyd_mla_merge(st, F, h.sp + t0, nb, h.latent + (size_t)t0 * H * kvl, H * kvl));
    }
    DSK_TRY(launch_quantize_q8k(st, h.latent, P * H * kvl, h.a_lat.qs, h.a_lat.d, h.a_lat.bsums));
    {  // per-head wv_b (src/infer.cpp:1134-1137): one task per head over all tokens
      const DTensor& wv = L.t[DSK_ROLE_WV_B];
      DSK_TRY(launch_hyd_head_list(st, h.head_list, h.head_count, H, P, h.cap));
      HydGemmArgs A;
      memset(&A, 0, sizeof A);
      A.W = tile_w(h, wv); A.e_bytes = tile_mat_bytes(vd, kvl); A.n_experts = H; A.rows = vd; A.n = kvl;
      A.a_qs = h.a_lat.qs; A.a_d = h.a_lat.d; A.a_bsums = h.a_lat.bsums; A.a_rows = P * H; A.a_div = 1;
      A.list = h.head_list; A.count = h.head_count; A.list_stride = h.cap;
      A.out = h.att; A.out_stride = vd; A.epilogue = EPI_STORE; A.act = c.act;
      DSK_TRY(launch_hyd_gemm(st, A, P <= 4 ? 1 : 2));
    }
    DSK_TRY(hyd_tap_q8(m, l, "q_a", h.a_qa, P, qlr));
    DSK_TRY(hyd_tap(m, l, "q_rope", h.q_rope, P, (size_t)H * rope * 4));
    DSK_TRY(hyd_tap(m, l, "q_c", h.q_c, P, (size_t)H * kvl * 4));
    DSK_TRY(hyd_tap(m, l, "latent_out", h.latent, P, (size_t)H * kvl * 4));
    DSK_TRY(hyd_tap_q8(m, l, "latent", h.a_lat, P, (size_t)H * kvl));
    DSK_TRY(hyd_tap(m, l, "vb_out", h.att, P, (size_t)H * vd * 4));
  } else {
    HydLatentArgs A;
    memset(&A, 0, sizeof A);
    A.q_a = h.q_a; A.kv_a = h.kv_a; A.q_norm = f32w(L.t[DSK_ROLE_Q_A_NORM]); A.kv_norm = f32w(L.t[DSK_ROLE_KV_A_NORM]);
    A.q_stride = qlr; A.kv_stride = kvl + rope; A.nq = qlr; A.nkv = kvl; A.eps = c.norm_eps;
    A.qq_qs = h.a_qa.qs; A.qq_d = h.a_qa.d; A.qq_bsums = h.a_qa.bsums;
    A.kq_qs = h.a_kva.qs; A.kq_d = h.a_kva.d; A.kq_bsums = h.a_kva.bsums;
    DSK_TRY(launch_hyd_latent_q8(st, A, P));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WQ_B], nullptr, h.a_qa, P, P, h.q, H * hd, EPI_STORE));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WKV_B], nullptr, h.a_kva, P, P, h.kv_b, H * nv, EPI_STORE));
    const AttnMhaArgs& a = m->head_attn[l].a;
    DSK_TRY(launch_hyd_kv_write(st, a, h.sp, P, h.kv_b, H * nv, h.kv_a, kvl + rope));
    DSK_TRY(launch_hyd_attn(st, a, h.sp, P, max_kv, h.q, H * hd, h.att, H * vd, m->mha_split_part && m->mha_split_counter ? m->mha_split : 1, m->mha_split_min));
    DSK_TRY(hyd_tap_q8(m, l, "q_a", h.a_qa, P, qlr));
    DSK_TRY(hyd_tap_q8(m, l, "kv_a", h.a_kva, P, kvl));
    DSK_TRY(hyd_tap(m, l, "q", h.q, P, (size_t)H * hd * 4));
    DSK_TRY(hyd_tap(m, l, "kv_b", h.kv_b, P, (size_t)H * nv * 4));
    DSK_TRY(hyd_tap(m, l, "att_out", h.att, P, (size_t)H * vd * 4));
  }
  DSK_TRY(launch_quantize_q8k(st, h.att, P * H * vd, h.a_att.qs, h.a_att.d, h.a_att.bsums));
  DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_WO], nullptr, h.a_att, P, P, h.X, dim, EPI_ADD));
  DSK_TRY(hyd_tap_q8(m, l, "att", h.a_att, P, (size_t)H * vd));
  DSK_TRY(hyd_tap(m, l, "x_mid", h.X, P, (size_t)dim * 4));
  // ---- FFN half (src/infer.cpp:836-931) ----
  if (!L.is_moe) {
    DSK_TRY(launch_hyd_norm_q8(st, m->plans[m->lp_w13[l]].NW, h.X, P, dim, f32w(L.t[DSK_ROLE_FFN_NORM]), c.norm_eps, h.a_x.qs, h.a_x.d, h.a_x.bsums));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_W1], &L.t[DSK_ROLE_W3], h.a_x, P, P, h.hbd, c.hidden_dim, EPI_STORE));
    DSK_TRY(launch_quantize_q8k(st, h.hbd, P * c.hidden_dim, h.a_hd.qs, h.a_hd.d, h.a_hd.bsums));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_W2], nullptr, h.a_hd, P, P, h.X, dim, EPI_ADD));
    DSK_TRY(hyd_tap_q8(m, l, "x_ffn", h.a_x, P, dim));
    DSK_TRY(hyd_tap(m, l, "hb", h.hbd, P, (size_t)c.hidden_dim * 4));
    DSK_TRY(hyd_tap_q8(m, l, "hb", h.a_hd, P, c.hidden_dim));
    return DSK_OK;
  }
  const int K = c.n_active_routed, E = c.n_routed_experts, mi = c.moe_intermediate_size, shn = c.n_shared_experts * mi;
  {
    RouterArgs r;
    memset(&r, 0, sizeof r);
    r.w = f32w(L.t[DSK_ROLE_MOEGATE]);
    r.x = h.X;
    r.norm_w = f32w(L.t[DSK_ROLE_FFN_NORM]);
    r.eps = c.norm_eps;
    r.n_routed = E; r.dim = dim; r.ksplit = m->router_ksplit;
    r.partial = h.router_partial; r.counter = h.router_counter;
    r.bias = L.t[DSK_ROLE_MOEGATE_BIAS].bound() ? f32w(L.t[DSK_ROLE_MOEGATE_BIAS]) : nullptr;
    r.n_active = K; r.norm_topk_prob = c.norm_topk_prob; r.scoring = c.scoring_func; r.topk_method = c.topk_method;
    r.n_group = c.n_group; r.topk_group = c.topk_group; r.scaling = c.routed_scaling_factor;
    r.active_experts = h.route_e; r.active_weights = h.route_w;
    r.q_qs = h.a_x.qs; r.q_d = h.a_x.d; r.q_bsums = h.a_x.bsums;
    DSK_TRY(launch_hyd_router(st, r, P, h.hbd));  // (the dense FFN's hidden buffer holds the normed vectors: P x dim <= P x hidden_dim)
  }
#ifdef DSK_AB  // measurement builds only (tools/ab_build.sh): uniform routing instead of the synthetic model's skewed one
  if (m->hydrate_route_seed > 0) DSK_TRY(launch_hyd_route_override(st, h.route_e, P, K, E, (unsigned)m->hydrate_route_seed * 1000003u + (unsigned)l));
#endif
  DSK_TRY(launch_hyd_group(st, h.route_e, P * K, E, h.list, h.cap, h.count));
  if (shn > 0) {
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_SHARED_W1], &L.t[DSK_ROLE_SHARED_W3], h.a_x, P, P, h.hb_sh, shn, EPI_STORE));
    DSK_TRY(launch_quantize_q8k(st, h.hb_sh, P * shn, h.a_hsh.qs, h.a_hsh.d, h.a_hsh.bsums));
    DSK_TRY(hyd_gemm(m, L.t[DSK_ROLE_SHARED_W2], nullptr, h.a_hsh, P, P, h.eout_sh, dim, EPI_STORE));
  }
  const DTensor &w1 = L.t[DSK_ROLE_W1], &w2 = L.t[DSK_ROLE_W2], &w3 = L.t[DSK_ROLE_W3];
  // 4 tokens per wave and pass unless the experts are crowded: 8 tokens cost the GLU pair half of its waves (260 registers), while a
  // second pass over a strip re-reads 75 KB that the first pass just pulled through L2
  const int nq_e = 1;  // (tasks of 6 rows and more go through the 16-token form: what is left for the quads fits one or two passes)
  {
    HydGemmArgs A;
    memset(&A, 0, sizeof A);
    A.W = w1.qs; A.W3 = w3.qs; A.e_bytes = w1.e_qs; A.n_experts = E; A.rows = mi; A.n = dim;
    A.a_qs = h.a_x.qs; A.a_d = h.a_x.d; A.a_bsums = h.a_x.bsums; A.a_rows = P; A.a_div = K;
    DSK_TRY(launch_hyd_digits(st, h.a_x.bsums, h.a_x.dig, (size_t)P * (dim / 16)));
    A.a_dig = h.a_x.dig;
    A.list = h.list; A.count = h.count; A.list_stride = h.cap;
    A.out = h.hb; A.out_stride = mi; A.act = c.act;
    DSK_TRY(launch_hyd_gemm(st, A, nq_e));
  }
  DSK_TRY(launch_quantize_q8k(st, h.hb, P * K * mi, h.a_hb.qs, h.a_hb.d, h.a_hb.bsums));
  DSK_TRY(launch_hyd_digits(st, h.a_hb.bsums, h.a_hb.dig, (size_t)P * K * (mi / 16)));
  {
    HydGemmArgs A;
    memset(&A, 0, sizeof A);
    A.W = w2.qs; A.e_bytes = w2.e_qs; A.n_experts = E; A.rows = dim; A.n = mi;
    A.a_qs = h.a_hb.qs; A.a_d = h.a_hb.d; A.a_bsums = h.a_hb.bsums; A.a_rows = P * K; A.a_div = 1;
    A.a_dig = h.a_hb.dig;
    A.list = h.list; A.count = h.count; A.list_stride = h.cap;
    A.out = h.eout; A.out_stride = dim; A.epilogue = EPI_STORE; A.act = c.act;
    DSK_TRY(launch_hyd_gemm(st, A, nq_e));
  }
  DSK_TRY(launch_hyd_combine(st, h.X, h.eout, h.route_w, shn > 0 ? h.eout_sh : nullptr, P, K, dim));
  if (l == m->hydrate_tap_layer) {
    DSK_TRY(hyd_tap_q8(m, l, "x_ffn", h.a_x, P, dim));
    DSK_TRY(hyd_tap(m, l, "router_logits", h.router_partial, P, (size_t)E * 4));  // (the row-resident router form: launch_hyd_router)
    DSK_TRY(hyd_tap(m, l, "route_e", h.route_e, P, (size_t)K * 4));
    DSK_TRY(hyd_tap(m, l, "route_w", h.route_w, P, (size_t)K * 4));
    DSK_TRY(hyd_tap(m, l, "hb", h.hb, P, (size_t)K * mi * 4));
    DSK_TRY(hyd_tap_q8(m, l, "hb", h.a_hb, P, (size_t)K * mi));
    DSK_TRY(hyd_tap(m, l, "eout", h.eout, P, (size_t)K * dim * 4));
    if (shn > 0) {
      DSK_TRY(hyd_tap(m, l, "hb_sh", h.hb_sh, P, (size_t)shn * 4));
      DSK_TRY(hyd_tap_q8(m, l, "hb_sh", h.a_hsh, P, shn));
      DSK_TRY(hyd_tap(m, l, "eout_sh", h.eout_sh, P, (size_t)dim * 4));
    }
  }
  return DSK_OK;
}

// one chunk of P tokens at positions pos0 .. pos0 + P - 1 through every block (no ring wrap inside: the caller checked)
static int hyd_chunk(dsk_model* m, const int32_t* tokens, int P, int pos0) {
  const dsk_config& c = m->c;
  HydState& h = *m->hyd;
  hipStream_t st = m->ctx->stream;
  // the pinned step-parameter rows: two halves used in turn, so that the host fills the next chunk's rows while the device still
  // runs this one (only the upload that last read THIS half must have finished - the chunk before the previous one's)
  const int half = h.sp_turn;
  h.sp_turn ^= 1;
  StepParams* rows = h.sp_host + (size_t)half * h.cap;
  HIP_TRY(hipEventSynchronize(h.sp_done[half]));
  for (int p = 0; p < P; ++p) DSK_TRY(fill_step_params_at(m, tokens[p], pos0 + p, rows + p));
  HIP_TRY(hipMemcpyAsync(h.sp, rows, (size_t)P * sizeof(StepParams), hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(h.sp_done[half], st));
  // Model::_copy_embedding, src/infer.cpp:1217-1263: the P rows in one launch (the tokens come from the step rows)
  DSK_TRY(launch_embed_rows(st, m->g[DSK_ROLE_EMBED], h.sp, P, std::max(1, c.block_size[0]), std::max(1, c.block_size[1]), h.X));
  const int max_kv = pos0 + P;
  int n_short = P;
  if (c.use_mla && m->fl_part_o) {
    n_short = 0;
    while (n_short < P && rows[n_short].kv_len < m->mla_flash_min_kv) ++n_short;
    for (int p = n_short; p < P; ++p)
      if (rows[p].kv_len < m->mla_flash_min_kv) DSK_FAIL(DSK_ERR_STATE, "hydrate: context lengths of a chunk are not ascending");
  }
  for (int l = 0; l < c.n_layers; ++l) {
    DSK_TRY(hyd_layer(m, l, P, max_kv, n_short));
    if (h.trace) HIP_TRY(hipMemcpyAsync(h.trace + (size_t)l * h.cap * c.dim, h.X, (size_t)P * c.dim * 4, hipMemcpyDeviceToDevice, st));
  }
  return DSK_OK;
}

// the first position the batched path must leave to the loop (the per-token decode step changes its float association there)
static int hyd_position_limit(const dsk_model* m) {
  // positions before the ring wraps (src/infer.cpp:1271-1277: from pos >= W on the sink keys are rotated in place, token by token)
  int limit = std::min(m->c.max_seq_len, std::max(1, m->c.rs_original_max_position_embeddings));
  // (MLA: from mla_flash_min_kv cached positions on the decode path scores on the matrix cores - its own association; the batched
  // path runs that very kernel over the chunk's long-context tokens and merges like mla_head_kernel: no limit from that regime)
  // (MHA: from mha_split_min cached positions on decode runs mha_split workgroups per head over pieces of the context and merges
  // un-normalised partials - hyd_attn_kernel walks the same pieces and merges them the same way: no limit from that regime)
  // the attention launches keep one float per cached position in LDS
  limit = std::min(limit, 24 * 1024);
  return limit;
}

extern "C" int dsk_hydrate(dsk_model* m, const int32_t* tokens, int n_tokens, int pos0, int mode, float* host_logits) {
  if (!m || !tokens) DSK_FAIL(DSK_ERR_INVALID, "hydrate: null argument");
  if (!m->finalized) DSK_FAIL(DSK_ERR_STATE, "hydrate before finalize");
  if (n_tokens < 1 || pos0 < 0) DSK_FAIL(DSK_ERR_INVALID, "hydrate: %d tokens at pos %d", n_tokens, pos0);
  if (mode != DSK_MODE_HYDRATE_KV_CACHE && mode != DSK_MODE_OUTPUT_LOGITS) DSK_FAIL(DSK_ERR_INVALID, "hydrate: bad mode %d", mode);
  if (mode == DSK_MODE_OUTPUT_LOGITS && !host_logits) DSK_FAIL(DSK_ERR_INVALID, "hydrate: OUTPUT_LOGITS needs a logits buffer");
  for (int i = 0; i < n_tokens; ++i)
    if (tokens[i] < 0 || tokens[i] >= m->c.vocab_size) DSK_FAIL(DSK_ERR_INVALID, "hydrate: token %d out of range", tokens[i]);
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  int done = 0;
  bool last_batched = false;
  const char* why = m->hydrate_batched ? hyd_why_not(m) : "option hydrate_batched is off";
  if (!why && hyd_ensure(m) != DSK_OK) {  // no room for the chunk buffers: the call is still valid - it IS the loop
    (void)hipGetLastError();
    dsk_clear_error();
    why = "the chunk buffers (and tile copies) do not fit the device's free memory";
  }
  m->hydrate_why = why;
  if (!why) {
    const int limit = hyd_position_limit(m);
    int last_P = 0;
    while (done < n_tokens) {
      const int P = std::min(std::min(m->hyd->cap, n_tokens - done), limit - (pos0 + done));
      if (P < 1) break;
      DSK_TRY(hyd_chunk(m, tokens + done, P, pos0 + done));
      HIP_TRY(hipGetLastError());  // a launch the runtime refused surfaces here, before the next chunk builds on its cache rows
      done += P;
      last_P = P;
      m->hyd->last_P = P;
      m->hydrate_batched_tokens += P;
      last_batched = done == n_tokens;
    }
    if (last_batched && mode == DSK_MODE_OUTPUT_LOGITS) {  // final norm + classifier of the last token only (src/infer.cpp:1292-1316)
      HIP_TRY(hipMemcpyAsync(m->x, m->hyd->X + (size_t)(last_P - 1) * m->c.dim, (size_t)m->c.dim * 4, hipMemcpyDeviceToDevice, st));
      DSK_TRY(forward_head(m));
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (last_batched && mode == DSK_MODE_OUTPUT_LOGITS && host_logits != m->logits_host) memcpy(host_logits, m->logits_host, (size_t)m->c.vocab_size * 4);
  }
  for (int i = done; i < n_tokens; ++i) {  // the definition itself: what the batched path must equal
    const bool last = i == n_tokens - 1;
    DSK_TRY(dsk_forward(m, tokens[i], pos0 + i, last ? mode : DSK_MODE_HYDRATE_KV_CACHE, last ? host_logits : nullptr));
    m->hydrate_looped_tokens++;
  }
  return DSK_OK;
}

// the residual stream of token `index` of the LAST batched chunk after block `layer` (dsk_model_set_trace(1) before the first
// dsk_hydrate call): the batched counterpart of dsk_model_get_trace_x
extern "C" int dsk_hydrate_get_trace_x(dsk_model* m, int layer, int index, float* x_out) {
  if (!m || !m->hyd || !m->hyd->trace || !x_out) DSK_FAIL(DSK_ERR_STATE, "hydrate_get_trace_x: no batched trace (dsk_model_set_trace(1) before the first dsk_hydrate)");
  if (layer < 0 || layer >= m->c.n_layers || index < 0 || index >= m->hyd->last_P)
    DSK_FAIL(DSK_ERR_INVALID, "hydrate_get_trace_x: layer %d index %d (the last batched chunk held %d tokens)", layer, index, m->hyd->last_P);
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpy(x_out, m->hyd->trace + ((size_t)layer * m->hyd->cap + index) * m->c.dim, (size_t)m->c.dim * 4, hipMemcpyDeviceToHost));
  return DSK_OK;
}

extern "C" const char* dsk_hydrate_why_not(dsk_model* m) {
  if (!m || !m->finalized) return "model not finalized";
  const char* why = m->hydrate_batched ? hyd_why_not(m) : "option hydrate_batched is off";
  return why ? why : "";
}

// Parity harness: rows [row0, row0 + rows) of a named intermediate of the LAST batched chunk.
//   With option "hydrate_tap_layer" = l the names are block l's stages, copied aside while the chunk ran (hyd_tap above; the
//   names tests/teacher.py uses for dsk_model_get_stage): "q8.x_attn.qs" / ".d", "q_a", "kv_a", "q8.q_a.*", "q8.kv_a.*" (MHA),
//   "q", "kv_b", "att_out" (MHA), "q_rope", "q_c", "latent_out", "q8.latent.*", "vb_out" (MLA), "q8.att.*", "x_mid",
//   "q8.x_ffn.*", "hb", "q8.hb.*", and for MoE blocks "router_logits", "route_e", "route_w", "eout", "hb_sh", "q8.hb_sh.*",
//   "eout_sh".  A row is one token (for "hb" / "eout": the token's K slots).
//   Without a tap: "x" (the residual stream after the last block) and "route_e" of the last MoE block.
// `bytes` must be rows x the row size of that name.
extern "C" int dsk_hydrate_get_buffer(dsk_model* m, const char* name, int row0, int rows, void* out, size_t bytes) {
  if (!m || !m->hyd || !name || !out) DSK_FAIL(DSK_ERR_STATE, "hydrate_get_buffer: no batched chunk has run");
  const HydState& h = *m->hyd;
  const void* src = nullptr;
  size_t row_bytes = 0;
  const auto it = h.taps.find(name);
  if (it != h.taps.end()) { src = it->second.first; row_bytes = it->second.second; }
  else if (!strcmp(name, "x")) { src = h.X; row_bytes = (size_t)m->c.dim * 4; }
  else if (!strcmp(name, "route_e") && m->c.n_active_routed > 0) { src = h.route_e; row_bytes = (size_t)m->c.n_active_routed * 4; }
  if (!src) DSK_FAIL(DSK_ERR_INVALID, "hydrate_get_buffer: no buffer '%s' (option hydrate_tap_layer = %d)", name, m->hydrate_tap_layer);
  if (row0 < 0 || rows < 1 || row0 + rows > h.last_P)
    DSK_FAIL(DSK_ERR_INVALID, "hydrate_get_buffer: rows [%d, %d) of '%s': the last batched chunk held %d tokens", row0, row0 + rows, name, h.last_P);
  if (bytes != (size_t)rows * row_bytes) DSK_FAIL(DSK_ERR_INVALID, "hydrate_get_buffer: '%s' has %zu bytes per row, %zu bytes given for %d rows", name, row_bytes, bytes, rows);
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  HIP_TRY(hipMemcpy(out, static_cast<const uint8_t*>(src) + (size_t)row0 * row_bytes, bytes, hipMemcpyDeviceToHost));
  return DSK_OK;
}
