// forward.cpp -- the token step: Model::_forward_cpu / Block::_block_cpu (src/infer.cpp:1265-1317,
// 810-932) as a short chain of fused HIP launches on one stream, captured into a hipGraph.
//
// Per block (single GPU; K-quants shown, the F8/F16/F32 path has the same shape):
//   1. gemv  wq_a || wkv_a            prologue: rmsnorm(x, attn_norm) -> Q8_K        (infer.cpp:823,943,954)
//   2. gemv  wq_b || wkv_b            prologue: rmsnorm(q_a) / rmsnorm(kv_a[:lora]) -> Q8_K   (:946,974,950,977)
//   3. rope + KV-cache write + sinks                                                  (:956-1020)
//   4. attention (per head)                                                           (:1022-1045)
//   5. Q8_K of the attention output
//   6. gemv  wo, epilogue x += .                                                      (:1048, 832-834)
//   7. router GEMV + moe_gate         prologue: rmsnorm(x, ffn_norm); gate in the last workgroup (:839,847-852)
//   8. gemv  experts w1/w3 (k slots) || shared w1/w3, SiLU-GLU epilogue               (:857-872, 882-897)
//   9. gemv  experts w2 + shared w2, accumulate x += w_k * . in k order, then shared  (:873-878, 899-903)
// Routing decisions never leave HBM (the reference reads them on the host, :854).
#include "engine.h"

#include <math.h>
#include <string.h>
#include <algorithm>

static int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------
// per-kernel-class timing (dsk_profile_forward)
// ---------------------------------------------------------------------------------
struct Prof {
  dsk_model* m;
  int idx = -1;
  bool skip = false;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
static int prof_begin(dsk_model* m, const char* name, double bytes, Prof* p, bool record_start = true) {
  p->m = m;
  if (m->class_filter) {  // dsk_time_kernel_class: only the launches of one class are enqueued
    // the filter may name several classes joined by '+' (interleaved in their in-model order)
    p->skip = true;
    for (const char* f = m->class_filter; f && *f;) {
      const char* e = strchr(f, '+');
      const size_t len = e ? (size_t)(e - f) : strlen(f);
      if (strlen(name) == len && strncmp(name, f, len) == 0) p->skip = false;
      f = e ? e + 1 : nullptr;
    }
    if (!p->skip) { m->class_launches++; m->class_bytes += bytes; }
    return DSK_OK;
  }
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
  if (record_start) HIP_TRY(hipEventRecord(p->e0, m->ctx->stream));
  return DSK_OK;
}
static int prof_end(Prof* p) {
  if (!p->m->profiling || p->m->class_filter) return DSK_OK;
  HIP_TRY(hipEventRecord(p->e1, p->m->ctx->stream));
  p->m->ktimes[p->idx].ev.push_back({p->e0, p->e1});
  return DSK_OK;
}
#define PROFILED(name, bytes, call)                     \
  do {                                                  \
    Prof _p;                                            \
    DSK_TRY(prof_begin(m, name, (double)(bytes), &_p)); \
    if (!_p.skip) DSK_TRY(call);                        \
    DSK_TRY(prof_end(&_p));                             \
  } while (0)

// ---------------------------------------------------------------------------------
// launch descriptors
// ---------------------------------------------------------------------------------
static double weight_bytes_2d(const dsk_model* m, int quant, int rows, int n) {
  double b = (double)mat_bytes(quant, rows, n);
  if (quant == DSK_QUANT_F8E5M2) b += 4.0 * cdiv(rows, m->c.block_size[0]) * cdiv(n, m->c.block_size[1]);
  return b;
}
// activation + output bytes of one GEMV (SURVEY 8d per-GEMV unit: (n/256)*292 + 4*d for K-quants)
static double io_bytes(int quant, int n, int rows) { return (is_kq(quant) ? (double)n / 256 * 292 : (double)n * 4) + 4.0 * rows; }

static void task_weights(GemvTask& T, const DTensor& t) {
  T.qs = t.qs; T.sc = t.sc; T.hm = t.hm; T.dm = t.dm; T.scale = t.scale;
  T.rows = t.rows; T.n = t.n;
  T.local_experts = 1;
  T.wt_tiled = t.tiled;
}
static void task_weights2(GemvTask& T, const DTensor& t3) {
  T.qs2 = t3.qs; T.sc2 = t3.sc; T.hm2 = t3.hm; T.dm2 = t3.dm; T.scale2 = t3.scale;
}
static void task_expert(GemvTask& T, const DTensor& t, const int* ids, int slot) {
  T.e_qs = t.e_qs; T.e_sc = t.e_sc; T.e_hm = t.e_hm; T.e_dm = t.e_dm; T.e_scale = t.e_scale;
  T.expert_ids = ids; T.slot = slot; T.expert_base = t.expert_base; T.local_experts = t.local_experts;
}
static void task_act_norm(GemvTask& T, const float* x, const DTensor& norm, float eps) {
  T.act_mode = ACT_F32_NORM; T.a_f32 = x; T.norm_w = reinterpret_cast<const float*>(norm.qs); T.eps = eps;
}
static void task_act_f32(GemvTask& T, const float* x) { T.act_mode = ACT_F32; T.a_f32 = x; }
// hidden vector of a GLU launch; its consumer quantises it in the prologue.  (Recording per-block maxima
// in the producer's epilogue with atomics was measured: +7 us on experts_w13, nothing gained in W2.)
static void task_act_hb(const dsk_model* m, GemvTask& T, int layer, size_t off) {
  (void)layer;
  T.a_f32 = m->hb + off;
  T.act_mode = ACT_F32;
}
static void task_out_hb(const dsk_model* m, GemvTask& T, int layer, size_t off) {
  (void)layer;
  T.out = m->hb + off;
}
static void task_act_q8(GemvTask& T, const Q8Buf& q) { T.act_mode = ACT_Q8; T.a_qs = q.qs; T.a_d = q.d; T.a_bsums = q.bsums; }

static int add_plan(dsk_model* m, GemvLaunch& h, int* idx_out) {
  h.tiled = h.t[0].wt_tiled;  // every tensor of a launch has the same layout (engine.cpp role_tiled deals whole launches)
  for (int i = 1; i < h.n_tasks; ++i)
    if (h.t[i].wt_tiled != h.t[0].wt_tiled) DSK_FAIL(DSK_ERR_STATE, "launch plan: tasks with mixed weight layouts");
  h.b0 = std::max(1, m->c.block_size[0]);
  h.b1 = std::max(1, m->c.block_size[1]);
  h.act = m->c.act;
  DSK_TRY(gemv_plan(h, m->target_wgs));
  *idx_out = (int)m->plans.size();
  m->plans.push_back(h);
  return DSK_OK;
}

int build_plans(dsk_model* m) {
  const dsk_config& c = m->c;
  const int nl = c.n_layers, H = c.n_heads, K = c.n_active_routed, mi = c.moe_intermediate_size;
  const int wq = c.weight_quant;
  const bool kq = is_kq(wq);
  const int shared_n = c.n_shared_experts * mi;
  const int hb_stride = std::max(std::max(mi, shared_n), 1);
  m->lp_qkv_a.assign(nl, -1); m->lp_qkv_b.assign(nl, -1); m->lp_wv_b.assign(nl, -1); m->lp_wo.assign(nl, -1);
  m->lp_w13.assign(nl, -1); m->lp_w2.assign(nl, -1); m->lp_sh13.assign(nl, -1);
  m->plans.clear();
  m->head_attn.assign(nl, HeadAttnArgs());
  m->mla_head.assign(nl, MlaHeadArgs());
  {
    MlaFlashArgs z;
    memset(&z, 0, sizeof z);
    m->mla_flash.assign(nl, z);
  }
  m->head_attn_bytes.assign(nl, 0.0);
  {
    MoeFfnArgs z;
    memset(&z, 0, sizeof z);
    m->moe_ffn.assign(nl, z);
  }
  for (int l = 0; l < nl; ++l) {
    Layer& L = m->L[l];
    {  // 1. wq_a (or wq) || wkv_a on rmsnorm(x, attn_norm)
      GemvLaunch h;
      memset(&h, 0, sizeof h);
      h.quant = wq;
      GemvTask& a = h.t[h.n_tasks++];
      const DTensor& tq = c.q_lora_rank > 0 ? L.t[DSK_ROLE_WQ_A] : L.t[DSK_ROLE_WQ];
      task_weights(a, tq);
      task_act_norm(a, m->x, L.t[DSK_ROLE_ATTN_NORM], c.norm_eps);
      a.out = c.q_lora_rank > 0 ? m->q_a : m->q;
      GemvTask& b = h.t[h.n_tasks++];
      task_weights(b, L.t[DSK_ROLE_WKV_A]);
      task_act_norm(b, m->x, L.t[DSK_ROLE_ATTN_NORM], c.norm_eps);
      b.out = m->kv_a;
      h.algo_bytes = weight_bytes_2d(m, wq, a.rows, a.n) + weight_bytes_2d(m, wq, b.rows, b.n) + c.dim * 8.0 + 4.0 * (a.rows + b.rows);
      h.timeline = m->timeline_of(0);
      h.ahead = m->gemv_ahead & 1;  // option "gemv_ahead": the workgroup's weights requested ahead of the staging of x (kernels_gemv.hip gemv_ahead_kernel)
      DSK_TRY(add_plan(m, h, &m->lp_qkv_a[l]));
    }
    {  // 2. second-stage projections on the normed latents
      GemvLaunch h;
      memset(&h, 0, sizeof h);
      h.quant = wq;
      double bytes = 0;
      auto add = [&](const DTensor& t, const float* x, const DTensor& norm, float* out) {
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, t);
        task_act_norm(T, x, norm, c.norm_eps);
        T.out = out;
        bytes += weight_bytes_2d(m, wq, t.rows, t.n) + t.n * 8.0 + 4.0 * t.rows;
      };
      if (c.use_mla) {
        add(L.t[DSK_ROLE_WQ_ROPE_B], m->q_a, L.t[DSK_ROLE_Q_A_NORM], m->q_rope);
        add(L.t[DSK_ROLE_WC], m->q_a, L.t[DSK_ROLE_Q_A_NORM], m->q_c);
      } else {
        if (c.q_lora_rank > 0) add(L.t[DSK_ROLE_WQ_B], m->q_a, L.t[DSK_ROLE_Q_A_NORM], m->q);
        add(L.t[DSK_ROLE_WKV_B], m->kv_a, L.t[DSK_ROLE_KV_A_NORM], m->kv_b);
      }
      h.algo_bytes = bytes;
      if (c.use_mla && m->ride_kvwrite) h.reserve_wgs = 1;  // the latent's cache write rides as one more workgroup of this launch
      if (c.use_mla) h.ahead = m->gemv_ahead & 1;            // (gemv_kvwrite_ahead_kernel: the first row group ahead of the staging of q_a)
      DSK_TRY(add_plan(m, h, &m->lp_qkv_b[l]));
      if (!c.use_mla) {  // MHA: the same projections, consumed per head by the fused attention launch
        HeadAttnArgs A;
        memset(&A, 0, sizeof A);
        A.quant = wq; A.b0 = std::max(1, c.block_size[0]); A.b1 = std::max(1, c.block_size[1]);
        A.has_q = c.q_lora_rank > 0;
        A.timeline = m->timeline_of(1);
        const GemvLaunch& P = m->plans[m->lp_qkv_b[l]];
        if (A.has_q) { A.tq = P.t[0]; A.tkv = P.t[1]; }
        else A.tkv = P.t[0];
        A.tiled = P.tiled;
        AttnMhaArgs& a = A.a;
        a.q = m->q; a.kv_b = m->kv_b; a.kv_a = m->kv_a; a.key_cache = L.key_cache; a.value_cache = L.value_cache; a.out = m->att_out;
        a.n_heads = H; a.head_dim = m->head_dim; a.nope = c.qk_nope_head_dim; a.rope = c.qk_rope_head_dim; a.v_dim = c.v_head_dim;
        a.lora = c.kv_lora_rank; a.is_v3 = c.has_moegate_bias;
        a.q_counter = m->att_counter;
        A.split_part = m->mha_split_part; A.split_counter = m->mha_split_counter;
        if (kq && !m->att_q8_in_wo) { a.q_qs = m->a_att.qs; a.q_d = m->a_att.d; a.q_bsums = m->a_att.bsums; }
        DSK_TRY(head_attn_plan(A));
        A.a.out = m->att_out;
        m->head_attn[l] = A;
        m->head_attn_bytes[l] = bytes;
      }
    }
    if (c.use_mla) {  // per-head wv_b on the per-head latent outputs (src/infer.cpp:1134-1137): block-diagonal
      GemvLaunch h;
      memset(&h, 0, sizeof h);
      h.quant = wq;
      h.bd_heads = H;
      GemvTask& T = h.t[h.n_tasks++];
      task_weights(T, L.t[DSK_ROLE_WV_B]);
      T.rows = c.v_head_dim; T.n = c.kv_lora_rank;
      task_act_f32(T, m->att_out);
      T.out = m->vb_out;
      h.algo_bytes = weight_bytes_2d(m, wq, H * c.v_head_dim, c.kv_lora_rank) + 4.0 * H * (c.kv_lora_rank + c.v_head_dim);
      DSK_TRY(add_plan(m, h, &m->lp_wv_b[l]));
      MlaHeadArgs A;  // the fused per-head launch: attention over the latent cache + this head's wv_b rows + Q8_K
      memset(&A, 0, sizeof A);
      A.quant = wq; A.b0 = std::max(1, c.block_size[0]); A.b1 = std::max(1, c.block_size[1]);
      A.twv = m->plans[m->lp_wv_b[l]].t[0];
      A.tiled = m->plans[m->lp_wv_b[l]].tiled;
      A.timeline = m->timeline_of(1);
      A.a.q_rope = m->q_rope; A.a.q_c = m->q_c; A.a.kv_a = m->kv_a; A.a.nope_cache = L.nope_cache; A.a.rope_cache = L.rope_cache;
      A.a.out = m->att_out; A.a.n_heads = H; A.a.head_dim = m->head_dim; A.a.rope = c.qk_rope_head_dim; A.a.lora = c.kv_lora_rank;
      A.a.is_v3 = c.has_moegate_bias;
      A.fin.out = m->vb_out; A.fin.v_dim = c.v_head_dim; A.fin.n_heads = H; A.fin.q_counter = m->att_counter;
      if (kq && !m->att_q8_in_wo) { A.fin.q_qs = m->a_att.qs; A.fin.q_d = m->a_att.d; A.fin.q_bsums = m->a_att.bsums; }
      if (m->fl_part_o) {  // kv_len >= MLA_FLASH_MIN_KV: scores / values of all heads on the matrix cores, merged per head here
        const int max_kv = std::min(c.max_seq_len, std::max(1, c.rs_original_max_position_embeddings));
        MlaFlashArgs F;
        memset(&F, 0, sizeof F);
        F.q_c = m->q_c; F.q_rope = m->q_rope; F.rotate_q = 1; F.nope_cache = L.nope_cache; F.rope_cache = L.rope_cache;
        F.part_o = m->fl_part_o; F.part_ml = m->fl_part_ml; F.n_heads = H; F.head_dim = m->head_dim; F.lora = c.kv_lora_rank;
        F.rope = c.qk_rope_head_dim; F.is_v3 = c.has_moegate_bias; F.timeline = m->timeline_of(6); F.n_chunks = 64; F.chunk_len = 0;  // per step: MLA_FL_CHUNK(kv_len, 64)
        m->mla_flash[l] = F;
        A.flash_thresh = m->mla_flash_min_kv; A.fl_chunk_len = F.chunk_len; A.fl_n_chunks = F.n_chunks; A.fl_part_o = F.part_o; A.fl_part_ml = F.part_ml;
      }
      DSK_TRY(mla_head_plan(A));
      m->mla_head[l] = A;
      m->head_attn_bytes[l] = h.algo_bytes;
    }
    {  // 6. wo, x += .
      GemvLaunch h;
      memset(&h, 0, sizeof h);
      h.quant = wq;
      GemvTask& T = h.t[h.n_tasks++];
      task_weights(T, L.t[DSK_ROLE_WO]);
      const float* src = c.use_mla ? m->vb_out : m->att_out;
      if (kq && !m->att_q8_in_wo) task_act_q8(T, m->a_att);
      else task_act_f32(T, src);
      T.out = m->x; T.epilogue = EPI_ADD;
      h.algo_bytes = weight_bytes_2d(m, wq, T.rows, T.n) + io_bytes(wq, T.n, T.rows) + 4.0 * T.rows;
      h.timeline = m->timeline_of(2);
      h.ahead = m->gemv_ahead & 2;  // (gemv_ahead_q8_kernel: the first row group's first half requested ahead of the vector's copy into LDS)
      DSK_TRY(add_plan(m, h, &m->lp_wo[l]));
    }
    if (!L.is_moe) {
      {  // dense w1/w3 GLU on rmsnorm(x, ffn_norm)
        GemvLaunch h;
        memset(&h, 0, sizeof h);
        h.quant = wq; h.glu = 1;
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, L.t[DSK_ROLE_W1]);
        task_weights2(T, L.t[DSK_ROLE_W3]);
        task_act_norm(T, m->x, L.t[DSK_ROLE_FFN_NORM], c.norm_eps);
        task_out_hb(m, T, l, 0);
        h.algo_bytes = 2 * weight_bytes_2d(m, wq, T.rows, T.n) + c.dim * 8.0 + 4.0 * T.rows;
        DSK_TRY(add_plan(m, h, &m->lp_w13[l]));
      }
      {  // dense w2, x += .
        GemvLaunch h;
        memset(&h, 0, sizeof h);
        h.quant = wq;
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, L.t[DSK_ROLE_W2]);
        task_act_hb(m, T, l, 0);
        T.out = m->x; T.epilogue = EPI_ADD;
        h.algo_bytes = weight_bytes_2d(m, wq, T.rows, T.n) + 4.0 * T.n + 8.0 * T.rows;
        DSK_TRY(add_plan(m, h, &m->lp_w2[l]));
      }
      continue;
    }
    if (K + 1 > GEMV_MAX_TASKS) DSK_FAIL(DSK_ERR_UNSUPPORTED, "n_active_routed %d > %d", K, GEMV_MAX_TASKS - 1);
    const int* ae = m->route_e + (size_t)l * K;
    const float* aw = m->route_w + (size_t)l * K;
    const DTensor &w1 = L.t[DSK_ROLE_W1], &w2 = L.t[DSK_ROLE_W2], &w3 = L.t[DSK_ROLE_W3];
    const double e13 = 2 * weight_bytes_2d(m, wq, mi, c.dim), e2 = weight_bytes_2d(m, wq, c.dim, mi);
    {  // 8. routed w1/w3 (k slots) || shared w1/w3, GLU
      GemvLaunch h;
      memset(&h, 0, sizeof h);
      h.quant = wq; h.glu = 1;
      for (int k = 0; k < K; ++k) {
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, w1);
        task_weights2(T, w3);
        task_expert(T, w1, ae, k);
        if (kq && c.dim % 256 == 0) task_act_q8(T, m->a_xb);  // the router launch leaves Q8_K(rmsnorm(x)) behind
        else task_act_norm(T, m->x, L.t[DSK_ROLE_FFN_NORM], c.norm_eps);
        task_out_hb(m, T, l, (size_t)k * hb_stride);
      }
      double bytes = K * e13 + c.dim * 8.0 + 4.0 * K * mi;
      // 1 GPU, K-quants: the shared expert's w1/w3 rides in the router launch (router_shared_kernel) instead
      bool ride = false;
      if (c.n_shared_experts > 0 && kq && c.dim % 256 == 0 && !m->sharded() && m->ride_shared) {
        GemvLaunch hs;
        memset(&hs, 0, sizeof hs);
        hs.quant = wq; hs.glu = 1; hs.force_NW = 16;
        // 128 quarter-filled workgroups next to the router's 128: at 64 the rider was bound by what ONE CU streams
        // (~24 GB/s: 14.5 MB / 64 CUs = 9 us) and the launch by the rider (13.7 -> 12.5 us)
        hs.fill_div = m->rider_fill;  // option "rider_fill"
        GemvTask& T = hs.t[hs.n_tasks++];
        task_weights(T, L.t[DSK_ROLE_SHARED_W1]);
        task_weights2(T, L.t[DSK_ROLE_SHARED_W3]);
        task_act_norm(T, m->x, L.t[DSK_ROLE_FFN_NORM], c.norm_eps);
        task_out_hb(m, T, l, (size_t)K * hb_stride);
        hs.algo_bytes = 2 * weight_bytes_2d(m, wq, shared_n, c.dim) + 4.0 * shared_n + c.dim * 8.0;
        RouterArgs probe;
        memset(&probe, 0, sizeof probe);
        probe.ksplit = m->router_ksplit; probe.dim = c.dim; probe.norm_w = reinterpret_cast<const float*>(L.t[DSK_ROLE_FFN_NORM].qs);
        hs.tiled = hs.t[0].wt_tiled;
        GemvLaunch trial = hs;
        trial.b0 = std::max(1, c.block_size[0]); trial.b1 = std::max(1, c.block_size[1]); trial.act = c.act;
        // (V2-Lite's 2048-wide rows plan U = 1 and stay out: with the rider forced in - 2 column steps - its router launch
        // went 8.6 -> 11.2 us and the fused expert launch took 22.2 us against 9.0 + 12.4: 642 instead of 682 tok/s)
        if (gemv_plan(trial, m->target_wgs) == DSK_OK && router_shared_supported(probe, trial)) {
          hs.timeline = m->timeline_of(3);
          DSK_TRY(add_plan(m, hs, &m->lp_sh13[l]));
          ride = true;
        }
      }
      if (c.n_shared_experts > 0 && !ride) {
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, L.t[DSK_ROLE_SHARED_W1]);
        task_weights2(T, L.t[DSK_ROLE_SHARED_W3]);
        if (kq && c.dim % 256 == 0) task_act_q8(T, m->a_xb);
        else task_act_norm(T, m->x, L.t[DSK_ROLE_FFN_NORM], c.norm_eps);
        task_out_hb(m, T, l, (size_t)K * hb_stride);
        bytes += 2 * weight_bytes_2d(m, wq, shared_n, c.dim) + 4.0 * shared_n;
      }
      h.algo_bytes = bytes;
      h.compact_absent = m->ctx->world > 1 && m->compact_absent;  // option "compact_absent"
      DSK_TRY(add_plan(m, h, &m->lp_w13[l]));
    }
    {  // 9. per-slot W2 into eout[slot]; 1 GPU: the combine rides in the same launch; sharded: all-reduce first
      GemvLaunch h;
      memset(&h, 0, sizeof h);
      h.quant = wq;
      for (int k = 0; k < K; ++k) {
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, w2);
        task_expert(T, w2, ae, k);
        task_act_hb(m, T, l, (size_t)k * hb_stride);
        T.out = m->eout + (size_t)k * c.dim; T.epilogue = EPI_STORE; T.accum_w = aw + k;
      }
      double bytes = K * (e2 + 4.0 * mi + 4.0 * c.dim);
      if (c.n_shared_experts > 0) {
        GemvTask& T = h.t[h.n_tasks++];
        task_weights(T, L.t[DSK_ROLE_SHARED_W2]);
        task_act_hb(m, T, l, (size_t)K * hb_stride);
        T.out = m->eout + (size_t)K * c.dim; T.epilogue = EPI_STORE;
        bytes += weight_bytes_2d(m, wq, c.dim, shared_n) + 4.0 * shared_n + 4.0 * c.dim;
      }
      if (!m->sharded()) { h.comb_x = m->x; h.comb_counter = m->comb_counter; }
      else { h.comb_geometry = 1; h.zero_absent = 1; }
      h.algo_bytes = bytes;
      DSK_TRY(add_plan(m, h, &m->lp_w2[l]));
    }
    // 8+9 in ONE launch (kernels_moe.hip): K-quants, one GPU, the shared expert's w1/w3 riding in the router launch
    // (or no shared expert).  Same lanes per row as the two plans above => bit-identical results.
    // Float weights (F8E5M2 / F16 / F32): moe_ffn_f_kernel, the shared expert's w1 / w3 computed in the same launch.
    const bool fuse_kq = kq && c.dim % 256 == 0 && (c.n_shared_experts == 0 || m->lp_sh13[l] >= 0);
    const bool fuse_f = !kq && m->fuse_moe_float;
    if ((fuse_kq || fuse_f) && !m->sharded() && m->fuse_moe && K + 1 <= 16) {
      MoeFfnArgs a;
      memset(&a, 0, sizeof a);
      a.quant = wq;
      a.w1_qs = w1.qs; a.w1_sc = w1.sc; a.w1_hm = w1.hm; a.w1_dm = w1.dm;
      a.w3_qs = w3.qs; a.w3_sc = w3.sc; a.w3_hm = w3.hm; a.w3_dm = w3.dm;
      a.w2_qs = w2.qs; a.w2_sc = w2.sc; a.w2_hm = w2.hm; a.w2_dm = w2.dm;
      a.e13_qs = w1.e_qs; a.e13_sc = w1.e_sc; a.e13_hm = w1.e_hm; a.e13_dm = w1.e_dm;
      a.e2_qs = w2.e_qs; a.e2_sc = w2.e_sc; a.e2_hm = w2.e_hm; a.e2_dm = w2.e_dm;
      if (c.n_shared_experts > 0) {
        const DTensor& s2 = L.t[DSK_ROLE_SHARED_W2];
        a.sw2_qs = s2.qs; a.sw2_sc = s2.sc; a.sw2_hm = s2.hm; a.sw2_dm = s2.dm;
        a.shared_n = shared_n;
      }
      a.route_e = ae; a.route_w = aw;
      a.K = K; a.mi = mi; a.dim = c.dim; a.act = c.act;
      a.a_qs = m->a_xb.qs; a.a_d = m->a_xb.d; a.a_bsums = m->a_xb.bsums;
      a.hb = m->hb; a.hb_stride = hb_stride; a.eout = m->eout; a.x = m->x;
      a.slot_ctr = m->moe_ctr;
      if (m->moe_q8_handoff && K * (mi / 256) <= MOE_BLK_CTRS) {  // hidden vectors handed over as Q8_K (kernels_moe.hip)
        a.hq_qs = m->a_hb.qs; a.hq_d = m->a_hb.d; a.hq_bsums = m->a_hb.bsums; a.blk_ctr = m->moe_blk_ctr;
      }
      if (!kq) {
        const DTensor &s1 = L.t[DSK_ROLE_SHARED_W1], &s3 = L.t[DSK_ROLE_SHARED_W3], &s2 = L.t[DSK_ROLE_SHARED_W2];
        a.w1_scale = w1.scale; a.w3_scale = w3.scale; a.w2_scale = w2.scale;
        a.e13_scale = w1.e_scale; a.e2_scale = w2.e_scale;
        if (c.n_shared_experts > 0) { a.sw1_qs = s1.qs; a.sw3_qs = s3.qs; a.sw1_scale = s1.scale; a.sw3_scale = s3.scale; a.sw2_scale = s2.scale; }
        a.norm_w = reinterpret_cast<const float*>(L.t[DSK_ROLE_FFN_NORM].qs);
        a.eps = c.norm_eps;
        a.b0 = std::max(1, c.block_size[0]); a.b1 = std::max(1, c.block_size[1]);
        a.hq_qs = nullptr;
      }
      a.n_experts = c.n_routed_experts;
      a.tiled = w1.tiled;
      a.err = m->err_host;
      a.spin_limit = m->moe_spin_limit;
      a.pipe = m->moe_pipe;
      a.cand = m->moe_cand;
      a.timeline = m->timeline_of(4);
      a.lprA_log2 = m->plans[m->lp_w13[l]].lpr_log2;
      a.lprB_log2 = m->plans[m->lp_w2[l]].lpr_log2;
      a.algo_bytes = m->plans[m->lp_w13[l]].algo_bytes + m->plans[m->lp_w2[l]].algo_bytes;
      if (moe_ffn_plan(a, m->ctx->n_cus) == DSK_OK) m->moe_ffn[l] = a;
      else dsk_clear_error();  // "not fusable" is not an error: the two-launch form runs (dsk_model_get_info "fused_moe_layers" tells)
    }
  }
  {  // classifier on rmsnorm(x, final_norm) (src/infer.cpp:1292-1316)
    GemvLaunch h;
    memset(&h, 0, sizeof h);
    h.quant = wq;
    GemvTask& T = h.t[h.n_tasks++];
    const DTensor& cls = m->tied ? m->g[DSK_ROLE_EMBED] : m->g[DSK_ROLE_OUTPUT];
    task_weights(T, cls);
    task_act_norm(T, m->x, m->g[DSK_ROLE_FINAL_NORM], c.norm_eps);
    T.out = m->logits;
    h.algo_bytes = weight_bytes_2d(m, wq, T.rows, T.n) + c.dim * 8.0 + 4.0 * T.rows;
    DSK_TRY(add_plan(m, h, &m->lp_head));
  }
  (void)H;
  HIP_TRY(hipMalloc((void**)&m->plans_dev, m->plans.size() * sizeof(GemvLaunch)));
  if (m->tail_prefetch > 0) {  // (expert-sharded ranks too: the attention launches are replicated; the fused expert launch is not planned there)
    // Cold lines: every launch opens with lines nobody has touched since the previous token - its descriptor, its norm
    // weights - and waits ~1-2 us for them from HBM.  Workgroups that cost nothing read them into every XCD's L2 ahead of
    // time (one per XCD: workgroup b runs on XCD b % 8): behind the heads of the per-head attention launch, on the 128 CUs it
    // leaves idle, what wo and the FFN's first launch open with; behind the fused expert launch, as its first workgroups
    // leave, what the next block's first launch opens with.  Token 5.06-5.11 -> 4.94-4.96 ms on one box, MLA 5.53-5.55 -> 5.40-5.44
    // (first-stage projections 7.0 -> 6.7 us, wo 11.75 -> 10.9, the gate's bias row; the attention launch itself +0.6 with its 8
    // extra workgroups).  More lines behind the expert launch (the
    // next block's latent norm weights) lengthen its tail by 0.5 us and lose.
    auto f32v = [&](const DTensor& t, int n, const void** p, int* bytes) { if (t.bound()) { *p = t.qs; *bytes = n * 4; } };
    for (int l = 0; l < nl; ++l) {
      if (!c.use_mla && m->lp_wo[l] >= 0 && m->head_attn[l].tkv.n > 0) {
        HeadAttnArgs& A = m->head_attn[l];
        const int nxt = m->L[l].is_moe ? m->lp_sh13[l] : m->lp_w13[l];
        A.pf_wgs = m->tail_prefetch;
        A.pf_p[0] = m->plans_dev + m->lp_wo[l]; A.pf_n[0] = (int)sizeof(GemvLaunch);
        f32v(m->L[l].t[DSK_ROLE_FFN_NORM], c.dim, &A.pf_p[1], &A.pf_n[1]);
        if (nxt >= 0) { A.pf_p[2] = m->plans_dev + nxt; A.pf_n[2] = (int)sizeof(GemvLaunch); }
        if (m->L[l].is_moe) f32v(m->L[l].t[DSK_ROLE_MOEGATE_BIAS], c.n_routed_experts, &A.pf_p[3], &A.pf_n[3]);  // the gate workgroup's
        else if (m->lp_w2[l] >= 0) { A.pf_p[3] = m->plans_dev + m->lp_w2[l]; A.pf_n[3] = (int)sizeof(GemvLaunch); }  // dense block: w2's descriptor
        // (the step parameters - rope table - as a further range: no gain)
      }
      if (c.use_mla && m->lp_wo[l] >= 0 && m->mla_head[l].a.n_heads > 0) {
        MlaHeadArgs& A = m->mla_head[l];
        const int nxt = m->L[l].is_moe ? m->lp_sh13[l] : m->lp_w13[l];
        A.pf_wgs = m->tail_prefetch;
        A.pf_p[0] = m->plans_dev + m->lp_wo[l]; A.pf_n[0] = (int)sizeof(GemvLaunch);
        f32v(m->L[l].t[DSK_ROLE_FFN_NORM], c.dim, &A.pf_p[1], &A.pf_n[1]);
        if (nxt >= 0) { A.pf_p[2] = m->plans_dev + nxt; A.pf_n[2] = (int)sizeof(GemvLaunch); }
        if (m->L[l].is_moe) f32v(m->L[l].t[DSK_ROLE_MOEGATE_BIAS], c.n_routed_experts, &A.pf_p[3], &A.pf_n[3]);
      }
      if (m->moe_ffn[l].grid <= 0) continue;
      MoeFfnArgs& a = m->moe_ffn[l];
      const int nplan = l + 1 < nl ? m->lp_qkv_a[l + 1] : m->lp_head;
      a.pf_wgs = m->tail_prefetch;
      f32v(l + 1 < nl ? m->L[l + 1].t[DSK_ROLE_ATTN_NORM] : m->g[DSK_ROLE_FINAL_NORM], c.dim, &a.pf_p[0], &a.pf_n[0]);
      if (nplan >= 0) { a.pf_p[1] = m->plans_dev + nplan; a.pf_n[1] = (int)sizeof(GemvLaunch); }
    }
  }
  HIP_TRY(hipMemcpy(m->plans_dev, m->plans.data(), m->plans.size() * sizeof(GemvLaunch), hipMemcpyHostToDevice));
  m->scratch_bytes += (double)m->plans.size() * sizeof(GemvLaunch);
  return DSK_OK;
}

void free_plans(dsk_model* m) {
  if (m->plans_dev) hipFree(m->plans_dev);
  m->plans_dev = nullptr;
}

static int run_plan(dsk_model* m, const char* name, int idx) {
  if (idx < 0) DSK_FAIL(DSK_ERR_STATE, "missing launch plan for %s", name);
  if (m->profiling && !m->class_filter) {
    // the kernel's own dispatch timestamps (hipExtLaunchKernel): in-situ duration, as rocprofv3 reports it
    Prof p;
    DSK_TRY(prof_begin(m, name, m->plans[idx].algo_bytes, &p, false));
    g_prof_start = p.e0; g_prof_stop = p.e1;
    int r = gemv_launch(m->ctx->stream, m->plans_dev + idx, m->plans[idx]);
    g_prof_start = g_prof_stop = nullptr;
    DSK_TRY(r);
    m->ktimes[p.idx].ev.push_back({p.e0, p.e1});
    return DSK_OK;
  }
  PROFILED(name, m->plans[idx].algo_bytes, gemv_launch(m->ctx->stream, m->plans_dev + idx, m->plans[idx]));
  return DSK_OK;
}

// ---------------------------------------------------------------------------------
// one token
// ---------------------------------------------------------------------------------
int fill_step_params_at(dsk_model* m, int token, int pos, StepParams* sp) {
  const dsk_config& c = m->c;
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
static int fill_step_params(dsk_model* m, int token, int pos) { return fill_step_params_at(m, token, pos, m->sp_host); }

static int attention_mha(dsk_model* m, int l, int max_kv) {
  const dsk_config& c = m->c;
  hipStream_t st = m->ctx->stream;
  const int H = c.n_heads, hd = m->head_dim;
  DSK_TRY(run_plan(m, "gemv_qkv_a", m->lp_qkv_a[l]));
  // second-stage projections (wq_b, wkv_b) + rope + cache write + attention + Q8_K of the head outputs:
  // one launch, one workgroup per head (kernels_gemv.hip head_attn_kernel)
  HeadAttnArgs HA = m->head_attn[l];
  if (m->stage_layer == l && m->tap_qs) {  // parity taps of this block (dsk_model_run_block)
    HA.tap_qs = m->tap_qs + m->tap_off_qa;
    HA.tap_d = m->tap_d + m->tap_off_qa / 256;
    HA.tap_stride = (int)(m->tap_off_kva - m->tap_off_qa);
  }
  PROFILED("attn_mha", m->head_attn_bytes[l] + (double)m->sp_host->kv_len * H * (hd + c.v_head_dim) * 2 + (double)H * (hd * 2 + c.v_head_dim * 11),
           launch_head_attn(st, HA, m->sp_dev, max_kv,
                            m->mha_split > 1 && m->sp_host->kv_len >= m->mha_split_min ? m->mha_split : 1));
  DSK_TRY(run_plan(m, "gemv_wo", m->lp_wo[l]));  // residual: src/infer.cpp:832-834
  return DSK_OK;
}

static int attention_mla(dsk_model* m, int l, int max_kv) {
  const dsk_config& c = m->c;
  Layer& L = m->L[l];
  hipStream_t st = m->ctx->stream;
  const int H = c.n_heads;
  DSK_TRY(run_plan(m, "gemv_qkv_a", m->lp_qkv_a[l]));
  {  // second-stage projections; latent norm + this position's cache entries + sink rotation (src/infer.cpp:1089-1110):
     // one small workgroup, riding as the last workgroup of the projection launch when the plan allows
    MlaKvArgs kv;
    kv.kv_a = m->kv_a; kv.norm_w = reinterpret_cast<const float*>(L.t[DSK_ROLE_KV_A_NORM].qs); kv.eps = c.norm_eps;
    kv.nope_cache = L.nope_cache; kv.rope_cache = L.rope_cache; kv.lora = c.kv_lora_rank; kv.rope = c.qk_rope_head_dim;
    kv.is_v3 = c.has_moegate_bias;
    const GemvLaunch& hb = m->plans[m->lp_qkv_b[l]];
    if (gemv_kvwrite_supported(hb) && !m->profiling && m->ride_kvwrite) {
      PROFILED("gemv_qkv_b", hb.algo_bytes + (double)c.kv_lora_rank * 14 + c.qk_rope_head_dim * 6,
               launch_gemv_kvwrite(st, m->plans_dev + m->lp_qkv_b[l], hb, kv, m->sp_dev));
    } else {
      DSK_TRY(run_plan(m, "gemv_qkv_b", m->lp_qkv_b[l]));
      PROFILED("rope_kv", (double)c.kv_lora_rank * 14 + c.qk_rope_head_dim * 6, launch_mla_kv_write(st, kv, m->sp_dev));
    }
  }
  if (m->mla_flash[l].part_o && m->sp_host->kv_len >= m->mla_flash_min_kv)  // long-context regime (its own graph: dsk_forward)
    PROFILED("attn_mla_flash", (double)m->sp_host->kv_len * (c.kv_lora_rank + c.qk_rope_head_dim) * 2, launch_mla_flash(st, m->mla_flash[l], m->sp_dev, 0));
  // q rope + attention over the shared latent cache + per-head wv_b + Q8_K of the outputs: one launch
  MlaHeadArgs MA = m->mla_head[l];
  if (m->stage_layer == l && m->tap_qs) {  // parity taps of this block (dsk_model_run_block)
    MA.tap_qs = m->tap_qs + m->tap_off_latent;
    MA.tap_d = m->tap_d + m->tap_off_latent / 256;
    MA.tap_o = m->tap_latent;
  }
  PROFILED("attn_mla", m->head_attn_bytes[l] + (double)m->sp_host->kv_len * (c.kv_lora_rank + c.qk_rope_head_dim) * 2 + (double)H * c.v_head_dim * 9,
           launch_mla_head(st, MA, m->sp_dev, max_kv));
  DSK_TRY(run_plan(m, "gemv_wo", m->lp_wo[l]));
  return DSK_OK;
}

// MoE / dense FFN of one block (src/infer.cpp:844-931)
static int ffn(dsk_model* m, int l) {
  const dsk_config& c = m->c;
  Layer& L = m->L[l];
  hipStream_t st = m->ctx->stream;
  if (!L.is_moe) {
    DSK_TRY(run_plan(m, "gemv_dense_w13", m->lp_w13[l]));
    DSK_TRY(run_plan(m, "gemv_dense_w2", m->lp_w2[l]));
    return DSK_OK;
  }
  const int K = c.n_active_routed, E = c.n_routed_experts;
  RouterArgs r;
  memset(&r, 0, sizeof r);
  r.w = reinterpret_cast<const float*>(L.t[DSK_ROLE_MOEGATE].qs);
  r.x = m->x;
  r.norm_w = reinterpret_cast<const float*>(L.t[DSK_ROLE_FFN_NORM].qs);
  r.eps = c.norm_eps;
  r.timeline = m->timeline_of(5);
  r.n_routed = E; r.dim = c.dim; r.ksplit = m->router_ksplit;
  r.partial = m->router_partial; r.counter = m->router_counter;
  r.bias = L.t[DSK_ROLE_MOEGATE_BIAS].bound() ? reinterpret_cast<const float*>(L.t[DSK_ROLE_MOEGATE_BIAS].qs) : nullptr;
  r.n_active = K; r.norm_topk_prob = c.norm_topk_prob; r.scoring = c.scoring_func; r.topk_method = c.topk_method;
  r.n_group = c.n_group; r.topk_group = c.topk_group; r.scaling = c.routed_scaling_factor;
  r.active_experts = m->route_e + (size_t)l * K;
  r.active_weights = m->route_w + (size_t)l * K;
  r.scores_out = m->gate_scores + (size_t)l * E;
  if (is_kq(c.weight_quant) && c.dim % 256 == 0) { r.q_qs = m->a_xb.qs; r.q_d = m->a_xb.d; r.q_bsums = m->a_xb.bsums; }
  if (m->moe_ffn[l].grid > 0) { r.zero_ctr = m->moe_ffn[l].slot_ctr; r.zero_n = std::min(16, std::max(K, m->n_slots)); }  // re-arm the expert launch's slot counters
  if (m->lp_sh13[l] >= 0) {
    const GemvLaunch& hs = m->plans[m->lp_sh13[l]];
    PROFILED("router_gate", (double)E * c.dim * 4 + c.dim * 8.0 + hs.algo_bytes, launch_router_shared(st, r, m->plans_dev + m->lp_sh13[l], hs));
  } else {
    PROFILED("router_gate", (double)E * c.dim * 4 + c.dim * 8.0, launch_router_gate(st, r));
  }
  const bool exchange = m->sharded() && !m->class_filter;  // (class timing enqueues one kernel class only)
  if (m->moe_ffn[l].grid > 0) {  // routed experts (+ the shared expert's W2) + combine: one launch
    MoeFfnArgs a = m->moe_ffn[l];
    if (m->stage_layer == l && m->tap_qs) {
      a.tap_qs = m->tap_qs + m->tap_off_hb;
      a.tap_d = m->tap_d + m->tap_off_hb / 256;
      a.tap_stride = a.hb_stride;
    }
    if (m->profiling && !m->class_filter) {  // the kernel's own dispatch timestamps, like run_plan
      Prof p;
      DSK_TRY(prof_begin(m, "moe_ffn", a.algo_bytes, &p, false));
      DSK_TRY(launch_moe_ffn(st, a, p.e0, p.e1));
      m->ktimes[p.idx].ev.push_back({p.e0, p.e1});
    } else {
      Prof p;
      DSK_TRY(prof_begin(m, "moe_ffn", a.algo_bytes, &p));
      if (!p.skip) {
        // class timing enqueues this class alone: no router launch in front re-arms the slot counters
        if (m->class_filter) HIP_TRY(hipMemsetAsync(m->moe_ctr, 0, MOE_CTR_WORDS * 4, st));  // (the block counters re-arm themselves)
        DSK_TRY(launch_moe_ffn(st, a, nullptr, nullptr));
      }
      DSK_TRY(prof_end(&p));
    }
    return DSK_OK;
  }
  DSK_TRY(run_plan(m, "gemv_experts_w13", m->lp_w13[l]));
  DSK_TRY(run_plan(m, "gemv_experts_w2", m->lp_w2[l]));
  if (exchange || (m->sharded() && m->class_filter)) {
    // every routed slot is non-zero on exactly one rank: a sum all-reduce is exact and order-independent
    if (m->ctx->comm && exchange && m->egather) {
      // all-gather form (option "exchange_allgather"): ONE collective phase instead of the all-reduce's reduce-scatter + all-gather,
      // (world - 1) x K x dim floats received per rank instead of ~2 x K x dim.  Each rank contributes its K slot rows as they are
      // (a captured graph cannot size a message by the routing); the combine reads slot k from the copy of the rank that OWNS
      // expert k and never looks at the others.  Which form is faster is a question for an 8-GPU box.
      ncclResult_t rr = ncclAllGather(m->eout, m->egather, (size_t)K * c.dim, ncclFloat, m->ctx->comm, st);
      if (rr != ncclSuccess) DSK_FAIL(DSK_ERR_COMM, "ncclAllGather: %s", ncclGetErrorString(rr));
      m->exchange_calls++;
      PROFILED("moe_combine", (double)c.dim * (K + 3) * 4,
               launch_moe_combine_gathered(st, m->x, m->egather, m->eout, m->route_e + (size_t)l * K, m->route_w + (size_t)l * K, K,
                                           c.n_shared_experts > 0, c.dim, cdiv(E, std::max(1, m->ctx->world))));
      return DSK_OK;
    }
    if (m->ctx->comm && exchange) {  // (comm is null only in a single-rank dry run of a shard, dsk_comm_init with uid = NULL)
      ncclResult_t rr = ncclAllReduce(m->eout, m->eout, (size_t)K * c.dim, ncclFloat, ncclSum, m->ctx->comm, st);
      if (rr != ncclSuccess) DSK_FAIL(DSK_ERR_COMM, "ncclAllReduce: %s", ncclGetErrorString(rr));
      m->exchange_calls++;
    }
    PROFILED("moe_combine", (double)c.dim * (K + 3) * 4,
             launch_moe_combine(st, m->x, m->eout, m->route_w + (size_t)l * K, K, c.n_shared_experts > 0, c.dim));
  }
  return DSK_OK;
}

enum { MODE_ARGMAX = 2, MODE_SAMPLE = 3 };  // internal: OUTPUT_LOGITS + device argmax / sampling (dsk_forward_argmax, dsk_forward_sample)
// enqueue one whole token on the stream (no host synchronisation inside)
static int enqueue_forward(dsk_model* m, int mode, int max_kv) {
  const dsk_config& c = m->c;
  hipStream_t st = m->ctx->stream;
  if (!m->class_filter) HIP_TRY(hipMemcpyAsync(m->sp_dev, m->sp_host, sizeof(StepParams), hipMemcpyHostToDevice, st));
  PROFILED("embed", (double)mat_bytes(c.weight_quant, 1, c.dim), launch_embed(st, m->g[DSK_ROLE_EMBED], m->sp_dev, -1, std::max(1, c.block_size[0]),
                                                                              std::max(1, c.block_size[1]), m->x));
  for (int l = 0; l < c.n_layers; ++l) {
    if (c.use_mla) DSK_TRY(attention_mla(m, l, max_kv));
    else DSK_TRY(attention_mha(m, l, max_kv));
    DSK_TRY(ffn(m, l));
    if (m->trace) HIP_TRY(hipMemcpyAsync(m->trace_x + (size_t)l * c.dim, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToDevice, st));
  }
  if (mode == DSK_MODE_HYDRATE_KV_CACHE) return DSK_OK;  // src/infer.cpp:1284-1287
  DSK_TRY(run_plan(m, "gemv_lm_head", m->lp_head));
  if (mode == MODE_ARGMAX) {  // greedy step: the token id is all that leaves the device
    PROFILED("argmax", (double)c.vocab_size * 4, launch_argmax(st, m->logits, c.vocab_size, m->argmax_dev));
    if (!m->class_filter) HIP_TRY(hipMemcpyAsync(m->argmax_host, m->argmax_dev, 4, hipMemcpyDeviceToHost, st));
    return DSK_OK;
  }
  if (mode == MODE_SAMPLE) {  // Sampler::sample with temperature != 0: parameters and the random draw travel in StepParams
    PROFILED("sample", (double)c.vocab_size * 12, launch_sample(st, m->logits, c.vocab_size, m->sp_dev, 0.f, 0.f, 0.f, m->sample_scratch, m->argmax_dev));
    if (!m->class_filter) HIP_TRY(hipMemcpyAsync(m->argmax_host, m->argmax_dev, 4, hipMemcpyDeviceToHost, st));
    return DSK_OK;
  }
  if (!m->class_filter) HIP_TRY(hipMemcpyAsync(m->logits_host, m->logits, (size_t)c.vocab_size * 4, hipMemcpyDeviceToHost, st));
  return DSK_OK;
}

// final norm + classifier on the residual stream in m->x, logits to the pinned host buffer (dsk_hydrate's last token)
int forward_head(dsk_model* m) {
  DSK_TRY(run_plan(m, "gemv_lm_head", m->lp_head));
  HIP_TRY(hipMemcpyAsync(m->logits_host, m->logits, (size_t)m->c.vocab_size * 4, hipMemcpyDeviceToHost, m->ctx->stream));
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

// A bounded in-kernel spin gave up (kernels_moe.hip): the results of the work just synchronised are invalid.  Clears the
// flag, re-arms the arrival counters (unknown state), and retires the fused expert launch: every MoE layer falls back to
// its two-launch plans (still built: lp_w13 / lp_w2), captured graphs are dropped.  Returns whether that happened.
static bool handoff_gave_up(dsk_model* m) {
  if (!m->err_host || !*m->err_host) return false;
  *m->err_host = 0;
  hipMemset(m->moe_ctr, 0, MOE_CTR_WORDS * 4);
  hipMemset(m->moe_blk_ctr, 0, MOE_BLK_CTRS * 4);
  hipMemset(m->comb_counter, 0, (size_t)m->c.dim * 4);
  for (auto& a : m->moe_ffn) a.grid = 0;
  for (int i = 0; i < 8; ++i) {
    if (m->graph[i]) hipGraphExecDestroy(m->graph[i]);
    m->graph[i] = nullptr;
    m->graph_primed[i] = false;
  }
  m->handoff_fallbacks++;
  return true;
}

static int run_token(dsk_model* m, int token, int pos, int mode, bool retried = false) {
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  DSK_TRY(fill_step_params(m, token, pos));
  // The re-run of a token after a hand-off give-up (below) must be idempotent.  Everything a token writes is a pure function of
  // (token, pos, earlier cache rows) and is rewritten with the same values - except the attention-sink rotation: from pos >= W on
  // every layer's cache-write kernel rotates the two sink keys IN PLACE by one position (src/infer.cpp:1011-1024, 1103-1110),
  // and the first pass has already done that in every layer (the give-up does not stop the stream).  The sink loops take their
  // bound from StepParams::kv_sink and nothing else on the device reads it (kv_pos / kv_len are computed here), so the re-run
  // passes 0: the sink keys are rotated exactly once per token.
  if (retried) m->sp_host->kv_sink = 0;
  // LDS for attention scores is sized once (graph-replay safe): kv_len never exceeds the ring W nor the allocation
  const int max_kv = std::min(m->c.max_seq_len, std::max(1, m->c.rs_original_max_position_embeddings));
  // A sharded model (real communicator) is captured like the one-GPU step (option "graph_with_comm", default on): the first
  // token of a mode runs eagerly - RCCL initialises its collectives lazily, outside any capture - and the second is captured
  // with the exchange inside (validated on a 1-rank communicator: tests/test_comm_gpu.py; never run on more than one GPU).
  // (a communicator that no collective of this model uses - a generic launcher passing a uid at world 1 - costs nothing)
  const bool graphable = m->use_graph && !m->trace && !m->profiling && (!m->ctx->comm || !m->sharded() || m->graph_with_comm);
  // the long-context MLA regime enqueues one more launch per block: its own captured graph
  const bool long_mla = m->fl_part_o && m->sp_host->kv_len >= m->mla_flash_min_kv;
  // ... and so does the long-context MHA regime (split contexts: a different grid)
  const bool long_mha = !m->c.use_mla && m->mha_split > 1 && m->sp_host->kv_len >= m->mha_split_min;
  const int gi = mode + (long_mla || long_mha ? 4 : 0);  // 0 hydrate, 1 logits, 2 argmax, 3 sample
  if (graphable && !m->graph_primed[gi]) {
    m->graph_primed[gi] = true;  // first token of a mode runs eagerly (first-use initialisation), the second is captured
    DSK_TRY(enqueue_forward(m, mode, max_kv));
  } else if (graphable) {
    bool eager_instead = false;
    if (!m->graph[gi]) {
      // a graph with the RCCL exchange inside has only ever been validated on a 1-rank communicator (tests/test_comm_gpu.py):
      // if capture or instantiation fails on this ROCm / RCCL, the model drops to eager enqueueing instead of failing the token
      const bool comm_graph = m->ctx->comm && m->sharded();
      hipGraph_t g = nullptr;
      HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      int r = enqueue_forward(m, mode, max_kv);
      hipError_t e = hipStreamEndCapture(st, &g);
      hipError_t ei = hipSuccess;
      if (r == DSK_OK && e == hipSuccess) ei = hipGraphInstantiate(&m->graph[gi], g, nullptr, nullptr, 0);
      if (g) hipGraphDestroy(g);
      if (r != DSK_OK || e != hipSuccess || ei != hipSuccess) {
        m->graph[gi] = nullptr;
        if (!comm_graph) {
          if (r != DSK_OK) return r;
          DSK_FAIL(DSK_ERR_HIP, "graph capture: %s", hipGetErrorString(e != hipSuccess ? e : ei));
        }
        (void)hipGetLastError();
        dsk_clear_error();
        m->graph_with_comm = false;  // graphable is re-evaluated per token: every later token is enqueued eagerly
        m->graph_capture_fallbacks++;
        eager_instead = true;
      }
    }
    if (eager_instead) DSK_TRY(enqueue_forward(m, mode, max_kv));
    else HIP_TRY(hipGraphLaunch(m->graph[gi], st));
  } else {
    DSK_TRY(enqueue_forward(m, mode, max_kv));
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  if (handoff_gave_up(m)) {
    // The fused expert launch needs every workgroup resident at once; on a CU-masked or shared GPU its bounded spin gives
    // up.  The model then switches to the two-launch form (no in-launch waiting) for the rest of its life and THIS token is
    // run again: its side effects so far (the KV row of `pos`, the activations) are rewritten with the same values; the sink
    // rotation, the one in-place update of a token, is skipped in the re-run (above).
    if (retried) DSK_FAIL(DSK_ERR_HIP, "forward: an in-kernel hand-off timed out");
    return run_token(m, token, pos, mode, true);
  }
  return DSK_OK;
}

extern "C" int dsk_forward(dsk_model* m, int token, int pos, int mode, float* host_logits) {
  DSK_TRY(check_forward_args(m, token, pos, mode, host_logits));
  DSK_TRY(run_token(m, token, pos, mode));
  if (mode == DSK_MODE_OUTPUT_LOGITS && host_logits != m->logits_host) memcpy(host_logits, m->logits_host, (size_t)m->c.vocab_size * 4);
  return DSK_OK;
}

extern "C" float* dsk_model_host_logits(dsk_model* m) { return m && m->finalized ? m->logits_host : nullptr; }

extern "C" int dsk_forward_argmax(dsk_model* m, int token, int pos, int32_t* next_token) {
  DSK_TRY(check_forward_args(m, token, pos, DSK_MODE_OUTPUT_LOGITS, m ? m->logits_host : nullptr));
  if (!next_token) DSK_FAIL(DSK_ERR_INVALID, "forward_argmax: null output");
  DSK_TRY(run_token(m, token, pos, MODE_ARGMAX));
  *next_token = *m->argmax_host;
  return DSK_OK;
}

// Sampler::sample (src/sampler.cpp:41-75) on the device: temperature == 0 is the argmax step; otherwise softmax with
// temperature and the first index whose cumulative probability reaches coin * top_p.  `coin` = the caller's
// std::rand() / (float)RAND_MAX, so the host's random stream stays the reference's.
extern "C" int dsk_forward_sample(dsk_model* m, int token, int pos, float temperature, float top_p, float coin, int32_t* next_token) {
  if (temperature == 0.0f) return dsk_forward_argmax(m, token, pos, next_token);
  DSK_TRY(check_forward_args(m, token, pos, DSK_MODE_OUTPUT_LOGITS, m ? m->logits_host : nullptr));
  if (!next_token) DSK_FAIL(DSK_ERR_INVALID, "forward_sample: null output");
  m->sp_host->temperature = temperature;
  m->sp_host->top_p = top_p;
  m->sp_host->coin = coin;
  m->sp_host->prob_index = -1;
  DSK_TRY(run_token(m, token, pos, MODE_SAMPLE));
  *next_token = *m->argmax_host;
  return DSK_OK;
}

// Sampler::sample_prob (src/sampler.cpp:12-26) on the device: the softmax probability of `index` under the logits of
// this step -- what run_perplexity accumulates per token (src/main.cpp:386-401) -- without the logits leaving the GPU.
extern "C" int dsk_forward_prob(dsk_model* m, int token, int pos, int index, float* prob) {
  DSK_TRY(check_forward_args(m, token, pos, DSK_MODE_OUTPUT_LOGITS, m ? m->logits_host : nullptr));
  if (!prob || index < 0 || index >= m->c.vocab_size) DSK_FAIL(DSK_ERR_INVALID, "forward_prob: bad index %d", index);
  m->sp_host->temperature = 1.0f;
  m->sp_host->top_p = 1.0f;
  m->sp_host->coin = 0.0f;
  m->sp_host->prob_index = index;
  DSK_TRY(run_token(m, token, pos, MODE_SAMPLE));
  memcpy(prob, m->argmax_host, 4);
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

extern "C" int dsk_model_get_slot_outputs(dsk_model* m, float* out) {
  if (!m || !m->finalized || !out) DSK_FAIL(DSK_ERR_INVALID, "get_slot_outputs: bad argument");
  if (m->n_slots <= 0) DSK_FAIL(DSK_ERR_STATE, "get_slot_outputs: the model has no MoE layer");
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpy(out, m->eout, (size_t)m->n_slots * m->c.dim * 4, hipMemcpyDeviceToHost));
  return DSK_OK;
}

extern "C" int dsk_profile_forward(dsk_model* m, int token, int pos, dsk_kernel_time* out, int max_classes, int* n_classes) {
  DSK_TRY(check_forward_args(m, token, pos, DSK_MODE_OUTPUT_LOGITS, m ? m->logits_host : nullptr));
  if (!out || !n_classes) DSK_FAIL(DSK_ERR_INVALID, "profile_forward: null output");
  HIP_TRY(hipSetDevice(m->ctx->device));
  DSK_TRY(fill_step_params(m, token, pos));
  for (auto& k : m->ktimes) { k.launches = 0; k.algo_bytes = 0; }
  m->profiling = true;
  int r = enqueue_forward(m, DSK_MODE_OUTPUT_LOGITS, std::min(m->c.max_seq_len, std::max(1, m->c.rs_original_max_position_embeddings)));
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

// Duration of ONE kernel class inside the model: the class's launches of a whole token (every layer's own
// weights, so the stream of launches reads fresh HBM exactly as in a decode step) are enqueued back to
// back `reps` times between two HIP events on the engine stream.  Unlike dsk_profile_forward there is no
// event between launches, so the figure is what rocprofv3 --kernel-trace reports plus the ~1.5 us
// same-stream kernel boundary.  Leaves the activations / KV slot of `pos` in an undefined state.
extern "C" int dsk_time_kernel_class(dsk_model* m, const char* name, int pos, int reps, double* us_per_launch,
                                     double* bytes_per_launch, int* launches_per_token) {
  DSK_TRY(check_forward_args(m, 0, pos, DSK_MODE_OUTPUT_LOGITS, m ? m->logits_host : nullptr));
  if (!name || reps < 1 || !us_per_launch || !bytes_per_launch || !launches_per_token) DSK_FAIL(DSK_ERR_INVALID, "time_kernel_class: bad argument");
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  DSK_TRY(fill_step_params(m, 0, pos));
  HIP_TRY(hipMemcpyAsync(m->sp_dev, m->sp_host, sizeof(StepParams), hipMemcpyHostToDevice, st));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  m->class_filter = name;
  m->class_launches = 0;
  m->class_bytes = 0;
  int r = enqueue_forward(m, DSK_MODE_OUTPUT_LOGITS, std::min(m->c.max_seq_len, std::max(1, m->c.rs_original_max_position_embeddings)));  // warm-up pass (also counts the launches)
  const int per_token = m->class_launches;
  const double bytes = m->class_bytes;
  if (r == DSK_OK) r = hipEventRecord(e0, st) == hipSuccess ? DSK_OK : DSK_ERR_HIP;
  for (int i = 0; i < reps && r == DSK_OK; ++i) r = enqueue_forward(m, DSK_MODE_OUTPUT_LOGITS, std::min(m->c.max_seq_len, std::max(1, m->c.rs_original_max_position_embeddings)));
  m->class_filter = nullptr;
  if (r == DSK_OK) r = hipEventRecord(e1, st) == hipSuccess ? DSK_OK : DSK_ERR_HIP;
  if (r == DSK_OK) r = hipEventSynchronize(e1) == hipSuccess ? DSK_OK : DSK_ERR_HIP;
  float ms = 0.f;
  if (r == DSK_OK) hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (r != DSK_OK) { hipStreamSynchronize(st); if (r == DSK_ERR_HIP) dsk_set_error(r, "time_kernel_class: HIP event failure"); return r; }
  if (per_token == 0) DSK_FAIL(DSK_ERR_INVALID, "time_kernel_class: no launch of class '%s' in a token", name);
  *us_per_launch = (double)ms * 1e3 / ((double)reps * per_token);
  *bytes_per_launch = bytes / per_token;
  *launches_per_token = per_token;
  return DSK_OK;
}

// ---------------------------------------------------------------------------------
// Parity harness: ONE block (or the classifier) on a caller-supplied residual stream, with every Q8_K staging point
// of the block tapped -- teacher forcing per layer AND per quantisation point (tests/test_teacher_forced_gpu.py).
// The reference's dead DEBUG_MODEL hooks (src/infer.cpp:10-119) dump the same intermediate state on the CPU side.
// ---------------------------------------------------------------------------------
static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

static int ensure_taps(dsk_model* m) {
  if (m->tap_qs) return DSK_OK;
  const dsk_config& c = m->c;
  const int shared_n = c.n_shared_experts * c.moe_intermediate_size;
  const int hb_stride = std::max(std::max(c.moe_intermediate_size, shared_n), 1);
  const size_t hb_n = std::max<size_t>((size_t)std::max(1, m->n_slots) * hb_stride, (size_t)c.hidden_dim);
  size_t o = 0;
  m->tap_off_xattn = o; o += up256(c.dim);
  m->tap_off_qa = o; o += up256(std::max(1, c.q_lora_rank));
  m->tap_off_kva = o; o += up256(c.kv_lora_rank);
  m->tap_off_xffn = o; o += up256(c.dim);
  m->tap_off_xffn_sh = o; o += up256(c.dim);
  m->tap_off_hb = o; o += up256(hb_n);
  m->tap_off_latent = o; o += up256((size_t)c.n_heads * std::max(1, c.kv_lora_rank));
  m->tap_off_final = o; o += up256(c.dim);
  m->tap_off_att = o; o += up256((size_t)c.n_heads * c.v_head_dim);
  m->tap_total = o;
  HIP_TRY(hipMalloc((void**)&m->tap_qs, o));
  HIP_TRY(hipMalloc((void**)&m->tap_d, o / 256 * 4));
  HIP_TRY(hipMalloc((void**)&m->tap_latent, (size_t)c.n_heads * std::max(1, c.kv_lora_rank) * 4));
  HIP_TRY(hipMalloc((void**)&m->stage_x_mid, (size_t)c.dim * 4));
  return DSK_OK;
}

// point the device copy of launch plan `idx` at a tap region (stride = distance between activation groups), or restore it
static int patch_plan(dsk_model* m, int idx, size_t off, int stride, bool on) {
  if (idx < 0) return DSK_OK;
  GemvLaunch t = m->plans[idx];
  if (on) {
    t.tap_qs = m->tap_qs + off;
    t.tap_d = m->tap_d + off / 256;
    t.tap_stride = stride;
  }
  HIP_TRY(hipMemcpy(m->plans_dev + idx, &t, sizeof t, hipMemcpyHostToDevice));
  return DSK_OK;
}

static int patch_layer(dsk_model* m, int l, bool on) {
  const dsk_config& c = m->c;
  const int shared_n = c.n_shared_experts * c.moe_intermediate_size;
  const int hb_stride = std::max(std::max(c.moe_intermediate_size, shared_n), 1);
  DSK_TRY(patch_plan(m, m->lp_qkv_a[l], m->tap_off_xattn, 0, on));
  if (c.use_mla) DSK_TRY(patch_plan(m, m->lp_qkv_b[l], m->tap_off_qa, 0, on));  // wq_rope_b || wc on norm(q_a)
  if (m->att_q8_in_wo) DSK_TRY(patch_plan(m, m->lp_wo[l], m->tap_off_att, 0, on));
  DSK_TRY(patch_plan(m, m->lp_w13[l], m->tap_off_xffn, 0, on));
  DSK_TRY(patch_plan(m, m->lp_sh13[l], m->tap_off_xffn_sh, 0, on));
  DSK_TRY(patch_plan(m, m->lp_w2[l], m->tap_off_hb, m->L[l].is_moe ? hb_stride : 0, on));
  return DSK_OK;
}

extern "C" int dsk_model_run_block(dsk_model* m, int layer, const float* x_in, int pos, float* x_out) {
  if (!m || !m->finalized || !x_in || !x_out) DSK_FAIL(DSK_ERR_INVALID, "run_block: bad argument");
  if (layer < 0 || layer >= m->c.n_layers || pos < 0) DSK_FAIL(DSK_ERR_INVALID, "run_block: layer %d pos %d", layer, pos);
  if (m->ctx->world > 1) DSK_FAIL(DSK_ERR_UNSUPPORTED, "run_block: single-GPU models only");
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const dsk_config& c = m->c;
  DSK_TRY(ensure_taps(m));
  DSK_TRY(fill_step_params(m, 0, pos));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipMemset(m->tap_qs, 0x7f, m->tap_total));  // poison: a tap that did not fire is visible
  HIP_TRY(hipMemset(m->tap_d, 0xff, m->tap_total / 256 * 4));
  HIP_TRY(hipMemcpy(m->sp_dev, m->sp_host, sizeof(StepParams), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->x, x_in, (size_t)c.dim * 4, hipMemcpyHostToDevice));
  DSK_TRY(patch_layer(m, layer, true));
  m->stage_layer = layer;
  const int max_kv = std::min(c.max_seq_len, std::max(1, c.rs_original_max_position_embeddings));
  int r = c.use_mla ? attention_mla(m, layer, max_kv) : attention_mha(m, layer, max_kv);
  if (r == DSK_OK && hipMemcpyAsync(m->stage_x_mid, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) r = DSK_ERR_HIP;
  if (r == DSK_OK) r = ffn(m, layer);
  m->stage_layer = -1;
  hipError_t e = hipStreamSynchronize(st);
  int r2 = patch_layer(m, layer, false);
  if (r != DSK_OK) return r;
  if (e != hipSuccess) DSK_FAIL(DSK_ERR_HIP, "run_block: %s", hipGetErrorString(e));
  DSK_TRY(r2);
  HIP_TRY(hipGetLastError());
  if (handoff_gave_up(m)) {
    // (from pos >= W on the block has rotated its two cached sink keys in place: calling again would rotate them twice)
    if (m->sp_host->kv_sink > 0)
      DSK_FAIL(DSK_ERR_HIP, "run_block: an in-kernel hand-off timed out at pos >= %d: the block's sink keys were rotated; restore its cache rows "
                            "(dsk_model_set_cache_rows) before calling again (the model now uses the two-launch form)", m->c.rs_original_max_position_embeddings);
    DSK_FAIL(DSK_ERR_HIP, "run_block: an in-kernel hand-off timed out (the model now uses the two-launch form: call again)");
  }
  HIP_TRY(hipMemcpy(x_out, m->x, (size_t)c.dim * 4, hipMemcpyDeviceToHost));
  m->stage_kv_len = m->sp_host->kv_len;
  m->stage_last_layer = layer;
  return DSK_OK;
}

// final norm + classifier on a caller-supplied residual stream (src/infer.cpp:1292-1316); tap "q8.x_final"
extern "C" int dsk_model_run_head(dsk_model* m, const float* x_in, float* logits) {
  if (!m || !m->finalized || !x_in || !logits) DSK_FAIL(DSK_ERR_INVALID, "run_head: bad argument");
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  DSK_TRY(ensure_taps(m));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipMemcpy(m->x, x_in, (size_t)m->c.dim * 4, hipMemcpyHostToDevice));
  DSK_TRY(patch_plan(m, m->lp_head, m->tap_off_final, 0, true));
  int r = run_plan(m, "gemv_lm_head", m->lp_head);
  hipError_t e = hipStreamSynchronize(st);
  int r2 = patch_plan(m, m->lp_head, 0, 0, false);
  if (r != DSK_OK) return r;
  if (e != hipSuccess) DSK_FAIL(DSK_ERR_HIP, "run_head: %s", hipGetErrorString(e));
  DSK_TRY(r2);
  HIP_TRY(hipMemcpy(logits, m->logits, (size_t)m->c.vocab_size * 4, hipMemcpyDeviceToHost));
  return DSK_OK;
}

extern "C" int dsk_model_get_stage(dsk_model* m, const char* name, void* out, size_t bytes) {
  if (!m || !m->finalized || !name || !out) DSK_FAIL(DSK_ERR_INVALID, "get_stage: bad argument");
  if (!m->tap_qs) DSK_FAIL(DSK_ERR_STATE, "get_stage before dsk_model_run_block / dsk_model_run_head");
  HIP_TRY(hipSetDevice(m->ctx->device));
  const dsk_config& c = m->c;
  const int l = std::max(0, m->stage_last_layer);
  const Layer& L = m->L[l];
  const int H = c.n_heads, K = std::max(1, c.n_active_routed), E = std::max(1, c.n_routed_experts);
  const int shared_n = c.n_shared_experts * c.moe_intermediate_size;
  const int hb_stride = std::max(std::max(c.moe_intermediate_size, shared_n), 1);
  const size_t hb_n = std::max<size_t>((size_t)std::max(1, m->n_slots) * hb_stride, (size_t)c.hidden_dim);
  const size_t kv = (size_t)m->stage_kv_len;
  const void* src = nullptr;
  size_t avail = 0;
  const std::string s(name);
  auto f32 = [&](const float* p, size_t n) { src = p; avail = n * 4; };
  auto tapq = [&](size_t off, size_t n) { src = m->tap_qs + off; avail = n; };
  auto tapd = [&](size_t off, size_t n) { src = m->tap_d + off / 256; avail = n / 256 * 4; };
  if (s == "x_mid") f32(m->stage_x_mid, c.dim);
  else if (s == "q_a") f32(m->q_a, std::max(1, c.q_lora_rank));
  else if (s == "kv_a") f32(m->kv_a, c.kv_lora_rank + c.qk_rope_head_dim);
  else if (s == "att_out") f32(m->att_out, (size_t)H * c.v_head_dim);
  else if (s == "vb_out") f32(m->vb_out, (size_t)H * c.v_head_dim);
  else if (s == "latent_out") f32(m->tap_latent, (size_t)H * c.kv_lora_rank);
  else if (s == "q_c") f32(m->q_c, (size_t)H * c.kv_lora_rank);
  else if (s == "q_rope") f32(m->q_rope, (size_t)H * c.qk_rope_head_dim);
  else if (s == "router_logits") f32(m->router_partial, E);
  else if (s == "gate_scores") f32(m->gate_scores + (size_t)l * E, E);
  else if (s == "route_w") f32(m->route_w + (size_t)l * K, K);
  else if (s == "route_e") { src = m->route_e + (size_t)l * K; avail = (size_t)K * 4; }
  else if (s == "hb") f32(m->hb, hb_n);
  else if (s == "eout") f32(m->eout, (size_t)std::max(1, m->n_slots) * c.dim);
  else if (s == "q8.x_attn.qs") tapq(m->tap_off_xattn, c.dim);
  else if (s == "q8.x_attn.d") tapd(m->tap_off_xattn, c.dim);
  else if (s == "q8.q_a.qs") tapq(m->tap_off_qa, std::max(1, c.q_lora_rank));
  else if (s == "q8.q_a.d") tapd(m->tap_off_qa, std::max(1, c.q_lora_rank));
  else if (s == "q8.kv_a.qs") tapq(m->tap_off_kva, c.kv_lora_rank);
  else if (s == "q8.kv_a.d") tapd(m->tap_off_kva, c.kv_lora_rank);
  else if (s == "q8.x_ffn_tap.qs") tapq(m->tap_off_xffn, c.dim);
  else if (s == "q8.x_ffn_tap.d") tapd(m->tap_off_xffn, c.dim);
  else if (s == "q8.x_ffn_shared.qs") tapq(m->tap_off_xffn_sh, c.dim);
  else if (s == "q8.x_ffn_shared.d") tapd(m->tap_off_xffn_sh, c.dim);
  else if (s == "q8.x_ffn.qs") { src = m->a_xb.qs; avail = m->a_xb.qs ? (size_t)c.dim : 0; }
  else if (s == "q8.x_ffn.d") { src = m->a_xb.d; avail = m->a_xb.d ? (size_t)c.dim / 256 * 4 : 0; }
  else if (s == "q8.att.qs" && m->att_q8_in_wo) tapq(m->tap_off_att, (size_t)H * c.v_head_dim);
  else if (s == "q8.att.d" && m->att_q8_in_wo) tapd(m->tap_off_att, (size_t)H * c.v_head_dim);
  else if (s == "q8.att.qs") { src = m->a_att.qs; avail = m->a_att.qs ? (size_t)H * c.v_head_dim : 0; }
  else if (s == "q8.att.d") { src = m->a_att.d; avail = m->a_att.d ? (size_t)H * c.v_head_dim / 256 * 4 : 0; }
  else if (s == "q8.hb.qs") tapq(m->tap_off_hb, hb_n);
  else if (s == "q8.hb.d") tapd(m->tap_off_hb, hb_n);
  else if (s == "q8.latent.qs") tapq(m->tap_off_latent, (size_t)H * c.kv_lora_rank);
  else if (s == "q8.latent.d") tapd(m->tap_off_latent, (size_t)H * c.kv_lora_rank);
  else if (s == "q8.x_final.qs") tapq(m->tap_off_final, c.dim);
  else if (s == "q8.x_final.d") tapd(m->tap_off_final, c.dim);
  else if (s == "k_cache") { src = L.key_cache; avail = L.key_cache ? kv * H * m->head_dim * 2 : 0; }
  else if (s == "v_cache") { src = L.value_cache; avail = L.value_cache ? kv * H * c.v_head_dim * 2 : 0; }
  else if (s == "nope_cache") { src = L.nope_cache; avail = L.nope_cache ? kv * c.kv_lora_rank * 2 : 0; }
  else if (s == "rope_cache") { src = L.rope_cache; avail = L.rope_cache ? kv * c.qk_rope_head_dim * 2 : 0; }
  else DSK_FAIL(DSK_ERR_INVALID, "get_stage: unknown stage '%s'", name);
  if (!src || bytes > avail) DSK_FAIL(DSK_ERR_INVALID, "get_stage: '%s' holds %zu bytes, %zu requested", name, avail, bytes);
  HIP_TRY(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return DSK_OK;
}

// Parity harness: overwrite rows [row0, row0 + nrows) of one KV cache of `layer` with caller-supplied f16 bits, so that a
// block can be audited at a LONG context without decoding thousands of tokens first (tests/test_teacher_forced_gpu.py:
// the attention regimes that only exist from 320 / 1024 cached positions on, at full DeepSeek-V3 width).
extern "C" int dsk_model_set_cache_rows(dsk_model* m, int layer, const char* cache, int row0, int nrows, const uint16_t* rows) {
  if (!m || !m->finalized || !cache || !rows) DSK_FAIL(DSK_ERR_INVALID, "set_cache_rows: bad argument");
  if (layer < 0 || layer >= m->c.n_layers || row0 < 0 || nrows < 1 || row0 + nrows > m->c.max_seq_len)
    DSK_FAIL(DSK_ERR_INVALID, "set_cache_rows: layer %d rows [%d, %d) of %d", layer, row0, row0 + nrows, m->c.max_seq_len);
  HIP_TRY(hipSetDevice(m->ctx->device));
  const dsk_config& c = m->c;
  const Layer& L = m->L[layer];
  const std::string s(cache);
  uint16_t* base = nullptr;
  size_t width = 0;
  if (s == "k_cache") { base = L.key_cache; width = (size_t)c.n_heads * m->head_dim; }
  else if (s == "v_cache") { base = L.value_cache; width = (size_t)c.n_heads * c.v_head_dim; }
  else if (s == "nope_cache") { base = L.nope_cache; width = (size_t)c.kv_lora_rank; }
  else if (s == "rope_cache") { base = L.rope_cache; width = (size_t)c.qk_rope_head_dim; }
  else DSK_FAIL(DSK_ERR_INVALID, "set_cache_rows: unknown cache '%s'", cache);
  if (!base) DSK_FAIL(DSK_ERR_INVALID, "set_cache_rows: this model has no '%s'", cache);
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  HIP_TRY(hipMemcpy(base + (size_t)row0 * width, rows, (size_t)nrows * width * 2, hipMemcpyHostToDevice));
  return DSK_OK;
}

// ... and the read side: rows [row0, row0 + nrows) of one KV cache of `layer` as f16 bits (tests/test_hydrate_gpu.py compares the
// caches a batched prompt left with the ones the per-token loop leaves)
extern "C" int dsk_model_get_cache_rows(dsk_model* m, int layer, const char* cache, int row0, int nrows, uint16_t* rows) {
  if (!m || !m->finalized || !cache || !rows) DSK_FAIL(DSK_ERR_INVALID, "get_cache_rows: bad argument");
  if (layer < 0 || layer >= m->c.n_layers || row0 < 0 || nrows < 1 || row0 + nrows > m->c.max_seq_len)
    DSK_FAIL(DSK_ERR_INVALID, "get_cache_rows: layer %d rows [%d, %d) of %d", layer, row0, row0 + nrows, m->c.max_seq_len);
  HIP_TRY(hipSetDevice(m->ctx->device));
  const dsk_config& c = m->c;
  const Layer& L = m->L[layer];
  const std::string s(cache);
  const uint16_t* base = nullptr;
  size_t width = 0;
  if (s == "k_cache") { base = L.key_cache; width = (size_t)c.n_heads * m->head_dim; }
  else if (s == "v_cache") { base = L.value_cache; width = (size_t)c.n_heads * c.v_head_dim; }
  else if (s == "nope_cache") { base = L.nope_cache; width = (size_t)c.kv_lora_rank; }
  else if (s == "rope_cache") { base = L.rope_cache; width = (size_t)c.qk_rope_head_dim; }
  else DSK_FAIL(DSK_ERR_INVALID, "get_cache_rows: unknown cache '%s'", cache);
  if (!base) DSK_FAIL(DSK_ERR_INVALID, "get_cache_rows: this model has no '%s'", cache);
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  HIP_TRY(hipMemcpy(rows, base + (size_t)row0 * width, (size_t)nrows * width * 2, hipMemcpyDeviceToHost));
  return DSK_OK;
}

// debug: the 8 wall-clock stamps (100 MHz) per workgroup of the last fused expert launch (DSK_MOE_TIMELINE=1 at model creation)
extern "C" int dsk_model_get_timeline(dsk_model* m, int kind, unsigned long long* out, int n_wgs) {
  if (!m || !out || n_wgs < 1 || n_wgs > 1024 || kind < 0 || kind > 7) DSK_FAIL(DSK_ERR_INVALID, "get_timeline: bad argument");
  if (!m->moe_timeline) DSK_FAIL(DSK_ERR_STATE, "get_timeline: set DSK_TIMELINE=1 before creating the model");
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpy(out, m->timeline_of(kind), (size_t)n_wgs * 64, hipMemcpyDeviceToHost));
  return DSK_OK;
}
extern "C" int dsk_model_get_moe_timeline(dsk_model* m, unsigned long long* out, int n_wgs) { return dsk_model_get_timeline(m, 4, out, n_wgs); }
