// ops_api.cpp -- op-level C-ABI entry points on host buffers (copy in, run the SAME kernels the
// token step uses, copy out).  They mirror the reference's "exposed for tests" functions
// (src/model.h:503-535) plus its file-static helpers, so that the parity tests can pin every
// stage of the hot path separately.
#include "dsk_internal.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {
struct DevBuf {  // small RAII device allocation for one call
  void* p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  int alloc(size_t bytes) {
    HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
    return DSK_OK;
  }
  template <typename T>
  T* as() { return reinterpret_cast<T*>(p); }
};
int cdiv(int a, int b) { return (a + b - 1) / b; }
int begin(dsk_ctx* ctx) {
  if (!ctx) DSK_FAIL(DSK_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(ctx_device(ctx)));
  return DSK_OK;
}
int finish(dsk_ctx* ctx) {
  HIP_TRY(hipStreamSynchronize(ctx_stream(ctx)));
  HIP_TRY(hipGetLastError());
  return DSK_OK;
}
struct TensorGuard {
  DTensor t;
  ~TensorGuard() {
    if (t.base) hipFree(t.base);
  }
};
void fill_rope_cs(float* cs, int d, int pos, float theta) {  // src/infer.cpp:655-658
  for (int j = 0; j < d / 2; ++j) {
    const float freq = powf(theta, -((float)(2 * j) * (1.0f / (float)d)));  // see oracle/dsk_oracle.c ref_rope_freq
    const float v = pos * freq;
    cs[2 * j] = cosf(v);
    cs[2 * j + 1] = sinf(v);
  }
}
}  // namespace

extern "C" int dsk_q8k_quantize(dsk_ctx* ctx, const float* x, int n, int8_t* qs, float* d, int16_t* bsums) {
  DSK_TRY(begin(ctx));
  if (!x || !qs || !d || !bsums) DSK_FAIL(DSK_ERR_INVALID, "q8k_quantize: null buffer");
  if (n <= 0 || n % 256) DSK_FAIL(DSK_ERR_INVALID, "q8k_quantize: n=%d must be a positive multiple of 256", n);
  DevBuf dx, dq, dd, db;
  DSK_TRY(dx.alloc((size_t)n * 4));
  DSK_TRY(dq.alloc(n));
  DSK_TRY(dd.alloc((size_t)n / 256 * 4));
  DSK_TRY(db.alloc((size_t)n / 16 * 2));
  hipStream_t st = ctx_stream(ctx);
  HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
  DSK_TRY(launch_quantize_q8k(st, dx.as<float>(), n, dq.as<int8_t>(), dd.as<float>(), db.as<int16_t>()));
  HIP_TRY(hipMemcpyAsync(qs, dq.p, n, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(d, dd.p, (size_t)n / 256 * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(bsums, db.p, (size_t)n / 16 * 2, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

static int gemv_common(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale, const int32_t* block_size,
                       int n_experts, int expert, int d, int n, const float* x, float* out) {
  DSK_TRY(begin(ctx));
  if (!w || !x || !out) DSK_FAIL(DSK_ERR_INVALID, "gemv: null buffer");
  if (quant < DSK_QUANT_F32 || quant > DSK_QUANT_Q3_K) DSK_FAIL(DSK_ERR_INVALID, "gemv: bad quant %d", quant);
  if (d <= 0 || n <= 0) DSK_FAIL(DSK_ERR_INVALID, "gemv: empty shape");
  if (is_kq(quant) && n % 256) DSK_FAIL(DSK_ERR_INVALID, "gemv: k-quant n=%d not a multiple of 256", n);
  const size_t mats = n_experts > 0 ? n_experts : 1;
  if (w_bytes != mat_bytes(quant, d, n) * mats) DSK_FAIL(DSK_ERR_INVALID, "gemv: %zu weight bytes, expected %zu", w_bytes, mat_bytes(quant, d, n) * mats);
  if (n_experts > 0 && (expert < 0 || expert >= n_experts)) DSK_FAIL(DSK_ERR_INVALID, "gemv: expert %d of %d", expert, n_experts);
  int b0 = 1, b1 = 1;
  if (scale) {
    if (!block_size || block_size[0] <= 0 || block_size[1] <= 0) DSK_FAIL(DSK_ERR_INVALID, "gemv: scale without block_size");
    b0 = block_size[0];
    b1 = block_size[1];
  }
  hipStream_t st = ctx_stream(ctx);
  TensorGuard tg;
  // upload only the selected expert's slice (the engine proper keeps whole stacks resident)
  DTensor& t = tg.t;
  t.tiled = quant == DSK_QUANT_Q2_K;  // the layout the model's launches run (tile_device.h)
  DSK_TRY(alloc_tensor(b0, b1, t, quant, 0, d, n, 1, 0));
  const char* src = (const char*)w + (size_t)(n_experts > 0 ? expert : 0) * mat_bytes(quant, d, n);
  DSK_TRY(upload_tensor(ctx, t, src));
  if (scale && quant != DSK_QUANT_F8E5M2 && quant != DSK_QUANT_F16 && quant != DSK_QUANT_F32)
    DSK_FAIL(DSK_ERR_INVALID, "gemv: block scales with a k-quant");
  DevBuf dscale;
  const float* dsc = nullptr;
  const size_t nsc = (size_t)cdiv(d, b0) * cdiv(n, b1);
  if (scale) {
    DSK_TRY(dscale.alloc(nsc * 4));
    HIP_TRY(hipMemcpyAsync(dscale.p, scale + (size_t)(n_experts > 0 ? expert : 0) * nsc, nsc * 4, hipMemcpyHostToDevice, st));
    dsc = dscale.as<float>();
  }
  DevBuf dx, dout, dq, dd, db;
  DSK_TRY(dx.alloc((size_t)n * 4));
  DSK_TRY(dout.alloc((size_t)d * 4));
  HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
  // one-task launch of the same kernel the token step uses; the Q8_K quantisation of x
  // (quantize_acts, src/infer.cpp:327-337) happens in the kernel prologue (ACT_F32)
  GemvLaunch h;
  memset(&h, 0, sizeof h);
  h.quant = quant; h.n_tasks = 1; h.b0 = b0; h.b1 = b1;
  GemvTask& T = h.t[0];
  T.qs = t.qs; T.sc = t.sc; T.hm = t.hm; T.dm = t.dm; T.scale = dsc;
  T.rows = d; T.n = n; T.local_experts = 1;
  h.tiled = t.tiled;
  T.act_mode = ACT_F32; T.a_f32 = dx.as<float>();
  T.out = dout.as<float>(); T.epilogue = EPI_STORE;
  DSK_TRY(gemv_plan(h, 1024));
  DevBuf dh;
  DSK_TRY(dh.alloc(sizeof h));
  HIP_TRY(hipMemcpyAsync(dh.p, &h, sizeof h, hipMemcpyHostToDevice, st));
  DSK_TRY(gemv_launch(st, dh.as<GemvLaunch>(), h));
  HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)d * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_gemv(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale, const int32_t block_size[2], int d,
                        int n, const float* x, float* out) {
  return gemv_common(ctx, quant, w, w_bytes, scale, block_size, 0, 0, d, n, x, out);
}
extern "C" int dsk_gemv_expert(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale, const int32_t block_size[2],
                               int n_experts, int expert, int d, int n, const float* x, float* out) {
  if (n_experts <= 0) DSK_FAIL(DSK_ERR_INVALID, "gemv_expert: n_experts must be positive");
  return gemv_common(ctx, quant, w, w_bytes, scale, block_size, n_experts, expert, d, n, x, out);
}

extern "C" int dsk_embed_row(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale, const int32_t block_size[2],
                             int vocab, int dim, int token, float* out) {
  DSK_TRY(begin(ctx));
  if (!w || !out) DSK_FAIL(DSK_ERR_INVALID, "embed_row: null buffer");
  if (token < 0 || token >= vocab) DSK_FAIL(DSK_ERR_INVALID, "embed_row: token %d of %d", token, vocab);
  if (is_kq(quant) && dim % 256) DSK_FAIL(DSK_ERR_INVALID, "embed_row: k-quant dim %d", dim);
  if (w_bytes != mat_bytes(quant, vocab, dim)) DSK_FAIL(DSK_ERR_INVALID, "embed_row: %zu bytes, expected %zu", w_bytes, mat_bytes(quant, vocab, dim));
  int b0 = 1, b1 = 1;
  if (quant == DSK_QUANT_F8E5M2) {
    if (!scale || !block_size || block_size[0] <= 0 || block_size[1] <= 0) DSK_FAIL(DSK_ERR_INVALID, "embed_row: f8e5m2 needs scales");
    b0 = block_size[0];
    b1 = block_size[1];
  }
  hipStream_t st = ctx_stream(ctx);
  TensorGuard tg;
  tg.t.tiled = quant == DSK_QUANT_Q2_K;
  DSK_TRY(alloc_tensor(b0, b1, tg.t, quant, 0, vocab, dim, 1, 0));
  DSK_TRY(upload_tensor(ctx, tg.t, w));
  if (quant == DSK_QUANT_F8E5M2)
    HIP_TRY(hipMemcpyAsync(tg.t.scale, scale, (size_t)cdiv(vocab, b0) * cdiv(dim, b1) * 4, hipMemcpyHostToDevice, st));
  DevBuf dout;
  DSK_TRY(dout.alloc((size_t)dim * 4));
  DSK_TRY(launch_embed(st, tg.t, nullptr, token, b0, b1, dout.as<float>()));
  HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)dim * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_rmsnorm(dsk_ctx* ctx, const float* x, const float* weight, int size, float eps, float* out) {
  DSK_TRY(begin(ctx));
  if (!x || !weight || !out || size <= 0) DSK_FAIL(DSK_ERR_INVALID, "rmsnorm: bad argument");
  hipStream_t st = ctx_stream(ctx);
  DevBuf dx, dw, dy;
  DSK_TRY(dx.alloc((size_t)size * 4));
  DSK_TRY(dw.alloc((size_t)size * 4));
  DSK_TRY(dy.alloc((size_t)size * 4));
  HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)size * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dw.p, weight, (size_t)size * 4, hipMemcpyHostToDevice, st));
  NormJob j;
  memset(&j, 0, sizeof j);
  j.x = dx.as<float>(); j.weight = dw.as<float>(); j.n = size; j.eps = eps; j.y_f32 = dy.as<float>();
  DSK_TRY(launch_norm_jobs(st, &j, 1, nullptr));
  HIP_TRY(hipMemcpyAsync(out, dy.p, (size_t)size * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

// Sampler::sample / sample_argmax on a host logits vector (the op behind dsk_forward_sample / dsk_forward_argmax)
extern "C" int dsk_sample(dsk_ctx* ctx, const float* logits, int vocab_size, float temperature, float top_p, float coin, int32_t* token) {
  DSK_TRY(begin(ctx));
  if (!logits || !token || vocab_size <= 0) DSK_FAIL(DSK_ERR_INVALID, "sample: bad argument");
  hipStream_t st = ctx_stream(ctx);
  DevBuf dl, dout, dscr;
  DSK_TRY(dl.alloc((size_t)vocab_size * 4));
  DSK_TRY(dscr.alloc(sample_scratch_bytes()));
  HIP_TRY(hipMemsetAsync(dscr.p, 0, sample_scratch_bytes(), st));
  DSK_TRY(dout.alloc(16));
  HIP_TRY(hipMemcpyAsync(dl.p, logits, (size_t)vocab_size * 4, hipMemcpyHostToDevice, st));
  if (temperature == 0.0f) DSK_TRY(launch_argmax(st, dl.as<float>(), vocab_size, dout.as<int>()));
  else DSK_TRY(launch_sample(st, dl.as<float>(), vocab_size, nullptr, temperature, top_p, coin, dscr.as<float>(), dout.as<int>()));
  HIP_TRY(hipMemcpyAsync(token, dout.p, 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_moe_gate(dsk_ctx* ctx, const float* scores, const float* bias, int n_routed, int n_active, int norm_topk_prob,
                            float routed_scaling_factor, int scoring_func, int topk_method, int n_group, int topk_group,
                            int32_t* active_experts, float* active_weights) {
  DSK_TRY(begin(ctx));
  if (!scores || !active_experts || !active_weights || n_routed <= 0 || n_active <= 0) DSK_FAIL(DSK_ERR_INVALID, "moe_gate: bad argument");
  hipStream_t st = ctx_stream(ctx);
  DevBuf ds, dbias, de, dw;
  DSK_TRY(ds.alloc((size_t)n_routed * 4));
  DSK_TRY(de.alloc((size_t)n_active * 4));
  DSK_TRY(dw.alloc((size_t)n_active * 4));
  HIP_TRY(hipMemcpyAsync(ds.p, scores, (size_t)n_routed * 4, hipMemcpyHostToDevice, st));
  if (bias) {
    DSK_TRY(dbias.alloc((size_t)n_routed * 4));
    HIP_TRY(hipMemcpyAsync(dbias.p, bias, (size_t)n_routed * 4, hipMemcpyHostToDevice, st));
  }
  DSK_TRY(launch_gate(st, ds.as<float>(), 1, bias ? dbias.as<float>() : nullptr, n_routed, n_active, norm_topk_prob, routed_scaling_factor,
                      scoring_func, topk_method, n_group, topk_group, de.as<int>(), dw.as<float>(), nullptr));
  HIP_TRY(hipMemcpyAsync(active_experts, de.p, (size_t)n_active * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(active_weights, dw.p, (size_t)n_active * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_rope(dsk_ctx* ctx, float* vec, int n_heads, int d, int pos, float theta, int is_v3) {
  DSK_TRY(begin(ctx));
  if (!vec || n_heads <= 0 || d <= 0 || (d & 1) || d > 128) DSK_FAIL(DSK_ERR_INVALID, "rope: bad argument (d even, <= 128)");
  hipStream_t st = ctx_stream(ctx);
  float cs[128];
  fill_rope_cs(cs, d, pos, theta);
  DevBuf dv, dcs;
  DSK_TRY(dv.alloc((size_t)n_heads * d * 4));
  DSK_TRY(dcs.alloc(sizeof cs));
  HIP_TRY(hipMemcpyAsync(dv.p, vec, (size_t)n_heads * d * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dcs.p, cs, sizeof cs, hipMemcpyHostToDevice, st));
  DSK_TRY(launch_rope_only(st, dv.as<float>(), n_heads, d, dcs.as<float>(), is_v3));
  HIP_TRY(hipMemcpyAsync(vec, dv.p, (size_t)n_heads * d * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_attn_mha(dsk_ctx* ctx, const float* q, const uint16_t* kb, const uint16_t* vb, int n_heads, int head_dim, int v_head_dim,
                            int kv_len, float* out) {
  DSK_TRY(begin(ctx));
  if (!q || !kb || !vb || !out || n_heads <= 0 || kv_len <= 0) DSK_FAIL(DSK_ERR_INVALID, "attn_mha: bad argument");
  hipStream_t st = ctx_stream(ctx);
  DevBuf dq, dk, dv, dout;
  const size_t kb_n = (size_t)kv_len * n_heads * head_dim, vb_n = (size_t)kv_len * n_heads * v_head_dim;
  DSK_TRY(dq.alloc((size_t)n_heads * head_dim * 4));
  DSK_TRY(dk.alloc(kb_n * 2));
  DSK_TRY(dv.alloc(vb_n * 2));
  DSK_TRY(dout.alloc((size_t)n_heads * v_head_dim * 4));
  HIP_TRY(hipMemcpyAsync(dq.p, q, (size_t)n_heads * head_dim * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dk.p, kb, kb_n * 2, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dv.p, vb, vb_n * 2, hipMemcpyHostToDevice, st));
  AttnMhaArgs a;
  memset(&a, 0, sizeof a);
  memset(&a, 0, sizeof a);
  a.q = dq.as<float>(); a.key_cache = dk.as<uint16_t>(); a.value_cache = dv.as<uint16_t>(); a.out = dout.as<float>();
  a.n_heads = n_heads; a.head_dim = head_dim; a.v_dim = v_head_dim;
  DSK_TRY(launch_attn_mha(st, a, nullptr, kv_len, kv_len));
  HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n_heads * v_head_dim * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_attn_mla(dsk_ctx* ctx, const float* q_c, const float* q_rope, const uint16_t* ckv, const uint16_t* krope, int n_heads,
                            int head_dim, int kv_lora_rank, int rope_dim, int kv_len, float* out) {
  DSK_TRY(begin(ctx));
  if (!q_c || !q_rope || !ckv || !krope || !out || n_heads <= 0 || kv_len <= 0) DSK_FAIL(DSK_ERR_INVALID, "attn_mla: bad argument");
  hipStream_t st = ctx_stream(ctx);
  DevBuf dqc, dqr, dc, dr, dout;
  DSK_TRY(dqc.alloc((size_t)n_heads * kv_lora_rank * 4));
  DSK_TRY(dqr.alloc((size_t)n_heads * rope_dim * 4));
  DSK_TRY(dc.alloc((size_t)kv_len * kv_lora_rank * 2));
  DSK_TRY(dr.alloc((size_t)kv_len * rope_dim * 2));
  DSK_TRY(dout.alloc((size_t)n_heads * kv_lora_rank * 4));
  HIP_TRY(hipMemcpyAsync(dqc.p, q_c, (size_t)n_heads * kv_lora_rank * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dqr.p, q_rope, (size_t)n_heads * rope_dim * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dc.p, ckv, (size_t)kv_len * kv_lora_rank * 2, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dr.p, krope, (size_t)kv_len * rope_dim * 2, hipMemcpyHostToDevice, st));
  AttnMlaArgs a;
  memset(&a, 0, sizeof a);
  a.q_c = dqc.as<float>(); a.q_rope = dqr.as<float>(); a.nope_cache = dc.as<uint16_t>(); a.rope_cache = dr.as<uint16_t>();
  a.out = dout.as<float>(); a.n_heads = n_heads; a.head_dim = head_dim; a.rope = rope_dim; a.lora = kv_lora_rank;
  if (kv_len >= MLA_FLASH_MIN_KV && kv_lora_rank == 512 && rope_dim == 64) {  // long context: matrix-core path (partials + merge)
    MlaFlashArgs f;
    memset(&f, 0, sizeof f);
    f.q_c = a.q_c; f.q_rope = a.q_rope; f.rotate_q = 0; f.nope_cache = a.nope_cache; f.rope_cache = a.rope_cache;
    f.n_heads = n_heads; f.head_dim = head_dim; f.lora = kv_lora_rank; f.rope = rope_dim;
    f.n_chunks = 64; f.chunk_len = 0;  // derived from kv_len in the kernels (MLA_FL_CHUNK), as in the model
    DevBuf po, pml;
    DSK_TRY(po.alloc((size_t)f.n_chunks * n_heads * kv_lora_rank * 4));
    DSK_TRY(pml.alloc((size_t)f.n_chunks * n_heads * 8));
    f.part_o = po.as<float>(); f.part_ml = pml.as<float>();
    DSK_TRY(launch_mla_flash(st, f, nullptr, kv_len));
    DSK_TRY(launch_mla_merge(st, f, nullptr, kv_len, dout.as<float>()));
    HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n_heads * kv_lora_rank * 4, hipMemcpyDeviceToHost, st));
    return finish(ctx);
  }
  DSK_TRY(launch_attn_mla(st, a, nullptr, kv_len, kv_len));
  HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n_heads * kv_lora_rank * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}

extern "C" int dsk_measure_read_bw(dsk_ctx* ctx, size_t bytes, int iters, double* gbps_out) {
  DSK_TRY(begin(ctx));
  if (!gbps_out || bytes < (1u << 25) || iters <= 0) DSK_FAIL(DSK_ERR_INVALID, "measure_read_bw: bad argument (at least 32 MiB)");
  bytes = bytes / (4096u * 8192u) * (4096u * 8192u);  // 4096 waves x whole 8 KiB steps: every byte counted is read
  hipStream_t st = ctx_stream(ctx);
  DevBuf buf, sink;
  DSK_TRY(buf.alloc(bytes));
  DSK_TRY(sink.alloc(16));
  HIP_TRY(hipMemsetAsync(buf.p, 0x5a, bytes, st));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  double best = 0;
  for (int i = 0; i < iters + 1; ++i) {
    HIP_TRY(hipEventRecord(e0, st));
    DSK_TRY(launch_read_bw(st, buf.p, bytes, sink.as<float>()));
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (i > 0 && ms > 0) best = std::max(best, (double)bytes / (ms * 1e-3) / 1e9);
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *gbps_out = best;
  return finish(ctx);
}

// ------------------------------------------------------------------------------------
// micro-benchmark of the GEMV kernel on device-resident synthetic weights (diagnostics; used by
// tools/kbench.py to pick launch geometry).  kind: 0 plain, 1 GLU pair, 2 or 3 per-slot MoE W2 + fused combine.
// Weight sets are rotated through > 512 MB so that the 256 MB Infinity Cache cannot serve them.
// ------------------------------------------------------------------------------------
// The launch planner by itself (host logic, no GPU): the geometry gemv_plan picks for n_tasks equal (rows x n) matrices of
// one activation group.  kind as in dsk_bench_gemv (0 plain, 1 GLU pair, 2/3 fused-combine tasks); out[8] = lanes per row,
// R, U, waves per workgroup, grid, LDS bytes, activation groups, rows per workgroup step.
extern "C" int dsk_plan_gemv(int quant, int rows, int n, int n_tasks, int kind, int act_mode, int target_wgs, int* out) {
  if (!out || rows < 1 || n < 1 || n_tasks < 1 || n_tasks > GEMV_MAX_TASKS) DSK_FAIL(DSK_ERR_INVALID, "plan_gemv: bad argument");
  GemvLaunch h;
  memset(&h, 0, sizeof h);
  const bool tiled = (kind & 0x100) != 0;  // kind | 0x100: the weights in the tiled Q2_K layout (tile_device.h; kernels_tile.hip plans it)
  kind &= 0xff;
  h.tiled = tiled;
  h.quant = quant; h.glu = kind == 1; h.act = DSK_ACT_SILU; h.b0 = h.b1 = 128;
  static float dummy_x[4], dummy_res[4];
  static unsigned dummy_cnt[4];
  static int8_t dummy_q[4];
  for (int i = 0; i < n_tasks; ++i) {
    GemvTask& T = h.t[h.n_tasks++];
    T.qs = T.sc = T.hm = T.dm = reinterpret_cast<const uint8_t*>(dummy_q);  // never dereferenced: planning reads shapes only
    if (kind == 1) T.qs2 = T.sc2 = T.hm2 = T.dm2 = T.qs;
    T.rows = rows; T.n = n; T.local_experts = 1; T.act_mode = act_mode; T.wt_tiled = tiled;
    T.a_qs = dummy_q; T.a_f32 = dummy_x + (kind >= 2 ? i : 0); T.norm_w = dummy_x; T.eps = 1e-6f;
  }
  if (kind >= 2) { h.comb_x = dummy_res; h.comb_counter = dummy_cnt; }
  DSK_TRY(gemv_plan(h, target_wgs > 0 ? target_wgs : 1024));
  const int lpr = 1 << h.lpr_log2;
  out[0] = lpr; out[1] = h.R; out[2] = h.U; out[3] = h.NW; out[4] = h.grid; out[5] = (int)h.lds_bytes; out[6] = h.n_groups;
  out[7] = tiled ? h.t_rcap : h.NW * (64 / lpr) * h.R;  // tiled: item partials a round of strips may hold in LDS
  return DSK_OK;
}

// Which "weights ahead of the staging" kernel (kernels_gemv.hip, round 6) the engine runs for such a launch at its default options:
// 0 none (gemv_kernel), 1 gemv_ahead_kernel, 2 gemv_ahead_q8_kernel, 3 gemv_kvwrite_ahead_kernel (kvwrite != 0: the MLA second-stage
// launch with its cache-write rider).  Host only, like dsk_plan_gemv, whose arguments it shares.
extern "C" int dsk_plan_gemv_ahead(int quant, int rows, int n, int n_tasks, int kind, int act_mode, int kvwrite, int* ahead_kind) {
  if (!ahead_kind || rows < 1 || n < 1 || n_tasks < 1 || n_tasks > GEMV_MAX_TASKS) DSK_FAIL(DSK_ERR_INVALID, "plan_gemv_ahead: bad argument");
  GemvLaunch h;
  memset(&h, 0, sizeof h);
  h.quant = quant; h.glu = kind == 1; h.act = DSK_ACT_SILU; h.b0 = h.b1 = 128;
  h.ahead = 3;
  if (kvwrite) h.reserve_wgs = 1;
  static float dummy_x[4], dummy_res[4];
  static unsigned dummy_cnt[4];
  static int8_t dummy_q[4];
  for (int i = 0; i < n_tasks; ++i) {
    GemvTask& T = h.t[h.n_tasks++];
    T.qs = T.sc = T.hm = T.dm = reinterpret_cast<const uint8_t*>(dummy_q);
    if (kind == 1) T.qs2 = T.sc2 = T.hm2 = T.dm2 = T.qs;
    T.rows = rows; T.n = n; T.local_experts = 1; T.act_mode = act_mode;
    T.a_qs = dummy_q; T.a_f32 = dummy_x + (kind >= 2 ? i : 0); T.norm_w = dummy_x; T.eps = 1e-6f;
  }
  if (kind >= 2) { h.comb_x = dummy_res; h.comb_counter = dummy_cnt; }
  DSK_TRY(gemv_plan(h, 1024));
  *ahead_kind = gemv_ahead_kind(h, kvwrite != 0);
  return DSK_OK;
}

extern "C" int dsk_bench_gemv(dsk_ctx* ctx, int quant, int rows, int n, int n_tasks, int kind, int act_mode, int force_lpr,
                              int force_R, int force_U, int target_wgs, int iters, double* us_per_launch, double* bytes_per_launch) {
  DSK_TRY(begin(ctx));
  if (!us_per_launch || !bytes_per_launch || n_tasks < 1 || n_tasks > GEMV_MAX_TASKS || iters < 1) DSK_FAIL(DSK_ERR_INVALID, "bench_gemv: bad argument");
  hipStream_t st = ctx_stream(ctx);
  const int mats = n_tasks * (kind == 1 ? 2 : 1);
  const double wbytes = (double)mat_bytes(quant, rows, n) * mats;
  int copies = (int)(600e6 / wbytes) + 1;
  if (copies > 48) copies = 48;
  if (copies < 2) copies = 2;
  if (dsk_ab_env("DSK_BENCH_COPIES")) copies = std::max(1, atoi(dsk_ab_env("DSK_BENCH_COPIES")));  // 1: the Infinity Cache serves the weights
  std::vector<TensorGuard> W((size_t)copies * mats);
  for (size_t i = 0; i < W.size(); ++i) {
    W[i].t.tiled = quant == DSK_QUANT_Q2_K && !dsk_ab_env("DSK_BENCH_PLANES");
    DSK_TRY(alloc_tensor(128, 128, W[i].t, quant, 0, rows, n, 1, 0));
    DSK_TRY(launch_fill_tensor(st, W[i].t, 1234 + i, 1.0f / sqrtf((float)n)));
  }
  DevBuf x, nw, out, q8, wts, plans, xres, cnt;
  DSK_TRY(xres.alloc((size_t)rows * 4));
  DSK_TRY(cnt.alloc((size_t)rows * 4));
  HIP_TRY(hipMemsetAsync(xres.p, 0, (size_t)rows * 4, st));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, (size_t)rows * 4, st));
  DSK_TRY(x.alloc((size_t)n * n_tasks * 4));
  DSK_TRY(nw.alloc((size_t)n * 4));
  DSK_TRY(out.alloc((size_t)rows * n_tasks * 4));
  DSK_TRY(q8.alloc((size_t)n * 2));
  DSK_TRY(wts.alloc(64));
  DSK_TRY(launch_fill_f32(st, x.as<float>(), (size_t)n * n_tasks, 7, 0.f, 1.f));
  DSK_TRY(launch_fill_f32(st, nw.as<float>(), (size_t)n, 8, 1.f, 0.1f));
  DSK_TRY(launch_fill_f32(st, wts.as<float>(), 16, 9, 1.f, 0.1f));
  HIP_TRY(hipMemsetAsync(out.p, 0, (size_t)rows * n_tasks * 4, st));
  int8_t* aq = q8.as<int8_t>();
  float* ad = reinterpret_cast<float*>(aq + n);
  int16_t* ab = reinterpret_cast<int16_t*>(aq + n + (n / 256 + 4) * 4);
  if (is_kq(quant)) DSK_TRY(launch_quantize_q8k(st, x.as<float>(), n, aq, ad, ab));
  std::vector<GemvLaunch> H(copies);
  for (int c = 0; c < copies; ++c) {
    GemvLaunch& h = H[c];
    memset(&h, 0, sizeof h);
    h.quant = quant; h.glu = kind == 1; h.act = DSK_ACT_SILU;
    h.tiled = W[0].t.tiled;
    h.b0 = h.b1 = 128; h.force_lpr = force_lpr; h.force_R = force_R; h.force_U = force_U;
    h.ahead = dsk_ab_env("DSK_NO_AHEAD") ? 0 : 3;  // the engine's default: where a plan fits an ahead kernel (kernels_gemv.hip) the op-level timing runs it too
    if (h.tiled) { h.force_NW = force_R; h.force_U = target_wgs > 8 ? target_wgs : 0; }  // tiled launches: R -> waves per workgroup, target_wgs -> workgroups
    if (dsk_ab_env("DSK_FORCE_NW")) h.force_NW = atoi(dsk_ab_env("DSK_FORCE_NW"));  // tuning knob of tools/kbench.py
    for (int i = 0; i < n_tasks; ++i) {
      GemvTask& T = h.t[h.n_tasks++];
      const DTensor& t = W[(size_t)c * mats + i * (kind == 1 ? 2 : 1)].t;
      T.qs = t.qs; T.sc = t.sc; T.hm = t.hm; T.dm = t.dm; T.scale = t.scale; T.rows = rows; T.n = n; T.local_experts = 1;
      if (kind == 1) {
        const DTensor& t3 = W[(size_t)c * mats + i * 2 + 1].t;
        T.qs2 = t3.qs; T.sc2 = t3.sc; T.hm2 = t3.hm; T.dm2 = t3.dm; T.scale2 = t3.scale;
      }
      T.act_mode = act_mode;
      if (act_mode == ACT_Q8) { T.a_qs = aq; T.a_d = ad; T.a_bsums = ab; }
      T.a_f32 = x.as<float>() + (kind >= 2 ? (size_t)i * n : 0);
      T.norm_w = nw.as<float>(); T.eps = 1e-6f;
      T.out = out.as<float>() + (size_t)i * rows;
      T.accum_w = kind >= 2 ? wts.as<float>() + i : nullptr;
    }
    if (kind >= 2) { h.comb_x = xres.as<float>(); h.comb_counter = cnt.as<unsigned>(); }
    DSK_TRY(gemv_plan(h, target_wgs > 0 ? target_wgs : 1024));
  }
  DSK_TRY(plans.alloc(sizeof(GemvLaunch) * copies));
  HIP_TRY(hipMemcpyAsync(plans.p, H.data(), sizeof(GemvLaunch) * copies, hipMemcpyHostToDevice, st));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  for (int c = 0; c < copies; ++c) DSK_TRY(gemv_launch(st, plans.as<GemvLaunch>() + c, H[c]));  // warm-up
  HIP_TRY(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) DSK_TRY(gemv_launch(st, plans.as<GemvLaunch>() + (i % copies), H[i % copies]));
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *us_per_launch = (double)ms * 1e3 / iters;
  *bytes_per_launch = wbytes;
  if (dsk_ab_env("DSK_TIMELINE")) {  // one more launch with per-workgroup wall-clock stamps, summarised on stderr
    DevBuf tlb;
    const int grid = H[0].grid;
    DSK_TRY(tlb.alloc((size_t)grid * 64));
    HIP_TRY(hipMemsetAsync(tlb.p, 0, (size_t)grid * 64, st));
    H[0].timeline = tlb.as<unsigned long long>();
    HIP_TRY(hipMemcpyAsync(plans.p, H.data(), sizeof(GemvLaunch), hipMemcpyHostToDevice, st));
    DSK_TRY(gemv_launch(st, plans.as<GemvLaunch>() + 1, H[1]));  // a neighbour first, as in a stream of launches
    DSK_TRY(gemv_launch(st, plans.as<GemvLaunch>(), H[0]));
    std::vector<unsigned long long> tl((size_t)grid * 8);
    HIP_TRY(hipMemcpyAsync(tl.data(), tlb.p, (size_t)grid * 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    unsigned long long t00 = ~0ull;
    for (int i = 0; i < grid; ++i) if (tl[i * 8] && tl[i * 8] < t00) t00 = tl[i * 8];
    double mx[8] = {0}, av[8] = {0}, mn[8] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
    int cnt = 0;
    for (int i = 0; i < grid; ++i) {
      if (!tl[i * 8 + 3]) continue;
      ++cnt;
      for (int k = 0; k < 8; ++k) {
        const double v = tl[i * 8 + k] ? (double)(tl[i * 8 + k] - t00) * 0.01 : 0;  // us
        mx[k] = v > mx[k] ? v : mx[k]; mn[k] = v < mn[k] ? v : mn[k]; av[k] += v;
      }
    }
    fprintf(stderr, "timeline grid=%d (NW=%d R=%d U=%d lpr=%d) us since first entry [min avg max]: entry %.2f %.2f %.2f | staged %.2f %.2f %.2f | first group %.2f %.2f %.2f | exit %.2f %.2f %.2f\n",
            grid, H[0].NW, H[0].R, H[0].U, 1 << H[0].lpr_log2, mn[0], av[0] / cnt, mx[0], mn[1], av[1] / cnt, mx[1], mn[2], av[2] / cnt, mx[2], mn[3], av[3] / cnt, mx[3]);
    if (H[0].tiled)
      fprintf(stderr, "   tiled [min avg max]: descriptor %.2f %.2f %.2f | staged (wave 0) %.2f %.2f %.2f | round-1 items done (wave 0) %.2f %.2f %.2f | after its barrier %.2f %.2f %.2f | rows reduced %.2f %.2f %.2f\n",
              mn[6], av[6] / cnt, mx[6], mn[7], av[7] / cnt, mx[7], mn[2], av[2] / cnt, mx[2], mn[4], av[4] / cnt, mx[4], mn[5], av[5] / cnt, mx[5]);
    else
    fprintf(stderr, "   prologue avg: descriptor read %.2f | x loaded + sumsq %.2f | barrier + scale %.2f | quantised %.2f | all waves %.2f\n", av[6] / cnt, av[4] / cnt, av[5] / cnt, av[7] / cnt, av[1] / cnt);
  }
  return finish(ctx);
}

// Router + gate micro-benchmark (DeepSeek-V3: E = 256, dim = 7168).  flags: see RouterArgs::dbg.
extern "C" int dsk_bench_router(dsk_ctx* ctx, int n_routed, int dim, int ksplit, int flags, int iters, double* us_per_launch) {
  DSK_TRY(begin(ctx));
  if (!us_per_launch || n_routed < 1 || n_routed > 256 || dim < 4 || ksplit < 1 || iters < 1) DSK_FAIL(DSK_ERR_INVALID, "bench_router: bad argument");
  hipStream_t st = ctx_stream(ctx);
  const int copies = 48;
  DevBuf w, x, nw, partial, cnt, bias, ae, aw;
  DSK_TRY(w.alloc((size_t)copies * n_routed * dim * 4));
  DSK_TRY(x.alloc((size_t)dim * 4));
  DSK_TRY(nw.alloc((size_t)dim * 4));
  DSK_TRY(partial.alloc((size_t)ksplit * n_routed * 4));
  DSK_TRY(cnt.alloc(64));
  DSK_TRY(bias.alloc((size_t)n_routed * 4));
  DSK_TRY(ae.alloc(64));
  DSK_TRY(aw.alloc(64));
  DSK_TRY(launch_fill_f32(st, w.as<float>(), (size_t)copies * n_routed * dim, 3, 0.f, 0.02f));
  DSK_TRY(launch_fill_f32(st, x.as<float>(), (size_t)dim, 4, 0.f, 1.f));
  DSK_TRY(launch_fill_f32(st, nw.as<float>(), (size_t)dim, 5, 1.f, 0.1f));
  DSK_TRY(launch_fill_f32(st, bias.as<float>(), (size_t)n_routed, 6, 0.f, 0.01f));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, 64, st));
  RouterArgs r;
  memset(&r, 0, sizeof r);
  r.x = x.as<float>(); r.norm_w = nw.as<float>(); r.eps = 1e-6f; r.n_routed = n_routed; r.dim = dim; r.ksplit = ksplit;
  r.partial = partial.as<float>(); r.counter = cnt.as<unsigned>(); r.bias = bias.as<float>(); r.n_active = n_routed >= 8 ? 8 : 1;
  r.norm_topk_prob = 1; r.scoring = DSK_SCORE_SIGMOID; r.topk_method = n_routed % 8 == 0 && n_routed >= 64 ? DSK_TOPK_GROUP_LIMITED_GREEDY : DSK_TOPK_GREEDY;
  r.n_group = 8; r.topk_group = 4; r.scaling = 2.5f; r.active_experts = ae.as<int>(); r.active_weights = aw.as<float>(); r.dbg = flags;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  for (int i = 0; i < copies + iters; ++i) {
    if (i == copies) HIP_TRY(hipEventRecord(e0, st));
    r.w = w.as<float>() + (size_t)(i % copies) * n_routed * dim;
    DSK_TRY(launch_router_gate(st, r));
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *us_per_launch = (double)ms * 1e3 / iters;
  return finish(ctx);
}

// The router GEMV exactly as the model launches it (router_body: 16-wave workgroups, RW rows x 16/RW column slices,
// slices summed in slice order; rmsnorm recomputed per workgroup): raw logits before scoring (src/infer.cpp:847).
extern "C" int dsk_router_logits(dsk_ctx* ctx, const float* w, const float* x, const float* norm_w, float eps, int n_routed, int dim,
                                 float* logits) {
  DSK_TRY(begin(ctx));
  if (!w || !x || !logits || n_routed < 1 || n_routed > 256 || dim < 4 || dim % 4) DSK_FAIL(DSK_ERR_INVALID, "router_logits: bad argument");
  hipStream_t st = ctx_stream(ctx);
  int ksplit = 8;  // the model's choice (dsk_model_finalize): 8 column slices unless the rows are short
  while (ksplit > 1 && dim / ksplit < 256) ksplit /= 2;
  DevBuf dw, dx, dn, partial, cnt, ae, aw;
  DSK_TRY(dw.alloc((size_t)n_routed * dim * 4));
  DSK_TRY(dx.alloc((size_t)dim * 4));
  DSK_TRY(dn.alloc((size_t)dim * 4));
  DSK_TRY(partial.alloc((size_t)ksplit * n_routed * 4));
  DSK_TRY(cnt.alloc(64));
  DSK_TRY(ae.alloc(64));
  DSK_TRY(aw.alloc(64));
  HIP_TRY(hipMemcpyAsync(dw.p, w, (size_t)n_routed * dim * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)dim * 4, hipMemcpyHostToDevice, st));
  if (norm_w) HIP_TRY(hipMemcpyAsync(dn.p, norm_w, (size_t)dim * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, 64, st));
  RouterArgs r;
  memset(&r, 0, sizeof r);
  r.w = dw.as<float>(); r.x = dx.as<float>(); r.norm_w = norm_w ? dn.as<float>() : nullptr; r.eps = eps;
  r.n_routed = n_routed; r.dim = dim; r.ksplit = ksplit; r.partial = partial.as<float>(); r.counter = cnt.as<unsigned>();
  r.n_active = 1; r.norm_topk_prob = 0; r.scoring = DSK_SCORE_SIGMOID; r.topk_method = DSK_TOPK_GREEDY; r.n_group = 1; r.topk_group = 1;
  r.scaling = 1.f; r.active_experts = ae.as<int>(); r.active_weights = aw.as<float>();
  DSK_TRY(launch_router_gate(st, r));
  HIP_TRY(hipMemcpyAsync(logits, partial.p, (size_t)n_routed * 4, hipMemcpyDeviceToHost, st));
  return finish(ctx);
}
