// ref_shim.cpp -- extern "C" window onto the UNMODIFIED reference, compiled from the sources
// where they lie under /root/reference (never copied).  TEST INFRASTRUCTURE ONLY.
//
// Built by oracle/Makefile into oracle/_ref/libdskref.so with the reference's own flags
// (Makefile:31-35: -O3 -ffast-math -std=c++20 -fopenmp -mf16c -mavx2 -mfma).  This TU
// textually includes the reference's src/infer.cpp so that its file-static kernels
// (matmul, moe_gate, rmsnorm, rope, ...) are reachable; the other reference TUs
// (quant.cpp, codec.cpp, model.cpp, profile.cpp, vendor/format.cc) are compiled as-is.
//
// Used (a) to pin oracle/dsk_oracle.c against the real reference and to generate
// tests/golden/*.npz (tools/make_golden.py), (b) as bench.py's cpu_baseline "reference".
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>
#include <iostream>
#include <sstream>
#include <omp.h>
#include "json.hpp"
#include "fmt/format.h"

// Model::_copy_embedding / _forward_cpu internals are private in the reference
// (src/model.h:474-478); the shim needs them to tap per-layer state.
#define private public
#define protected public
#include "model.h"
#undef private
#undef protected

#include "infer.cpp"  // the reference hot path itself (file-static functions included)
#include "sampler.h"

namespace {
struct RefSession {
  std::unique_ptr<YALMData> yalm;
  std::unique_ptr<Model> model;
  std::unique_ptr<InferenceState> state;
  std::vector<float> trace_x;     // n_layers * dim
  std::vector<int> route_e;       // n_layers * n_active
  std::vector<float> route_w;
  std::vector<float> gate_scores; // n_layers * n_routed (post scoring + bias)
};

QTensor make_qt(int quant, void* w, int d, int n, size_t bytes) {
  return QTensor(static_cast<Quant>(quant), {d, n, 0, 0}, w, bytes);
}
size_t wbytes(int quant, size_t numel) {
  switch (static_cast<Quant>(quant)) {
    case Quant::F32: return numel * 4;
    case Quant::F16: return numel * 2;
    case Quant::F8E5M2: return numel;
    case Quant::Q2_K: return numel / QK_K * sizeof(block_q2_K);
    case Quant::Q3_K: return numel / QK_K * sizeof(block_q3_K);
  }
  return 0;
}
}  // namespace

extern "C" {

void ref_set_threads(int n) { omp_set_num_threads(n); }
int ref_max_threads() { return omp_get_max_threads(); }

// quantize_row_q8_K_ref (src/quant.cpp:616-653) -> SoA
void ref_q8k_quantize(const float* x, int n, int8_t* qs, float* d, int16_t* bsums) {
  int nb = n / QK_K;
  std::vector<block_q8_K> blocks(nb);
  std::memset(blocks.data(), 0, sizeof(block_q8_K) * nb);
  quantize_row_q8_K_ref(x, blocks.data(), n);
  for (int i = 0; i < nb; i++) {
    std::memcpy(qs + (size_t)i * QK_K, blocks[i].qs, QK_K);
    d[i] = blocks[i].d;
    std::memcpy(bsums + (size_t)i * 16, blocks[i].bsums, 32);
  }
}

void ref_dequant_row(int quant, const void* row, int n, float* y) {
  if (static_cast<Quant>(quant) == Quant::Q2_K) dequantize_row_q2_K((const block_q2_K*)row, y, n);
  else dequantize_row_q3_K((const block_q3_K*)row, y, n);
}

// offline quantizers (src/quant.cpp:147,308): used only to build synthetic checkpoints
void ref_quantize_row(int quant, const float* x, void* y, int64_t k) {
  if (static_cast<Quant>(quant) == Quant::Q2_K) quantize_row_q2_K_ref(x, (block_q2_K*)y, k);
  else quantize_row_q3_K_ref(x, (block_q3_K*)y, k);
}

// matmul (src/infer.cpp:381-417)
void ref_gemv(int quant, void* w, float* scale, const int* block_size, int d, int n, float* x, float* out) {
  QTensor qt = make_qt(quant, w, d, n, wbytes(quant, (size_t)d * n));
  std::optional<QTensor> sc = std::nullopt;
  if (scale) {
    int sr = cdiv(d, block_size[0]), scn = cdiv(n, block_size[1]);
    sc = QTensor(Quant::F32, {sr, scn, 0, 0}, scale, (size_t)sr * scn * 4);
  }
  std::vector<uint8_t> aqb((size_t)std::max(1, n / QK_K) * sizeof(block_q8_K));
  matmul(out, x, qt, block_size, sc, aqb.data());
}

// matmul_expert (src/infer.cpp:423-469)
void ref_gemv_expert(int quant, void* w, float* scale, const int* block_size, int n_experts, int expert,
                     int d, int n, float* x, float* out) {
  QTensor qt(static_cast<Quant>(quant), {n_experts, d, n, 0}, w, wbytes(quant, (size_t)n_experts * d * n));
  std::optional<QTensor> sc = std::nullopt;
  if (scale) {
    int sr = cdiv(d, block_size[0]), scn = cdiv(n, block_size[1]);
    sc = QTensor(Quant::F32, {n_experts, sr, scn, 0}, scale, (size_t)n_experts * sr * scn * 4);
  }
  std::vector<uint8_t> aqb((size_t)std::max(1, n / QK_K) * sizeof(block_q8_K));
  matmul_expert(out, x, qt, expert, block_size, sc, aqb.data());
}

void ref_rmsnorm(float* o, float* x, float* weight, int size, float eps) { rmsnorm(o, x, weight, size, eps); }

// moe_gate (src/infer.cpp:493-599); scores is modified in place like the reference does
void ref_moe_gate(float* scores, float* bias, int n_routed, int n_active, int norm_topk_prob,
                  float routed_scaling_factor, int scoring_func, int topk_method, int n_group, int topk_group,
                  int* active_experts, float* active_weights) {
  // The reference reads x[-1] in the group-limited first pass (src/infer.cpp:558); give it the
  // 0.0f it finds in practice in front of a heap array.
  std::vector<float> buf(n_routed + 1, 0.0f);
  std::memcpy(buf.data() + 1, scores, sizeof(float) * n_routed);
  std::optional<QTensor> b = std::nullopt;
  if (bias) b = QTensor(Quant::F32, {n_routed, 0, 0, 0}, bias, (size_t)n_routed * 4);
  moe_gate(active_weights, b, active_experts, buf.data() + 1, n_routed, n_active, norm_topk_prob != 0,
           routed_scaling_factor, static_cast<ScoringFunc>(scoring_func), static_cast<TopKMethod>(topk_method),
           n_group, topk_group);
  std::memcpy(scores, buf.data() + 1, sizeof(float) * n_routed);
}

void ref_rope(float* vec, int d, int head_dim, int pos, float theta, int is_v3) {
  std::vector<float> buf(d);
  if (is_v3) rope_v3(vec, d, head_dim, pos, theta);
  else rope(buf.data(), vec, d, head_dim, pos, theta);
}
void ref_rope_f16(uint16_t* vec, int d, int head_dim, int pos, float theta, int is_v3) {
  std::vector<float> buf(d);
  if (is_v3) rope_v3(vec, d, head_dim, pos, theta);
  else rope(buf.data(), vec, d, head_dim, pos, theta);
}

void ref_attn(float* xout, float* atth, const float* qh, const uint16_t* kh, const uint16_t* vh, int head_dim,
              int v_head_dim, int n_heads, int kv_len) {
  attn(xout, atth, qh, kh, vh, head_dim, v_head_dim, n_heads, kv_len);
}
void ref_attn_mla(float* xout, float* atth, const float* qh_c, const float* qh_rope, const uint16_t* ckv,
                  const uint16_t* krope, int head_dim, int kv_lora_rank, int rope_dim, int kv_len) {
  attn_mla(xout, atth, qh_c, qh_rope, ckv, krope, head_dim, kv_lora_rank, rope_dim, kv_len);
}

uint16_t ref_float_to_half(float x) { return float_to_half(x); }
float ref_half_to_float(uint16_t x) { return half_to_float(x); }
uint8_t ref_float_to_f8e5m2(float x) { return float_to_float8e5m2(x); }
float ref_f8e5m2_to_float(uint8_t x) { return float8e5m2_to_float(x); }

// ---- whole model: YALMData + Model + InferenceState, like Session (src/main.cpp:71-83) ----
void* ref_session_create(const char* dir, int context) {
  auto* s = new RefSession();
  std::streambuf* old = std::cout.rdbuf();
  std::ostringstream sink;  // the loader is chatty on stdout (src/codec.cpp:263,366)
  std::cout.rdbuf(sink.rdbuf());
  s->yalm = std::make_unique<YALMData>(std::string(dir), false);
  s->model = std::make_unique<Model>(*s->yalm, context);
  s->state = std::make_unique<InferenceState>(s->model->config);
  std::cout.rdbuf(old);
  const Config& c = *s->model->config;
  s->trace_x.assign((size_t)c.n_layers * c.dim, 0.f);
  s->route_e.assign((size_t)c.n_layers * std::max(1, c.n_active_routed), -1);
  s->route_w.assign((size_t)c.n_layers * std::max(1, c.n_active_routed), 0.f);
  s->gate_scores.assign((size_t)c.n_layers * std::max(1, c.n_routed_experts), 0.f);
  return s;
}
void ref_session_destroy(void* h) { delete static_cast<RefSession*>(h); }

// Model::forward as shipped (src/model.cpp:874-883)
void ref_forward(void* h, int token, int pos, int mode, float* logits) {
  auto* s = static_cast<RefSession*>(h);
  s->model->forward(*s->state, token, pos, static_cast<InferenceMode>(mode));
  if (logits && static_cast<InferenceMode>(mode) == InferenceMode::OUTPUT_LOGITS)
    std::memcpy(logits, s->state->logits(), sizeof(float) * s->model->config->vocab_size);
}

// Same statements as Model::_forward_cpu (src/infer.cpp:1265-1317) with taps between blocks.
void ref_forward_traced(void* h, int token, int pos, int mode, float* logits) {
  auto* S = static_cast<RefSession*>(h);
  Model& m = *S->model;
  InferenceState& s = *S->state;
  const Config& c = *m.config;
  m._copy_embedding(s, token);
  int W = c.rs_original_max_position_embeddings;
  int kv_sink = pos >= W ? KV_SINKS : 0;
  int kv_pos = kv_sink + (pos - kv_sink) % (W - kv_sink);
  int kv_len = pos >= W ? W : pos + 1;
  int K = std::max(1, c.n_active_routed), E = std::max(1, c.n_routed_experts);
  for (size_t l = 0; l < m.blocks.size(); l++) {
    m.blocks[l]->block(s, pos, kv_sink, kv_pos, kv_len);
    std::memcpy(&S->trace_x[l * c.dim], s.x(), sizeof(float) * c.dim);
    bool is_moe = c.n_routed_experts > 0 && m.blocks[l]->moegate() != std::nullopt;
    for (int k = 0; k < K; k++) {
      S->route_e[l * K + k] = is_moe ? s.active_experts()[k] : -1;
      S->route_w[l * K + k] = is_moe ? s.active_experts_weights()[k] : 0.f;
    }
    if (is_moe) std::memcpy(&S->gate_scores[l * E], s.moe_weights(), sizeof(float) * c.n_routed_experts);
  }
  if (static_cast<InferenceMode>(mode) == InferenceMode::HYDRATE_KV_CACHE) return;
  rmsnorm(s.x(), s.x(), static_cast<float*>(m.rms_final_weight->data), c.dim, c.norm_eps);
  switch (c.weight_quant) {
    case Quant::F32:
    case Quant::F16: matmul_unscaled(s.logits(), s.x(), *m.wcls); break;
    default: matmul(s.logits(), s.x(), *m.wcls, c.block_size.data(), m.scls, s.aqb()); break;
  }
  if (logits) std::memcpy(logits, s.logits(), sizeof(float) * c.vocab_size);
}

void ref_get_routing(void* h, int* experts, float* weights) {
  auto* s = static_cast<RefSession*>(h);
  std::memcpy(experts, s->route_e.data(), s->route_e.size() * 4);
  std::memcpy(weights, s->route_w.data(), s->route_w.size() * 4);
}
void ref_get_trace_x(void* h, int layer, float* out) {
  auto* s = static_cast<RefSession*>(h);
  int dim = s->model->config->dim;
  std::memcpy(out, &s->trace_x[(size_t)layer * dim], sizeof(float) * dim);
}
void ref_get_gate_scores(void* h, int layer, float* out) {
  auto* s = static_cast<RefSession*>(h);
  int E = s->model->config->n_routed_experts;
  std::memcpy(out, &s->gate_scores[(size_t)layer * E], sizeof(float) * E);
}
// Sampler::sample as shipped (src/sampler.cpp:41-75) on a given logits vector.  The sampler seeds std::rand in its
// constructor and draws once per call: *coin_out is that draw (rand() / (float)RAND_MAX after srand(seed)).
int ref_sample(void* h, const float* logits, float temperature, float top_p, unsigned seed, float* coin_out) {
  auto* s = static_cast<RefSession*>(h);
  const int V = s->model->config->vocab_size;
  std::memcpy(s->state->logits(), logits, sizeof(float) * V);
  std::srand(seed);
  if (coin_out) *coin_out = std::rand() / (float)RAND_MAX;
  Sampler smp(s->model->config, seed);
  return smp.sample(*s->state, temperature, top_p);
}

// Sampler::sample_prob as shipped (src/sampler.cpp:12-26)
float ref_sample_prob(void* h, const float* logits, int index) {
  auto* s = static_cast<RefSession*>(h);
  std::memcpy(s->state->logits(), logits, sizeof(float) * s->model->config->vocab_size);
  Sampler smp(s->model->config, 0);
  return smp.sample_prob(index, *s->state);
}

double ref_active_bytes(void* h, int pos) { return static_cast<RefSession*>(h)->model->active_bytes(pos); }

}  // extern "C"

// Per-block wall time of one forward (same statements as ref_forward_traced): seconds[l] for every
// block, seconds[n_layers] for final norm + lm_head.  Used by bench.py's cpu_baseline leg to
// extrapolate a full-depth token from a reduced-depth checkpoint.
extern "C" void ref_forward_timed(void* h, int token, int pos, double* seconds) {
  auto* S = static_cast<RefSession*>(h);
  Model& m = *S->model;
  InferenceState& s = *S->state;
  const Config& c = *m.config;
  m._copy_embedding(s, token);
  int W = c.rs_original_max_position_embeddings;
  int kv_sink = pos >= W ? KV_SINKS : 0;
  int kv_pos = kv_sink + (pos - kv_sink) % (W - kv_sink);
  int kv_len = pos >= W ? W : pos + 1;
  for (size_t l = 0; l < m.blocks.size(); l++) {
    double t0 = omp_get_wtime();
    m.blocks[l]->block(s, pos, kv_sink, kv_pos, kv_len);
    seconds[l] = omp_get_wtime() - t0;
  }
  double t0 = omp_get_wtime();
  rmsnorm(s.x(), s.x(), static_cast<float*>(m.rms_final_weight->data), c.dim, c.norm_eps);
  switch (c.weight_quant) {
    case Quant::F32:
    case Quant::F16: matmul_unscaled(s.logits(), s.x(), *m.wcls); break;
    default: matmul(s.logits(), s.x(), *m.wcls, c.block_size.data(), m.scls, s.aqb()); break;
  }
  seconds[m.blocks.size()] = omp_get_wtime() - t0;
}
