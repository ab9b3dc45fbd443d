/*
 * dsk_oracle.h -- CPU oracle for the decode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the algorithm of andrewkchan/deepseek.cpp's
 * Model::_forward_cpu / Block::_block_cpu / quant vec_dot kernels, written from the
 * reference's behaviour (every function cites the reference file:line it follows).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (deepseek.cpp_amd/csrc) never links or calls it.
 *
 * Pinning: tests/test_oracle_pin.py checks this restatement against (a) the known-answer
 * vectors in the reference's src/test.cpp and (b) outputs of the reference itself compiled
 * from /root/reference into oracle/_ref (fixtures committed under tests/golden/).
 */
#ifndef DSK_ORACLE_H
#define DSK_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/dsk.h" /* dsk_config, dsk_quant, dsk_role: the boundary's POD types */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_model orc_model;

/* ---- scalar codecs (src/codec.h:22-57) ---- */
float orc_half_to_float(uint16_t h);
uint16_t orc_float_to_half(float f);          /* RNE, like _cvtss_sh(x, 0) */
float orc_f8e5m2_to_float(uint8_t b);
uint8_t orc_float_to_f8e5m2(float f);         /* truncating, src/codec.h:49-57 */

/* ---- ops ---- */
void orc_q8k_quantize(const float* x, int n, int8_t* qs, float* d, int16_t* bsums);
void orc_dequant_row(int quant, const void* row, int n, float* y);   /* Q2_K / Q3_K rows */
int orc_gemv(int quant, const void* w, const float* scale, const int32_t* block_size,
             int d, int n, const float* x, float* out);
/* K-quant GEMV on a given Q8_K vector (codes + block scales), for teacher-forced parity at a staging point */
int orc_gemv_q8(int quant, const void* w, int d, int n, const int8_t* qs, const float* yd, float* out);
/* Q2_K GEMV on a given Q8_K vector with the f32 association of the device's tiled kernels (csrc/tile_device.h) */
int orc_gemv_q2k_tiles(const void* w, int d, int n, const int8_t* qs, const float* yd, float* out);
int orc_gemv_expert(int quant, const void* w, const float* scale, const int32_t* block_size,
                    int expert, int d, int n, const float* x, float* out);
int orc_embed_row(int quant, const void* w, const float* scale, const int32_t* block_size,
                  int dim, int token, float* out);
void orc_rmsnorm(float* o, const float* x, const float* weight, int size, float eps);
void orc_softmax(float* o, const float* x, int size);
/* Sampler::sample / sample_argmax (src/sampler.cpp:28-75); coin = rand() / (float)RAND_MAX */
int orc_sample(const float* logits, int vocab_size, float temperature, float top_p, float coin);
float orc_sample_prob(const float* logits, int vocab_size, int index);  /* Sampler::sample_prob, src/sampler.cpp:12-26 */
void orc_moe_gate(const float* scores_in, const float* bias, int n_routed, int n_active,
                  int norm_topk_prob, float routed_scaling_factor, int scoring_func,
                  int topk_method, int n_group, int topk_group,
                  int32_t* active_experts, float* active_weights, float* scores_out);
void orc_rope(float* vec, int d, int head_dim, int pos, float theta, int is_v3);
void orc_rope_f16(uint16_t* vec, int d, int head_dim, int pos, float theta, int is_v3);
void orc_attn(float* xout, float* atth, const float* qh, const uint16_t* kh, const uint16_t* vh,
              int head_dim, int v_head_dim, int n_heads, int kv_len);
void orc_attn_mla(float* xout, float* atth, const float* qh_c, const float* qh_rope,
                  const uint16_t* ckv, const uint16_t* krope, int head_dim, int kv_lora_rank,
                  int rope_dim, int kv_len);

/* ---- model (same life-cycle as the dsk_* boundary; host pointers are BORROWED and must
 * outlive the model, like the reference's mmap'd QTensor views, src/codec.h:99,110) ---- */
int orc_model_create(const dsk_config* cfg, orc_model** out);
int orc_model_bind(orc_model* m, int role, int layer, int quant, const int32_t shape[4],
                   const void* host_ptr, size_t bytes);
int orc_model_finalize(orc_model* m);
int orc_model_destroy(orc_model* m);
int orc_forward(orc_model* m, int token, int pos, int mode, float* logits);
/* taps: routing of every layer in the last forward; x after each layer */
int orc_model_get_routing(orc_model* m, int32_t* experts, float* weights);
int orc_model_get_trace_x(orc_model* m, int layer, float* x_out);
/* raw router logits of `layer` in the last forward (n_routed floats, before scoring) */
int orc_model_get_router_logits(orc_model* m, int layer, float* out);
const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
