// dsk_internal.h -- shared declarations of the gfx950 decode engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include "../../include/dsk.h"

#define QK_K 256
// A/B knobs of the measurement tools (tools/kbench.py, tools/ab_build.sh): read from the environment ONLY in -DDSK_AB
// builds.  The shipped library never looks at the environment: a stray variable must not change kernels or numerics
// (every model-level choice is a dsk_model_set_option key).
#ifdef DSK_AB
#include <stdlib.h>
static inline const char* dsk_ab_env(const char* k) { return getenv(k); }
#else
static inline const char* dsk_ab_env(const char*) { return nullptr; }
#endif
#define DSK_TL_WGS 1024  // workgroups per kind a timeline region holds (8 stamps each; engine.h timeline_of): larger grids stamp their first 1024

// ---- error plumbing (never abort across the boundary; include/dsk.h conventions) ----
void dsk_set_error(int code, const char* fmt, ...);
void dsk_clear_error();  // a recoverable planning failure (a fused form that does not apply) must not leave its message behind
#define DSK_FAIL(code, ...)            \
  do {                                 \
    dsk_set_error((code), __VA_ARGS__); \
    return (code);                     \
  } while (0)
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) DSK_FAIL(DSK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define DSK_TRY(expr)      \
  do {                     \
    int _r = (expr);       \
    if (_r != DSK_OK) return _r; \
  } while (0)

// ---- last-arriver hand-offs inside a launch (MoE combine, router gate, attention Q8_K finisher, split-context merge) ----
// Protocol: the payload goes out with write-through stores (__hip_atomic_store relaxed/agent = global_store sc1), EVERY
// storing wave drains its stores (asm s_waitcnt vmcnt(0)), __syncthreads, ONE lane arrives on a counter (relaxed agent
// fetch_add); the workgroup that sees the last count reads the payload with sc1 LOADS (__hip_atomic_load relaxed/agent),
// which are served by L2 / memory and never by the reading CU's L1.  That is the "{sc1 stores and sc1 loads on both
// sides}" form MI355X_MICROARCH.md lists as valid on gfx950 without any agent-scope fence: the drained write-through
// store is in memory before the arrival is issued, and the finisher's loads cannot hit a stale L1 line.  The agent
// acquire the finisher ALSO issues (buffer_inv sc1) is therefore belt and braces; it stays on because it is cheap where
// it sits (measured: the whole token 5.350 ms with it, 5.327 ms without, -DDSK_FINISHER_ACQUIRE=0) and makes the
// protocol a textbook release-by-drain / acquire pair.  tests/test_fused_moe_gpu.py replays every hand-off thousands
// of times from a captured graph and checks the bits.
#ifndef DSK_FINISHER_ACQUIRE
#define DSK_FINISHER_ACQUIRE 1
#endif
#if DSK_FINISHER_ACQUIRE
#define FINISHER_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define FINISHER_ACQUIRE() ((void)0)
#endif

// ---- device-side tensor views -------------------------------------------------
// K-quant matrices are re-laid-out at upload into byte planes (same total bytes as the
// reference's AoS blocks, src/quant.h:41-52,70-76), so that a wave reads each plane with
// fully coalesced 16-byte-per-lane loads:
//   Q2_K: qs  [row][blk][64]   the 2-bit quants, unchanged byte order
//         sc  [row][blk][16]   scale/min nibbles permuted to "quarter order":
//                              sc'[4*q + s] = scales[8*h + 2*s + lh], q = 2*h + lh
//         dm  [row][blk] u32   d (f16) | dmin (f16) << 16
//   Q3_K: qs  [row][blk][64], hm [row][blk][32], sc [row][blk][12] (packed 6-bit, unchanged),
//         dm  [row][blk] u16   d (f16)
// F8E5M2 / F16 / F32 matrices stay row-major as stored; F8 block scales stay (ceil(d/b0), ceil(n/b1)).
struct DTensor {
  int quant = DSK_QUANT_F32;
  int n_experts = 0;  // 0 = plain 2-D (rows, n); >0 = stacked (E, rows, n)
  int local_experts = 0, expert_base = 0;  // expert-sharded storage: experts [base, base+local)
  int rows = 0, n = 0;
  uint8_t* base = nullptr;  // one allocation
  size_t bytes = 0;         // device bytes of this tensor (weights + scales)
  // plane pointers (device)
  uint8_t* qs = nullptr;
  uint8_t* sc = nullptr;
  uint8_t* hm = nullptr;
  uint8_t* dm = nullptr;
  float* scale = nullptr;   // F8 block scales
  bool scale_bound = false; // the ".scale" tensor was actually uploaded / synthesised (alloc_tensor only reserves it)
  size_t e_qs = 0, e_sc = 0, e_hm = 0, e_dm = 0, e_scale = 0;  // per-expert strides in bytes (scale: floats)
  // Q2_K in the TILED layout (tile_device.h): qs = the tile records (rows padded to a multiple of 16 per matrix), e_qs = the
  // bytes of one padded matrix, sc / dm unused
  bool tiled = false;
  bool bound() const { return base != nullptr; }
};

// ---- GEMV launch descriptors (kernels_gemv.hip) ----------------------------------
// A launch descriptor lives in device memory (built once per layer at finalize time: every
// pointer in it is fixed for the life of the model, so a captured graph can replay it).
enum { EPI_STORE = 0, EPI_ADD = 1 };
enum { ACT_Q8 = 0, ACT_F32 = 1, ACT_F32_NORM = 2 };
#define GEMV_MAX_TASKS 12

struct GemvTask {
  // weights: K-quant planes, or qs = the row-major F8/F16/F32 matrix; second set = w3 of a GLU pair
  const uint8_t *qs, *sc, *hm, *dm;
  const float* scale;
  const uint8_t *qs2, *sc2, *hm2, *dm2;
  const float* scale2;
  size_t e_qs, e_sc, e_hm, e_dm, e_scale;  // expert strides (bytes; scale in floats); e_qs == 0: not a stack
  const int* expert_ids;                    // device: slot -> expert id (null: expert = slot)
  int slot, expert_base, local_experts;
  int rows, n;
  int wt_tiled;            // host-side: the weights are tile records (tile_device.h)
  // activation vector
  int act_mode;            // ACT_*
  const int8_t* a_qs;      // ACT_Q8: ready Q8_K vector
  const float* a_d;
  const int16_t* a_bsums;
  const float* a_f32;      // ACT_F32 / ACT_F32_NORM: f32 vector (quantised / normed in the prologue)
  const float* norm_w;     // rmsnorm weight
  float eps;
  // output
  float* out;
  int epilogue;            // EPI_*
  const float* accum_w;    // fused MoE combine: device pointer to this slot's mixing weight (null: 1, the shared expert)
  // Consecutive tasks reading the SAME activation vector form an activation group: they share one
  // workgroup range and their rows are concatenated into one virtual row space that the group's
  // workgroups split evenly (so 8 routed experts + the shared expert balance like one dense matrix)
  int wg_begin, wg_end;      // workgroup range of the task's activation group
  int vrow_begin, vrow_end;  // this task's rows inside the group's virtual row space
};

struct GemvLaunch {
  // activation-group table, first so that ONE scalar load brings it in (walking the tasks to find a
  // workgroup's group cost a dependent scalar load per task: 3 us in front of a 9-task launch)
  int n_groups;
  int grp_wg_end[GEMV_MAX_TASKS];   // exclusive workgroup end of group g
  int grp_t0[GEMV_MAX_TASKS + 1];   // first task of group g (grp_t0[n_groups] = n_tasks)
  int pad_[6];
  GemvTask t[GEMV_MAX_TASKS];
  int n_tasks;
  int quant, glu, act;  // act = DSK_ACT_* of the GLU epilogue
  int lpr_log2, R, U, grid;
  int NW;                     // waves per workgroup (4 or 16)
  int part_unit;              // rows are dealt to workgroups in multiples of this
  int force_lpr, force_R, force_U, force_NW;  // > 0: override the planner (micro-benchmarks)
  int ahead;             // the caller allows gemv_ahead_kernel (kernels_gemv.hip: weights requested ahead of the staging) where the plan fits it
  int reserve_wgs;       // 16-wave workgroups of the same launch that are NOT this plan's (the MLA latent's cache-write rider): the plan
                         // takes that many fewer, so that every workgroup of the launch is resident at once - a workgroup behind the
                         // 256th starts when the first one leaves (kvwrite: 13.2 us = the GEMV's 10.7 + the rider's 3.6 behind it)
  int fill_div;               // > 0: a group may take up to fill_div x rows / (rows per step) workgroups (default 2: half-filled)
  int b0, b1;                 // F8 block-scale geometry
  // block-diagonal stack (MLA per-head wv_b, src/infer.cpp:1134-1137): t[0] describes head 0 of
  // `bd_heads` equal (rows, n) matrices stacked along rows; head h reads activation a_f32 + h*n and
  // writes out + h*rows; bd_wgs workgroups per head
  int bd_heads, bd_wgs;
  // Q2_K weights in the tiled layout (tile_device.h; kernels_tile.hip): qs / qs2 point at tile records, e_qs = bytes of one
  // padded matrix of a stack, vrow_* count rows padded to 16 per task, part_unit is a multiple of 16; t_act = bytes of the
  // staged activation vector in LDS, t_rcap = items a round's partials hold (behind it)
  int tiled, t_act, t_rcap;
  // MoE combine folded into a TASKS launch (tasks = routed slots in k order, then the shared expert;
  // every task writes its own vector t[i].out): the LAST task to finish a row group (arrival counter
  // per group) runs  x[row] += w_k * out_k[row]  in k order, then + shared (src/infer.cpp:874-877,900-903)
  float* comb_x;
  unsigned* comb_counter;
  int compact_absent; // expert-sharded w1/w3 launch (one activation group): the group's virtual row space is compacted ON THE
                      // DEVICE over the tasks whose expert lives on this GPU, so that all workgroups share the present rows
                      // (with the static row space an absent expert's workgroups exit and the launch takes as long as on one GPU)
  int zero_absent;    // expert-sharded W2 launch: a task whose expert lives on another GPU stores zeros (the sum
                      // all-reduce then needs no separate zero-fill of the slot buffer)
  int comb_geometry;  // plan exactly like a fused-combine launch (the expert-sharded W2 launch: same
                      // summation trees as on one GPU => bit-identical slot outputs)
  // debug (DSK_TIMELINE=1 with dsk_bench_gemv): 4 wall-clock stamps per workgroup (entry, staged, first
  // row group done, exit), 100 MHz s_memrealtime ticks
  unsigned long long* timeline;
  // parity taps (dsk_model_run_block): when set, the first workgroup of activation group g copies the Q8_K vector it
  // staged (int8 codes, block scales) to tap_qs + g * tap_stride / tap_d + g * tap_stride / 256.  Null in production.
  int8_t* tap_qs;
  float* tap_d;
  int tap_stride;
  size_t lds_bytes;
  double algo_bytes;          // host-side bookkeeping for the roofline report
};

// dsk_profile_forward: when both are set, the next GEMV launch records them as the kernel's OWN start / stop
// (hipExtLaunchKernel: dispatch timestamps, no extra barrier packets), i.e. the duration rocprofv3 reports
extern thread_local hipEvent_t g_prof_start, g_prof_stop;
int gemv_plan(GemvLaunch& h, int target_wgs);
int gemv_launch(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& host);
// the same for Q2_K weights in the tiled layout (kernels_tile.hip; gemv_plan / gemv_launch forward to them when h.tiled)
int gemv_plan_tile(GemvLaunch& h, int target_wgs);
int gemv_launch_tile(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& host);
size_t tile_mat_bytes(size_t rows, size_t n);  // bytes of one (rows, n) Q2_K matrix in the tiled layout (rows padded to 16)

// The routed experts of one MoE block in ONE launch (kernels_moe.hip): w1/w3 GLU units, per-slot hand-off, W2 units,
// k-ordered combine.  Planes of the expert stacks + per-expert strides; the shared expert's W2 as a plain matrix.
#ifndef MOE_CTR_STRIDE
#define MOE_CTR_STRIDE 64  // words between the slot counters of the fused expert launch: a 256-byte line each (all 8 in one
                          // line: 35.6 us per launch, apart: 35.2)
#endif
#define MOE_CTR_WORDS (16 * MOE_CTR_STRIDE + 16)
#define MOE_GAVE_UP_WORD (16 * MOE_CTR_STRIDE + 8)  // device-side copy of "a hand-off gave up during this token" (cleared with the counters)
#define MOE_CAND_BYTES 65536  // the pipelined launch's candidate records: 16 strips x (|max|, max) per 256-block of a hidden vector
#define MOE_BLK_CTRS 1024  // per-block arrival counters of the fused expert launch (K x mi / 256 <= 1024)
struct MoeFfnArgs {
  int quant;
  const uint8_t *w1_qs, *w1_sc, *w1_hm, *w1_dm, *w3_qs, *w3_sc, *w3_hm, *w3_dm;
  const uint8_t *w2_qs, *w2_sc, *w2_hm, *w2_dm;
  size_t e13_qs, e13_sc, e13_hm, e13_dm, e2_qs, e2_sc, e2_hm, e2_dm;
  const uint8_t *sw2_qs, *sw2_sc, *sw2_hm, *sw2_dm;  // shared expert's W2 (dim, shared_n); unused when shared_n == 0
  int shared_n;
  const int* route_e;    // this layer's K selected experts / mixing weights (device, written by the router launch)
  const float* route_w;
  int K, mi, dim, act;
  const int8_t* a_qs;    // Q8_K of rmsnorm(x, ffn_norm): left behind by the router launch
  const float* a_d;
  const int16_t* a_bsums;
  float* hb;             // hidden vectors [slot][hb_stride] (slot K = the shared expert's, written by the router launch)
  int hb_stride;
  float* eout;           // slot outputs [slot][dim]
  float* x;              // residual stream
  int n_experts;         // experts in the stacks (offset range check)
  unsigned* slot_ctr;    // [K x MOE_CTR_STRIDE] phase-A arrivals per slot; zeroed by the router launch of the same block
  // the hidden vectors are handed over ALREADY QUANTISED (hq != null): the last phase-A unit of a 256-block of h_k to
  // arrive (blk_ctr, re-armed by that unit) quantises the block (quantize_row_q8_K_ref) and only then arrives on the slot's
  // counter, which therefore counts blocks; phase B copies 292 bytes per block instead of reading 1 KB and quantising it
  // in each of the 256 workgroups
  int8_t* hq_qs;         // [slot][hb_stride]
  float* hq_d;           // [slot][hb_stride / 256]
  int16_t* hq_bsums;     // [slot][hb_stride / 16]
  unsigned* blk_ctr;     // [K x (mi / 256)] arrivals per block; zero between launches
  unsigned* err;         // host-visible: set when a bounded spin gives up
  // tail prefetch (option "tail_prefetch"): pf_wgs extra workgroups behind the grid (one per XCD: workgroup b runs on XCD
  // b % 8; they start when the grid's first workgroups have left) READ the cold lines the NEXT launch of the token opens
  // with - the next block's attention-norm weights and the descriptor of its first-stage projections - into their XCD's L2:
  // first-stage projections 7.12 -> 6.77 us, this launch unchanged.  (The same behind wo for the router launch and behind
  // the dense w2: no gain for the consumer, +0.7 us for the producer - its tail is short.)
  int pf_wgs;
  const void* pf_p[6];
  int pf_n[6];
  int lprA_log2, lprB_log2;  // lanes per row of the two halves (= the two-launch plans': bit-identical results)
  int UA, rows_wg, lds_a, lds_b, lds_o, grid;  // filled by moe_ffn_plan
  int tiled, lds_red;    // Q2_K weights in the tiled layout (kernels_moe_tile.hip: w*_qs = tile records, e13_qs / e2_qs = bytes of one
                         // padded expert matrix); lds_red = bytes of the partials region
  float* cand;           // the pipelined form's candidate records (MOE_CAND_BYTES)
  int pipe;              // option "moe_pipe": the launch pipelined by slot halves where its deals exist (kernels_moe_pipe.hip)
  int spin_limit;        // polls before the hand-off wait gives up (0: 2^20); < 0: fault injection (workgroup 0 reports a give-up)
  // float-weight models (F8E5M2 / F16 / F32; moe_ffn_f_kernel): block scales (F8 only, per-expert strides in floats), the
  // shared expert's w1 / w3 (computed in phase A too: these models have no rider in the router launch), the FFN norm (x is
  // normalised in the prologue; K-quant models get the router launch's Q8_K copy instead)
  const float *w1_scale, *w3_scale, *w2_scale, *sw1_scale, *sw3_scale, *sw2_scale;
  size_t e13_scale, e2_scale;
  const uint8_t *sw1_qs, *sw3_qs;
  const float* norm_w;
  float eps;
  int b0, b1;
  int US;                // phase-A units of the shared expert (float-weight kernel)
  unsigned long long* timeline;  // debug (DSK_MOE_TIMELINE=1): 8 wall-clock stamps per workgroup (100 MHz ticks)
  int8_t* tap_qs;        // parity taps (dsk_model_run_block): slot s's staged hidden vector at s * tap_stride
  float* tap_d;
  int tap_stride;
  double algo_bytes;
};
int moe_ffn_plan(MoeFfnArgs& a, int n_cus);
int launch_moe_ffn(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop);
int moe_ffn_plan_tile(MoeFfnArgs& a, int n_cus);
bool moe_pipe_applies(const MoeFfnArgs& a);  // kernels_moe_pipe.hip: the same launch pipelined by slot halves (option "moe_pipe")
int launch_moe_ffn_pipe(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop);
int launch_moe_ffn_tile(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop);

// per-token parameters living in device memory so that a captured graph can be replayed
struct StepParams {
  int token, pos, kv_sink, kv_pos, kv_len;
  float temperature, top_p, coin;  // dsk_forward_sample (Sampler::sample's arguments + its one random draw)
  int prob_index;                  // >= 0: dsk_forward_prob (Sampler::sample_prob of this index) instead of sampling
  float rope_cs[2 * 64];   // cos,sin for pair j at `pos` (host libm: powf/cosf/sinf, src/infer.cpp:655-658)
  float rope_cs1[2 * 64];  // same for pos = 1 (attention-sink rotation, src/infer.cpp:1015)
};

struct NormJob {         // one workgroup of norm_q8_kernel
  const float* x;        // input vector
  float* x_store;        // if non-null: write the (combined) input back here
  const float* weight;   // rmsnorm weight (null: no norm, y = x)
  int n;                 // length (multiple of 256 if q8 requested)
  float eps;
  float* y_f32;          // optional f32 output
  int8_t* q_qs;          // optional q8 output
  float* q_d;
  int16_t* q_bsums;
  // optional MoE combine folded in front (src/infer.cpp:874-877,900-903):
  //   x += sum_k w_k * eout[k]  (k order)  then  x += eout[n_add-1] if add_shared
  const float* eout;     // [n_slots][n]
  const float* eweights; // [k]
  int n_routed_slots;
  int add_shared;        // slot index n_routed_slots holds the shared expert output
  // optional rope on a tail vector (k_rope), V2 or V3 style
  float* rope_vec;
  int rope_d;
  int rope_v3;
};

// ---- launchers (kernels_misc.hip) --------------------------------------------------
// x[i] += sum_k w[k] * eout[k][i] (k order), then + eout[n_slots][i] if add_shared (src/infer.cpp:874-877, 900-903)
int launch_moe_combine(hipStream_t st, float* x, const float* eout, const float* weights, int n_slots, int add_shared, int n);
int launch_moe_combine_gathered(hipStream_t st, float* x, const float* gathered, const float* eout, const int* experts,
                                const float* weights, int n_slots, int add_shared, int n, int per);
int launch_quantize_q8k(hipStream_t st, const float* x, int n, int8_t* qs, float* d, int16_t* bsums);
int launch_norm_jobs(hipStream_t st, const NormJob* jobs, int n_jobs, const StepParams* sp);
int launch_repack_q2k(hipStream_t st, const uint8_t* aos, size_t n_blocks, uint8_t* qs, uint8_t* sc, uint8_t* dm);
int launch_repack_q2k_tiles(hipStream_t st, const uint8_t* aos, size_t gb0, size_t n_blocks, int rows, int nb, size_t e_bytes, uint8_t* tiles);
int launch_planes_to_tiles_q2k(hipStream_t st, const uint8_t* qs, const uint8_t* sc, const uint8_t* dm, size_t gb0, size_t n_blocks, int rows, int nb,
                               size_t e_bytes, uint8_t* tiles);
int launch_repack_q3k(hipStream_t st, const uint8_t* aos, size_t n_blocks, uint8_t* qs, uint8_t* hm, uint8_t* sc, uint8_t* dm);
int launch_embed(hipStream_t st, const DTensor& t, const StepParams* sp, int token_override, int b0, int b1, float* x);
int launch_embed_rows(hipStream_t st, const DTensor& t, const StepParams* sps, int P, int b0, int b1, float* x);  // P rows, tokens from the step rows
struct RouterArgs {       // F32 router GEMV (+ optional rmsnorm prologue) + moe_gate in the last workgroup
  const float* w;         // (E, dim)
  const float* x;         // input vector (f32)
  const float* norm_w;    // if non-null: x is normalised first (rmsnorm with eps)
  float eps;
  int n_routed, dim, ksplit;
  float* partial;         // (ksplit, E)
  unsigned* counter;      // arrival counter (zeroed by the kernel that consumes it)
  const float* bias;
  int n_active, norm_topk_prob, scoring, topk_method, n_group, topk_group;
  float scaling;
  int* active_experts;
  float* active_weights;
  float* scores_out;
  // optional: the Q8_K quantisation of the normed x (what the experts' w1/w3 GEMV consumes), written by the
  // first dim/256 workgroups -- every workgroup knows the norm scale anyway, so this costs one wave a block
  int8_t* q_qs;
  float* q_d;
  int16_t* q_bsums;
  int dbg;                // micro-benchmark only: 1 = skip the gate, 2 = skip the weight stream, 4 = skip the norm
  // arrival counters of a LATER launch of the same block that the gate workgroup zeroes (the fused expert launch's slot
  // counters, kernels_moe.hip): their previous users finished before this launch started (stream order)
  unsigned* zero_ctr;
  int zero_n;
  unsigned long long* timeline;  // debug (DSK_TIMELINE=1): 8 stamps per router workgroup
};
int launch_router_gate(hipStream_t st, const RouterArgs& a);
struct GemvLaunch;
// the router launch that also runs the shared expert's w1/w3 GLU (kernels_gemv.hip router_shared_kernel)
bool router_shared_supported(const RouterArgs& a, const GemvLaunch& h);
int launch_router_shared(hipStream_t st, const RouterArgs& a, const GemvLaunch* dev, const GemvLaunch& h);
int launch_gate(hipStream_t st, const float* partial, int ksplit, const float* bias, int n_routed, int n_active,
                int norm_topk_prob, float scaling, int scoring, int topk_method, int n_group, int topk_group,
                int* active_experts, float* active_weights, float* scores_out);
struct AttnMhaArgs {
  float* q;             // (H, head_dim) f32, rope applied in place by rope_kv
  const float* kv_b;    // (H, nope + v)
  const float* kv_a;    // (lora + rope): k_rope = kv_a + lora (un-rotated)
  uint16_t* key_cache;  // (seq, H*head_dim) f16
  uint16_t* value_cache;
  float* out;           // (H, v)
  int n_heads, head_dim, nope, rope, v_dim, lora, is_v3;
  // fused launch only: Q8_K copy of `out` for the wo GEMV (null: none); one arrival counter per 256-block
  int8_t* q_qs;
  float* q_d;
  int16_t* q_bsums;
  unsigned* q_counter;
};
int launch_rope_kv_mha(hipStream_t st, const AttnMhaArgs& a, const StepParams* sp);
// Per-head fusion of the second-stage projections with attention (kernels_gemv.hip head_attn_kernel):
// workgroup h computes head h's rows of wq_b (optional) and wkv_b into LDS, then runs the attention step.
struct HeadAttnArgs {
  GemvTask tq, tkv;      // rows = the FULL matrices; the kernel takes rows [h * rows_per_head, +rows_per_head)
  int has_q;             // 0: q was produced by the first-stage launch (q_lora_rank == 0), read a.q
  int quant, b0, b1;
  int lq_log2, lkv_log2; // lanes per row of the two projections
  int lds_q, lds_kv;     // bytes of the two staged activation vectors
  int tiled;             // Q2_K projections stored as tile records (tile_device.h): the head's strips on the matrix pipe
  int red_off, red_bytes; // tiled: the partials [step][64] live behind the scores (red_off floats behind lds_q + lds_kv)
  AttnMhaArgs a;
  // long contexts: n_split workgroups per head, each over a contiguous share of the cached positions (both
  // redo the head's projections: the other CUs would idle anyway); partial (O, m, l) per (head, split) go to
  // split_part, the LAST split of a head to arrive (split_counter[h]) merges them and finishes the head
  unsigned long long* timeline;  // debug (DSK_TIMELINE=1): 8 stamps per workgroup
  int n_split;
  float* split_part;        // (H, n_split, v_dim + 2)
  unsigned* split_counter;  // (H), zero between launches
  // parity taps (dsk_model_run_block): workgroup 0 copies the staged Q8_K vectors of norm(q_a) (at 0) and
  // norm(kv_a[:lora]) (at tap_stride); null in production
  int8_t* tap_qs;
  float* tap_d;
  int tap_stride;
  // prefetch workgroups behind the heads' (MoeFfnArgs::pf_wgs; here they run at once on CUs this launch leaves idle): the
  // cold lines wo opens with
  int pf_wgs;
  const void* pf_p[6];
  int pf_n[6];
};
#define MHA_SPLIT_MIN_KV 1024  // below this one workgroup per head is faster (default; DSK_MHA_SPLIT_MIN overrides)
#define MHA_SPLIT_MAX 16
int head_attn_plan(HeadAttnArgs& A);
int launch_head_attn(hipStream_t st, const HeadAttnArgs& A, const StepParams* sp, int max_kv, int n_split);
int launch_attn_mha(hipStream_t st, const AttnMhaArgs& a, const StepParams* sp, int kv_len_override, int max_kv);
struct AttnMlaArgs {
  float* q_rope;         // (H, rope)
  const float* q_c;      // (H, lora)
  const float* kv_a;     // (lora + rope), latent part already normed
  uint16_t* nope_cache;  // (seq, lora)
  uint16_t* rope_cache;  // (seq, rope)
  float* out;            // (H, lora)
  int n_heads, head_dim, rope, lora, is_v3;
};
int launch_rope_kv_mla(hipStream_t st, const AttnMlaArgs& a, const StepParams* sp);
// MLA attention for LONG contexts on the matrix cores (kernels_misc.hip mla_flash_kernel): all heads share one latent
// cache, so scores = Q[H x 576] . C^T and out = P . C are GEMMs.  Workgroup (chunk, head group of 32) walks its
// chunk of positions 32 at a time on v_mfma_f32_32x32x8_f16 (the cache entries are f16; q and the softmax weights go in
// as hi + lo f16 halves: exact products, f32 accumulation) with an online softmax, and leaves (m, l, O) partials; they are merged per head by mla_head_kernel / mla_merge_kernel.
#define MLA_FLASH_MIN_KV 320  // below this the per-head kernel's own attention is faster (token, ms: 6.12 / 6.20 / 6.31 at 256 /
                             // 320 / 384 against a flat 6.22-6.24 with this path; DSK_MLA_FLASH_MIN overrides)
struct MlaFlashArgs {
  const float* q_c;          // (H, lora)
  const float* q_rope;       // (H, rope)
  int rotate_q;              // 1: q_rope is un-rotated, apply sp->rope_cs (model path)
  const uint16_t* nope_cache;
  const uint16_t* rope_cache;
  float* part_o;             // (n_chunks, H, lora)
  float* part_ml;            // (n_chunks, H, 2): running max, running sum
  int n_heads, head_dim, lora, rope, is_v3;
  unsigned long long* timeline;  // debug (DSK_TIMELINE=1): 8 stamps per workgroup (chunk x head group)
  int chunk_len, n_chunks;   // positions per chunk (multiple of 32; 0: derived from the step's kv_len, MLA_FL_CHUNK), chunks in the grid
  // the batched prompt path (kernels_hydrate.hip): blockIdx.z = token of the chunk - its own step row (sp + z), q rows (z * the
  // strides below, in floats) and partials (z * n_chunks * H * lora / * 2); tokens whose context is shorter than min_kv are skipped
  // (they take the per-head kernel's own attention).  All zero in the decode launches (grid z = 1).
  int tok_qc_stride, tok_qr_stride, min_kv;
};
// positions per chunk for a context of kv_len: the grid (n_chunks x head groups) is fixed in the captured graph, the
// share of each chunk follows the context, so that a 1024-position context occupies 32 chunks of 32 positions instead
// of 11 chunks of 96 (mla_flash_kernel at kv_len 1024: 31 -> 13 us)
#define MLA_FL_CHUNK(kv_len, n_chunks) ((((kv_len) + (n_chunks) - 1) / (n_chunks) + 31) / 32 * 32)
int launch_mla_flash(hipStream_t st, const MlaFlashArgs& a, const StepParams* sp, int kv_len_override, int n_tokens = 1);
int launch_mla_merge(hipStream_t st, const MlaFlashArgs& a, const StepParams* sp, int kv_len_override, float* out);
// MLA, model path: (1) one workgroup normalises the latent, writes this position's cache entries and rotates the
// sink keys; (2) one 16-wave workgroup per head: RoPE of q_rope, attention over the shared latent cache, the head's
// wv_b rows, Q8_K of the concatenated outputs (kernels_gemv.hip mla_head_kernel).
struct MlaKvArgs {
  const float* kv_a;      // (lora + rope) raw output of wkv_a
  const float* norm_w;    // kv_a_norm
  float eps;
  uint16_t *nope_cache, *rope_cache;
  int lora, rope, is_v3;
};
int launch_mla_kv_write(hipStream_t st, const MlaKvArgs& a, const StepParams* sp);
// the second-stage projection launch of the MLA path with the cache write as its last workgroup (kernels_gemv.hip)
bool gemv_kvwrite_supported(const GemvLaunch& h);
int gemv_ahead_kind(const GemvLaunch& h, bool kvwrite);  // 0 none, 1 first-stage, 2 wo, 3 the MLA second stage (kernels_gemv.hip, round 6)
int launch_gemv_kvwrite(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h, const MlaKvArgs& kv, const StepParams* sp);
struct MlaHeadArgs {
  AttnMlaArgs a;          // q_rope (un-rotated), q_c, caches; out unused
  GemvTask twv;           // the (H * v_head_dim, lora) stack; the kernel takes rows [h * v, +v)
  int quant, b0, b1, lpr_log2, lds_act;
  int tiled;              // Q2_K wv_b stored as tile records: the head's strips on the matrix pipe (tile_device.h)
  unsigned long long* timeline;  // debug (DSK_TIMELINE=1): 8 stamps per workgroup
  AttnMhaArgs fin;        // out (H, v), v_dim, n_heads, Q8_K outputs + counter (only these fields are used)
  // long contexts: kv_len >= flash_thresh (> 0) => the attention part is the merge of mla_flash_kernel's partials
  int flash_thresh, fl_chunk_len, fl_n_chunks;
  const float* fl_part_o;
  const float* fl_part_ml;
  // parity taps (dsk_model_run_block): head h copies its latent output (lora floats) to tap_o + h * lora and the
  // Q8_K vector it staged for wv_b to tap_qs + h * lora / tap_d + h * lora / 256; null in production
  int8_t* tap_qs;
  float* tap_d;
  float* tap_o;
  int pf_wgs;             // prefetch workgroups behind the heads' (HeadAttnArgs::pf_wgs)
  const void* pf_p[6];
  int pf_n[6];
};
int mla_head_plan(MlaHeadArgs& A);
int launch_mla_head(hipStream_t st, const MlaHeadArgs& A, const StepParams* sp, int max_kv);
int launch_attn_mla(hipStream_t st, const AttnMlaArgs& a, const StepParams* sp, int kv_len_override, int max_kv);
int launch_rope_only(hipStream_t st, float* vec, int n_heads, int d, const float* cs, int is_v3);
int launch_fill_tensor(hipStream_t st, const DTensor& t, uint64_t seed, float wscale);
int launch_fill_f32(hipStream_t st, float* p, size_t n, uint64_t seed, float mean, float std);
int launch_read_bw(hipStream_t st, const void* p, size_t bytes, float* sink);
int launch_argmax(hipStream_t st, const float* x, int n, int* out);  // first maximum (strict >), src/sampler.cpp:28-39
// Sampler::sample (src/sampler.cpp:41-75) for temperature != 0; sp != nullptr: temperature / top_p / coin come from *sp;
// scratch: sample_scratch_bytes() of device memory, ZEROED once (the kernel re-arms its counter)
size_t sample_scratch_bytes();
int launch_sample(hipStream_t st, const float* logits, int n, const StepParams* sp, float temperature, float top_p, float coin, float* scratch, int* out);

// ---- batched prompt ingestion (kernels_hydrate.hip, hydrate.cpp) ------------------------------
// One i8-MFMA GEMM: out[v][row] (+)= W[row] . A[v / a_div] for the entries v of every task's list.  W / W3: Q2_K tile records
// (tile_device.h) of a plain matrix (n_experts 0) or an expert stack (task e = expert e, e_bytes apart); A: Q8_K rows in the
// linear struct-of-arrays form (codes [a_rows][n], scales [a_rows][n / 256], sub-block sums [a_rows][n / 16]).
struct HydGemmArgs {
  const uint8_t* W;
  const uint8_t* W3;      // non-null: GLU pair, out = act(W . a) * (W3 . a)
  size_t e_bytes;
  int n_experts;
  int rows, n;
  const int8_t* a_qs;
  const float* a_d;
  const int16_t* a_bsums;
  int a_rows, a_div;
  const int* list;        // per task: entries (null: 0 .. m - 1); count (null: m for the one task)
  const int* count;
  int list_stride, m;
  int cnt_min, cnt_max;   // only tasks whose row count lies in [cnt_min, cnt_max] are processed (cnt_max 0: no upper bound)
  float* out;
  int out_stride;
  int epilogue, act;      // EPI_STORE / EPI_ADD (ignored for GLU pairs); DSK_ACT_*
  const unsigned* a_dig;  // optional: the sub-block sums as digit words (launch_hyd_digits): the quads' min term then rides the matrix pipe
};
int launch_hyd_gemm(hipStream_t st, const HydGemmArgs& A, int nq);
int launch_hyd_digits(hipStream_t st, const int16_t* bsums, unsigned* dig, size_t count);
int launch_hyd_norm_q8(hipStream_t st, int NW, const float* X, int P, int n, const float* norm_w, float eps, int8_t* qs, float* d, int16_t* bsums);
struct HydLatentArgs {
  const float *q_a, *kv_a, *q_norm, *kv_norm;
  int q_stride, kv_stride, nq, nkv;
  float eps;
  int8_t *qq_qs, *kq_qs;
  float *qq_d, *kq_d;
  int16_t *qq_bsums, *kq_bsums;
};
int launch_hyd_latent_q8(hipStream_t st, const HydLatentArgs& A, int P);
int launch_hyd_kv_write(hipStream_t st, const AttnMhaArgs& a, const StepParams* sps, int P, const float* kv_b, int kvb_stride, const float* kv_a, int kva_stride);
int launch_hyd_attn(hipStream_t st, const AttnMhaArgs& a, const StepParams* sps, int P, int max_kv, const float* q, int q_stride, float* out, int out_stride,
                    int n_split, int split_min);
int launch_hyd_mla_kv_write(hipStream_t st, const MlaKvArgs& kv, const StepParams* sps, int P, int kva_stride);
int launch_hyd_mla_merge(hipStream_t st, const MlaFlashArgs& f, const StepParams* sps, int n_tokens, float* latent, int lat_stride);
int launch_hyd_mla_attn(hipStream_t st, const AttnMlaArgs& a, const StepParams* sps, int P, int max_kv, const float* q_c, int qc_stride, const float* q_rope,
                        int qr_stride, float* latent, int lat_stride);
int launch_hyd_head_list(hipStream_t st, int* list, int* count, int H, int P, int stride);
int launch_hyd_router(hipStream_t st, const RouterArgs& a, int P, float* Y);
int launch_hyd_route_override(hipStream_t st, int* route_e, int P, int K, int E, unsigned seed);
int launch_hyd_group(hipStream_t st, const int* route_e, int pairs, int E, int* list, int list_stride, int* count);
int launch_hyd_combine(hipStream_t st, float* X, const float* eout, const float* w, const float* eout_sh, int P, int K, int n);

// ---- engine.cpp helpers shared with ops_api.cpp ----------------------------------
struct dsk_ctx;
int ctx_scratch(dsk_ctx* c, int slot, size_t bytes, void** out);
hipStream_t ctx_stream(dsk_ctx* c);
int ctx_device(dsk_ctx* c);
bool is_kq(int q);
size_t mat_bytes(int quant, size_t rows, size_t n);
int alloc_tensor(int b0, int b1, DTensor& t, int quant, int e, int rows, int n, int local, int base);
int upload_tensor(dsk_ctx* ctx, DTensor& t, const void* host_ptr);
