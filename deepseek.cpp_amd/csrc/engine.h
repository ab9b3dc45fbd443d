// engine.h -- host-side data structures of the decode engine (shared by engine.cpp / forward.cpp).
#pragma once
#include "dsk_internal.h"

#include <map>
#include <string>
#include <vector>

#include <rccl/rccl.h>

// Where a tensor's bytes come from: host memory (dsk_model_bind) or a byte range of an open file (the .dseek
// loader, loader.cpp).  Either way they travel through the context's pinned staging ring (engine.cpp stage_copy).
struct HostSrc {
  const void* ptr = nullptr;
  int fd = -1;
  uint64_t off = 0;  // file offset of byte 0 (fd >= 0)
};

struct dsk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int n_cus = 256;       // compute units of the device (resident-grid kernels size themselves by it)
  int live_models = 0;   // models created on this context and not yet destroyed
  bool closing = false;  // dsk_ctx_destroy was called while models were alive: freed by the last dsk_model_destroy
  // scratch for op-level entry points
  void* op_buf[8] = {nullptr};
  size_t op_cap[8] = {0};
  // host -> HBM staging ring: STAGE_BUFS pinned buffers filled by reader threads (memcpy / pread) while the previous
  // one is in flight on the stream (hipMemcpyAsync); an event per buffer says when it may be refilled
  static const int STAGE_BUFS = 3;
  static const size_t STAGE_BYTES = (size_t)64 << 20;
  void* pin[STAGE_BUFS] = {nullptr};
  hipEvent_t pin_ev[STAGE_BUFS] = {nullptr};
  bool pin_busy[STAGE_BUFS] = {false};
  int pin_next = 0;
  double staged_bytes = 0, staged_fill_s = 0;  // bookkeeping for dsk_load_stats
};
void ctx_model_released(dsk_ctx* c);
int stage_copy(dsk_ctx* ctx, const HostSrc& src, uint64_t src_off, void* dev_dst, size_t bytes);
int upload_tensor(dsk_ctx* ctx, DTensor& t, const HostSrc& src);
struct dsk_model;
int bind_src(dsk_model* m, int role, int layer, int quant, const int32_t shape[4], const HostSrc& src, size_t bytes);
// a K-quant tensor stored in the engine's own plane layout (tools/repack.py): planes[0..3] = qs, sc, hm
// (Q3_K only), dm, each (experts x per-expert plane bytes) contiguous; copied into the device planes as they are
int bind_planes(dsk_model* m, int role, int layer, int quant, const HostSrc planes[4], const size_t bytes[4]);

static const int NROLES = 32;
int cdiv_i(int a, int b);
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
  hipGraphExec_t graph[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // (hydrate, logits, argmax, sample) x (short, long context)
  int mla_flash_min_kv = MLA_FLASH_MIN_KV;  // DSK_MLA_FLASH_MIN overrides (A/B)
  float* fl_part_o = nullptr;   // MLA long-context partials (n_chunks, H, lora) / (n_chunks, H, 2)
  float* fl_part_ml = nullptr;
  std::vector<MlaFlashArgs> mla_flash;
  float* sample_scratch = nullptr;  // segment sums / local maxima / arrival counter of dsk_forward_sample
  int* argmax_dev = nullptr;
  int* argmax_host = nullptr;
  // GEMV launch descriptors (forward.cpp): host copies + one device array
  std::vector<GemvLaunch> plans;
  GemvLaunch* plans_dev = nullptr;
  std::vector<int> lp_qkv_a, lp_qkv_b, lp_wv_b, lp_wo, lp_w13, lp_w2;  // per-layer indices into plans (-1: none)
  bool ride_shared = true, ride_kvwrite = true;  // options "fuse_shared" / "ride_kvwrite" (dsk_model_set_option)
  // ---- dsk_model_set_option (include/dsk.h): everything that selects kernels or changes numerics is an explicit,
  // per-model option; the environment is read only in -DDSK_AB builds (A/B tooling), never by the shipped library
  int rider_fill = 4;              // "rider_fill": workgroup fill divisor of the shared expert's rider (forward.cpp)
  bool compact_absent = true;      // "compact_absent": expert-sharded w1/w3 launch compacts its row space on the device
  bool want_timeline = false;      // "timeline": in-kernel wall-clock stamps (dsk_model_get_timeline)
  int moe_spin_limit = 0;          // "moe_spin_limit": polls before the fused expert launch's hand-off gives up (0: default 2^20; < 0: fault injection - workgroup 0 reports a give-up)
  bool force_exchange = false;     // "force_exchange": run the expert-sharded code path (two-launch form, RCCL exchange, combine launch) at world == 1 too
  int q2k_tiles = 1;               // "q2k_tiles": which Q2_K tensors live in the tiled layout (tile_device.h): 0 none, 1 the experts, 2 every converted role (engine.cpp role_tiled); set before the first bind
  bool any_bound = false;
  bool graph_with_comm = true;     // "graph_with_comm": the sharded step (RCCL exchange included) is captured into a hipGraph too, after the first eager token of a mode (0: enqueued eagerly)
  bool exchange_allgather = false;  // "exchange_allgather": the sharded exchange as ONE all-gather of the ranks' slot rows (each slot is read from
                                    // its owner's copy) instead of a sum all-reduce
  float* egather = nullptr;         // [world][n_active_routed][dim], allocated at finalize when that option is set
  int exchange_calls = 0;          // RCCL collectives enqueued by this model (eager path) - diagnostics
  int graph_capture_fallbacks = 0; // a sharded step whose capture / instantiation failed: the model went back to eager enqueueing (dsk_model_get_info)
  int handoff_fallbacks = 0;       // times a hand-off give-up switched this model to the two-launch form (dsk_model_get_info)
  bool sharded() const { return ctx->world > 1 || force_exchange; }
  std::vector<int> lp_sh13;  // shared expert's w1/w3 GLU riding in the router launch (-1: it is a task of lp_w13)
  // routed experts in one launch (kernels_moe.hip); grid == 0: the layer keeps the two-launch form (lp_w13, lp_w2)
  std::vector<MoeFfnArgs> moe_ffn;
  bool att_q8_in_wo = false;       // DSK_ATT_Q8_IN_WO=1: wo quantises the attention output in its own prologue (no finisher hand-off in the attention launch)
  bool fuse_moe = true;            // DSK_NO_FUSE_MOE at model creation switches it off (A/B, bit-identity tests)
  unsigned* moe_ctr = nullptr;     // slot_ctr[16] | slot_pass[16]
  float* moe_cand = nullptr;       // candidate records of the pipelined fused expert launch (kernels_moe_pipe.hip)
  unsigned* moe_blk_ctr = nullptr; // per-block arrivals of the fused expert launch (hidden vectors quantised by their producers)
  bool fuse_moe_float = true;      // option "fuse_moe_float": the fused expert launch for F8E5M2 / F16 / F32 weights too
  bool moe_q8_handoff = true;      // option "moe_q8_handoff"
  int moe_pipe = 0;                // option "moe_pipe": the fused expert launch pipelined by slot halves (kernels_moe_pipe.hip)
  int tail_prefetch = 8;           // option "tail_prefetch": cold-line prefetch workgroups per launch that has room for them (forward.cpp build_plans; 0: off)
  // DSK_TIMELINE=1 (debug): 8 wall-clock stamps per workgroup of the LAST launch of each kind in a token;
  // kind 0 first-stage projections, 1 per-head attention, 2 wo, 3 router + shared expert, 4 fused routed experts
  unsigned long long* moe_timeline = nullptr;  // base of [8 kinds][1024 workgroups][8]
  unsigned long long* timeline_of(int kind) const { return moe_timeline ? moe_timeline + (size_t)kind * DSK_TL_WGS * 8 : nullptr; }
  unsigned* err_host = nullptr;    // pinned, device-visible: bounded spins report here
  int lp_head = -1;
  // batched prompt ingestion (hydrate.cpp): buffers allocated by the first dsk_hydrate call that takes the batched path
  struct HydState* hyd = nullptr;
  int hydrate_chunk = 512;          // option "hydrate_chunk": tokens per batched chunk (every weight matrix is read once per chunk: a 512-token prompt
                                    // takes 424 ms in chunks of 128, 379 in chunks of 256, 333 in one; ~0.6 GB of chunk buffers at DeepSeek-V3 width)
  int hydrate_route_seed = 0;       // -DDSK_AB builds only (env DSK_HYD_ROUTE_SEED, tools/ab_build.sh): > 0 = the batched path routes every token to K
                                    // uniformly drawn experts - a measurement of balanced routing; the shipped library has no way to set it
  int hydrate_tap_layer = -1;       // option "hydrate_tap_layer" (parity harness): the block whose intermediates a batched chunk copies aside for
                                    // dsk_hydrate_get_buffer; the chunk itself runs unchanged
  int gemv_ahead = 3;               // option "gemv_ahead" (bits): 1 the first-stage projection launch, 2 wo request weights ahead of the staging of their vector (round 6, kernels_gemv.hip)
  bool hydrate_batched = true;      // option "hydrate_batched": 0 = dsk_hydrate always runs the per-token loop
  bool hydrate_tile_copies = true;  // option "hydrate_tile_copies": the batched path may keep tile-record copies of the plane-layout Q2_K matrices (hydrate.cpp)
  double hydrate_tile_copy_bytes = 0;  // bytes of those copies (made by the first batched dsk_hydrate call)
  long long hydrate_batched_tokens = 0, hydrate_looped_tokens = 0;  // dsk_model_get_info
  const char* hydrate_why = nullptr;  // why the last dsk_hydrate call looped (nullptr: it did not)
  std::vector<MlaHeadArgs> mla_head;   // per layer (MLA path)
  std::vector<double> head_attn_bytes;
  std::vector<HeadAttnArgs> head_attn;  // per layer (MHA path): second-stage projections + attention, one launch
  bool graph_primed[8] = {false, false, false, false, false, false, false, false};
  const char* class_filter = nullptr;  // dsk_time_kernel_class
  int class_launches = 0;
  double class_bytes = 0;
  unsigned* router_counter = nullptr;
  unsigned* att_counter = nullptr;   // one arrival counter per 256-block of the attention output
  // MHA long contexts: mha_split workgroups per head from kv_len >= mha_split_min on (head_attn_kernel)
  int mha_split = 1, mha_split_min = 0;
  float* mha_split_part = nullptr;
  unsigned* mha_split_counter = nullptr;
  unsigned* comb_counter = nullptr;  // one arrival counter per row group of the fused MoE combine
  int target_wgs = 1024;
  // parity harness (dsk_model_run_block): Q8_K staging taps of ONE block, device arenas allocated on first use
  int stage_layer = -1;          // >= 0 while a tapped block runs
  int stage_kv_len = 0;          // kv_len of the last tapped block (cache rows the harness may read back)
  int stage_last_layer = -1;
  int8_t* tap_qs = nullptr;      // arena of int8 codes; regions below (element offsets, multiples of 256)
  float* tap_d = nullptr;        // block scales: region offsets / 256
  float* tap_latent = nullptr;   // MLA: per-head latent outputs (H, lora)
  float* stage_x_mid = nullptr;  // residual stream after the attention half of the block
  size_t tap_off_xattn = 0, tap_off_qa = 0, tap_off_kva = 0, tap_off_xffn = 0, tap_off_xffn_sh = 0, tap_off_hb = 0,
         tap_off_latent = 0, tap_off_final = 0, tap_off_att = 0, tap_total = 0;
  // profiling
  bool profiling = false;
  std::vector<KTime> ktimes;
  std::map<std::string, int> kindex;
};

// expected logical shape of a role (src/model.cpp:184-285, 393-457, 557-616, 766-871)
struct RoleShape {
  int quant;      // expected quant
  int e, rows, n; // e = 0: 2-D
  bool ok;
};
RoleShape role_shape(const dsk_model* m, int role, int layer);
extern const int ALL_LAYER_ROLES[];
extern const int N_LAYER_ROLES;
int build_plans(dsk_model* m);   // forward.cpp: GEMV launch descriptors, built once at finalize
void free_plans(dsk_model* m);
void hydrate_free(dsk_model* m);  // hydrate.cpp
