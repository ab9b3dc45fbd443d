/*
 * dsk.h -- C ABI of the MI355X (gfx950) single-batch DeepSeek decode engine.
 *
 * This is the drop-in boundary for ONE hot path of andrewkchan/deepseek.cpp:
 *   Model::forward -> Model::_forward_cpu -> Block::_block_cpu
 *   (reference src/model.cpp:874-883, src/infer.cpp:1265-1317, src/infer.cpp:810-932)
 * The reference dispatches that path on `enum class Device` (src/model.h:36-38,
 * src/model.cpp:290-322,874-883).  A maintainer adds one enumerator (Device::HIP)
 * and routes Model::forward to dsk_forward(); main/sampler/tokenizer/codec stay
 * unchanged (see INTEGRATION.md for the binding a maintainer would write).
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++/torch types in any signature
 *   - every function returns 0 on success, a negative dsk_status on failure and never
 *     aborts across the boundary (the reference prints "FATAL:" and assert(false)s,
 *     src/model.cpp:131-132); the message is available from dsk_last_error()
 *   - host pointers passed to dsk_model_bind() need to stay valid only for the call:
 *     the engine copies (and re-lays-out) the bytes into HBM and owns device memory
 *   - one host thread drives one context; one HIP stream per context
 */
#ifndef DSK_H
#define DSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSK_ABI_VERSION 1

typedef enum dsk_status {
  DSK_OK = 0,
  DSK_ERR_INVALID = -1,      /* bad argument / shape / role */
  DSK_ERR_UNSUPPORTED = -2,  /* valid in the reference but not built here */
  DSK_ERR_HIP = -3,          /* a HIP runtime call failed */
  DSK_ERR_STATE = -4,        /* call out of order (e.g. forward before finalize) */
  DSK_ERR_NOMEM = -5,
  DSK_ERR_COMM = -6          /* RCCL failure */
} dsk_status;

/* Quant of in-memory tensors; same order as reference `enum class Quant` (src/codec.h:79-85). */
typedef enum dsk_quant {
  DSK_QUANT_F32 = 0,
  DSK_QUANT_F16 = 1,
  DSK_QUANT_F8E5M2 = 2,
  DSK_QUANT_Q2_K = 3,
  DSK_QUANT_Q3_K = 4
} dsk_quant;

/* reference `enum class InferenceMode` (src/model.h:40-43) */
typedef enum dsk_mode {
  DSK_MODE_HYDRATE_KV_CACHE = 0, /* update KV caches only, skip final norm + lm_head (src/infer.cpp:1284-1287) */
  DSK_MODE_OUTPUT_LOGITS = 1
} dsk_mode;

enum { DSK_ACT_GELU = 0, DSK_ACT_SILU = 1 };                 /* src/model.h:16-19 */
enum { DSK_TOPK_GREEDY = 0, DSK_TOPK_GROUP_LIMITED_GREEDY = 1 }; /* src/model.h:25-29 (NOAUX_TC asserts in the reference) */
enum { DSK_SCORE_SOFTMAX = 0, DSK_SCORE_SIGMOID = 1 };       /* src/model.h:31-34 */

/* POD mirror of reference `struct Config` (src/model.h:47-96); same field meaning. */
typedef struct dsk_config {
  int32_t dim;
  int32_t hidden_dim;
  int32_t n_layers;
  int32_t n_heads;
  int32_t vocab_size;
  int32_t max_seq_len;
  float rope_theta;
  float norm_eps;
  int32_t act;                    /* DSK_ACT_* */
  int32_t first_k_dense_replace;
  int32_t n_shared_experts;
  int32_t n_routed_experts;
  int32_t n_active_routed;
  int32_t moe_intermediate_size;
  float routed_scaling_factor;
  int32_t n_group;
  int32_t norm_topk_prob;         /* bool */
  int32_t scoring_func;           /* DSK_SCORE_* */
  int32_t topk_group;
  int32_t topk_method;            /* DSK_TOPK_* */
  int32_t has_moegate_bias;       /* bool; also selects rope_v3 (src/infer.cpp:958,1073) */
  int32_t use_mla;                /* bool; BlockMLA vs BlockMHA */
  int32_t kv_lora_rank;
  int32_t q_lora_rank;
  int32_t qk_nope_head_dim;
  int32_t qk_rope_head_dim;
  int32_t v_head_dim;
  int32_t weight_quant;           /* dsk_quant */
  int32_t block_size[2];          /* F8E5M2 block scales (src/model.h:85); {0,0} otherwise */
  int32_t rs_original_max_position_embeddings; /* KV ring modulus (src/infer.cpp:1274-1277) */
} dsk_config;

/*
 * Tensor roles = the tensors the reference Block/Model constructors bind by name
 * (src/model.cpp:184-285, 393-457, 557-616, 766-871).  `layer` is the block index,
 * or -1 for model-level tensors.  Scales (F8E5M2 only) use role + DSK_ROLE_SCALE.
 */
typedef enum dsk_role {
  /* model level (layer = -1) */
  DSK_ROLE_EMBED = 0,        /* model.embed.weight   (vocab, dim)            */
  DSK_ROLE_FINAL_NORM = 1,   /* model.norm.weight    (dim) F32               */
  DSK_ROLE_OUTPUT = 2,       /* model.output.weight  (vocab, dim); absent => tied to EMBED (src/model.cpp:852-856) */
  /* per layer */
  DSK_ROLE_ATTN_NORM = 10,   /* attn.norm.weight      (dim) F32              */
  DSK_ROLE_Q_A_NORM = 11,    /* attn.q_a_norm.weight  (q_lora_rank) F32      */
  DSK_ROLE_KV_A_NORM = 12,   /* attn.kv_a_norm.weight (kv_lora_rank) F32     */
  DSK_ROLE_FFN_NORM = 13,    /* mlp.norm.weight       (dim) F32              */
  DSK_ROLE_WQ = 14,          /* attn.wq   (n_heads*head_dim, dim)  [q_lora_rank == 0] */
  DSK_ROLE_WQ_A = 15,        /* attn.wq_a (q_lora_rank, dim)                 */
  DSK_ROLE_WQ_B = 16,        /* attn.wq_b (n_heads*head_dim, q_lora_rank) [MHA] */
  DSK_ROLE_WKV_A = 17,       /* attn.wkv_a (kv_lora_rank+rope, dim)          */
  DSK_ROLE_WKV_B = 18,       /* attn.wkv_b (n_heads*(nope+v), kv_lora_rank) [MHA] */
  DSK_ROLE_WO = 19,          /* attn.wo   (dim, n_heads*v_head_dim)          */
  DSK_ROLE_WC = 20,          /* attn.wc   (n_heads*kv_lora_rank, q_lora_rank) [MLA] */
  DSK_ROLE_WQ_ROPE_B = 21,   /* attn.wq_rope_b (n_heads*rope, q_lora_rank)  [MLA] */
  DSK_ROLE_WV_B = 22,        /* attn.wv_b (n_heads*v_head_dim, kv_lora_rank) [MLA] */
  DSK_ROLE_W1 = 23,          /* mlp.w1 (hidden,dim) or (E, moe_inter, dim)   */
  DSK_ROLE_W2 = 24,          /* mlp.w2 (dim,hidden) or (E, dim, moe_inter)   */
  DSK_ROLE_W3 = 25,          /* mlp.w3 like w1                               */
  DSK_ROLE_SHARED_W1 = 26,   /* shared_mlp.w1 (n_shared*moe_inter, dim)      */
  DSK_ROLE_SHARED_W2 = 27,   /* shared_mlp.w2 (dim, n_shared*moe_inter)      */
  DSK_ROLE_SHARED_W3 = 28,   /* shared_mlp.w3                                */
  DSK_ROLE_MOEGATE = 29,     /* moegate.weight (E, dim) always F32 (src/model.cpp:196-198) */
  DSK_ROLE_MOEGATE_BIAS = 30,/* moegate.bias (E) F32, V3 only                */
  DSK_ROLE_SCALE = 64        /* add to a weight role for its ".scale" tensor (F32 block scales) */
} dsk_role;

typedef struct dsk_ctx dsk_ctx;
typedef struct dsk_model dsk_model;

/* ---- context -------------------------------------------------------------- */
/* device_ordinal = HIP device index this context (process) drives. */
int dsk_ctx_create(int device_ordinal, dsk_ctx** out);
int dsk_ctx_destroy(dsk_ctx* ctx);
const char* dsk_last_error(void);
int dsk_abi_version(void);

/* ---- expert-sharded multi-GPU (one process per GPU, RCCL over xGMI) --------
 * No reference counterpart: routed experts e with
 * e / ceil(E/world) == rank live on this rank; everything else is replicated.
 * Per MoE layer every rank computes the slots whose expert it owns and writes ZEROS for the others; one sum
 * all-reduce over the K x dim slot buffer (each slot is non-zero on exactly one rank, so the sum is exact and
 * order-independent) hands every rank all slot outputs, which are then added in k order: bit-identical to the
 * 1-GPU run (an all-gather of per-rank partial sums would move the same K x dim bytes at world = K and give up
 * the bit identity).  EXPERIMENTAL: the exchange has never run on more than one GPU (the
 * single-GPU dry run below validates everything except the RCCL call).  uid = 128-byte ncclUniqueId made
 * by rank 0 with dsk_comm_unique_id() and broadcast by the launcher. */
int dsk_comm_unique_id(void* uid128);
int dsk_comm_init(dsk_ctx* ctx, const void* uid128, int rank, int world);
/* uid128 == NULL with world > 1: single-process DRY RUN of shard `rank`: the model binds / computes only the
   experts this rank owns and skips the all-reduce (used to validate the sharded path on one GPU). */
/* Which experts of an E-expert routed stack rank `rank` of `world` owns: [*base, *base + *count).
   Host arithmetic only (no GPU needed). */
int dsk_expert_shard(int n_experts, int world, int rank, int* base, int* count);
/* ... and the rank that owns one expert under that partition (what the all-gather form of the exchange evaluates per slot). */
int dsk_expert_owner(int n_experts, int world, int expert, int* rank);
/* Tensor-parallel partitions (DESIGN.md 4.4 - host arithmetic of the design, no engine mode yet):
   output rows [*row0, *row0 + *count) of a `rows`-row GEMV for rank `rank` of `world`, in multiples of `unit` rows
   (rows % unit == 0); heads likewise.  Rows are independent dot products: the all-gather of the ranges equals the one-GPU
   vector bit for bit.  No GPU needed. */
int dsk_tp_rows(int rows, int unit, int world, int rank, int* row0, int* count);
int dsk_tp_heads(int n_heads, int world, int rank, int* head0, int* count);

/* ---- model life-cycle (replaces Model::Model binding, src/model.cpp:756-872) */
int dsk_model_create(dsk_ctx* ctx, const dsk_config* cfg, dsk_model** out);
/* Upload one tensor.  `shape` is the logical shape as in QTensor::shape (src/codec.h:107-120;
 * 0 = unused dim); `bytes` must equal the reference byte count (src/codec.cpp:166-234).
 * For routed-expert tensors on a sharded context only this rank's experts are kept. */
int dsk_model_bind(dsk_model* m, int role, int layer, int quant,
                   const int32_t shape[4], const void* host_ptr, size_t bytes);
/* Fill EVERY tensor the config requires with valid random blocks directly in HBM
 * (220 GB of V3 Q2_K does not fit host disk/RAM).  Deterministic in seed. */
int dsk_model_synthesize(dsk_model* m, uint64_t seed);
/* Check all required tensors are bound, allocate KV caches (src/model.cpp:459-460,618-619)
 * and activation scratch (src/model.cpp:677-726). */
int dsk_model_finalize(dsk_model* m);
int dsk_model_destroy(dsk_model* m);
/* Per-model options, set between dsk_model_create and dsk_model_finalize.  Everything that selects kernels, launch
 * shapes or diagnostics is an explicit option of ONE model: the library never reads the environment (a stray variable
 * in the host application's environment must not change kernels or numerics; only -DDSK_AB builds, made by
 * tools/ab_build.sh for measurements, seed these options from DSK_* variables).  Keys (int values):
 *   "fuse_moe"        1  routed experts of a MoE block in one launch (0: two launches; bit-identical)
 *   "fuse_shared"     1  the shared expert's w1/w3 rides in the router launch (0: ninth task of the experts' launch)
 *   "ride_kvwrite"    1  MLA: the latent cache write rides in the second-stage projection launch
 *   "att_q8_in_wo"    0  wo quantises the attention output in its own prologue (no finisher in the attention launch)
 *   "compact_absent"  1  expert-sharded w1/w3 launch compacts its row space over the experts present on the rank
 *   "rider_fill"      4  workgroup fill divisor of the shared expert's rider (1..16)
 *   "mla_flash_min" 320  context length from which MLA attention runs on the matrix cores
 *   "mha_split_min" 1024 context length from which MHA attention splits a head's context over workgroups
 *   "tail_prefetch"   8  cold-line prefetch: workgroups on CUs the per-head attention launch leaves idle, and behind the fused
 *                        expert launch, read what the following launches open with (descriptors, norm weights, the gate's
 *                        bias row) into the XCDs' L2 (0: off; results unaffected)
 *   "moe_q8_handoff"  1  fused expert launch: hidden vectors handed over as Q8_K, quantised once by their producers
 *   "fuse_moe_float"  1  fused expert launch for F8E5M2 / F16 / F32 weights too (0: two launches; bit-identical)
 *   "timeline"        0  in-kernel wall-clock stamps for dsk_model_get_timeline
 *   "moe_spin_limit"  0  polls before the fused expert launch's hand-off wait gives up (0: 2^20); < 0: fault
 *                        injection (a give-up is reported although the hand-off succeeded: exercises the fallback)
 *   "force_exchange"  0  run the expert-sharded code path (two-launch experts, RCCL all-reduce when the context has a
 *                        communicator, separate combine launch) at world == 1 too
 *   "graph_with_comm" 1  the sharded step - RCCL exchange included - is captured into a hipGraph after the first (eager) token of a
 *                        mode, like the one-GPU step (0: enqueued eagerly; capture validated on a 1-rank communicator only)
 *   "exchange_allgather" 0  the sharded exchange as ONE ncclAllGather of every rank's K slot rows, each slot then read from the
 *                        copy of the rank that owns its expert (dsk_expert_owner), instead of a sum ncclAllReduce; bit-identical;
 *                        no multi-GPU measurement exists for either form
 *   "q2k_tiles"       1  HBM layout of Q2_K matrices, fixed BEFORE the first tensor is bound (DSK_ERR_STATE afterwards):
 *                        0 planes for the dot4 kernels; 1 the routed and shared experts' matrices as 16-row x 256-column tile
 *                        records (1344 B) whose sub-block dots run on v_mfma_i32_16x16x64_i8; 2 every role that has a tiled
 *                        kernel.  Same integer arithmetic at every level (src/quant.cpp:666-783), a different f32 association of
 *                        the block sums; the caller's bytes are reference-format blocks at every level
 *   "hydrate_chunk"   512  tokens per batched chunk of dsk_hydrate (1 .. 1024)
 *   "hydrate_batched" 1  0: dsk_hydrate always runs the per-token loop
 *   "hydrate_tile_copies" 1  at "q2k_tiles" = 1 the first batched dsk_hydrate call may copy the plane-layout matrices it multiplies
 *                        into tile records (0: no copies - such a model then runs the loop)
 *   "hydrate_tap_layer" -1  parity harness: copy block l's stages of every batched chunk aside for dsk_hydrate_get_buffer
 *   "gemv_ahead"      3  bits: 1 the first-stage projection launch (and the MLA second stage), 2 wo request their first weights
 *                        behind the loads of their vector and AHEAD of its staging (kernels_gemv.hip gemv_ahead_kernel & co;
 *                        bit-identical to 0; instantiated for the DeepSeek-V3 and V2-Lite row lengths, other shapes run the plain
 *                        kernel); get_info "gemv_ahead_plans" counts the launch plans that run such a kernel
 *   "moe_pipe"        0  experimental schedules of the fused expert launch (1: slot halves with service waves, 2: staged second
 *                        half; bit-identical to 0 and slower on MI355X: EXPERIMENTS.md 6.1-6.2) */
int dsk_model_set_option(dsk_model* m, const char* key, int value);
/* Read-only counters: "handoff_fallbacks" (times a hand-off give-up moved the model to the two-launch form; the token
 * that hit it was re-run transparently), "fused_moe_layers", "graph_captured", "exchange_calls" (RCCL collectives this
 * model has enqueued eagerly), "tiled_tensors" (weight tensors held as tile records, option "q2k_tiles"),
 * "hydrate_batched_tokens" / "hydrate_looped_tokens", "hydrate_tile_copy_mb" (MiB of tile copies the batched prompt path holds),
 * "gemv_ahead_plans". */
int dsk_model_get_info(dsk_model* m, const char* key, int* value);
/* models created on the context and not yet destroyed (a context destroyed while models are alive is freed by the last
 * dsk_model_destroy) */
int dsk_ctx_live_models(dsk_ctx* ctx);

/* ---- direct-to-HBM `.dseek` loader ---------------------------
 * Replaces, for this device, YALMData::from_directory (src/codec.cpp:333-365: every file of the directory in
 * sorted order, metadata from the first), Config::from_yalm (src/model.cpp:21-127) and the by-name tensor walk of the
 * Model / Block constructors (src/model.cpp:766-871).  Tensor byte ranges are read with parallel preads into pinned
 * staging buffers and copied to HBM while the next piece is read; a sharded context reads only its own experts.
 * `context` > 0 caps max_seq_len like the reference's -c option (src/model.cpp:73-76). */
typedef struct dsk_load_stats {
  uint64_t file_bytes;     /* bytes of the tensors the model consumed (all ranks' experts included) */
  uint64_t staged_bytes;   /* bytes that actually went host -> HBM on this rank */
  double seconds;          /* wall time of the whole call (open + parse + read + copy + finalize) */
  double read_seconds;     /* of which: filling the pinned buffers (pread) */
  int32_t n_files, n_tensors;
} dsk_load_stats;
/* Parse the checkpoint's header(s) only: the configuration the model would get, number of files and tensors,
 * total tensor bytes.  No GPU needed. */
int dsk_dseek_read_config(const char* dir, int context, dsk_config* out, int32_t* n_files, int32_t* n_tensors,
                          uint64_t* tensor_bytes);
/* create + bind every tensor + finalize.  `stats` may be NULL. */
int dsk_model_load_dseek(dsk_ctx* ctx, const char* dir, int context, dsk_model** out, dsk_load_stats* stats);
/* ... with model options applied between create and the first bind: "key=value,key=value" (dsk_model_set_option keys),
 * e.g. "q2k_tiles=2" (every Q2_K matrix as tile records: dsk_hydrate bit-identical to the per-token loop).  options may be NULL. */
int dsk_model_load_dseek_opts(dsk_ctx* ctx, const char* dir, int context, const char* options, dsk_model** out, dsk_load_stats* stats);

/* ---- the hot path (replaces Model::forward, src/model.cpp:874-883) -------- */
/* One token.  mode == OUTPUT_LOGITS: host_logits receives vocab_size floats
 * (what InferenceState::logits() holds, src/model.h:137).  HYDRATE: host_logits may be NULL. */
int dsk_forward(dsk_model* m, int token, int pos, int mode, float* host_logits);
/* The prompt phase.  The reference feeds a prompt to Model::forward one token at a time
 * (src/main.cpp:312-319, InferenceMode::HYDRATE_KV_CACHE for every token but the last; src/infer.cpp:1284-1287).
 * dsk_hydrate is DEFINED as that loop:
 *     for i < n_tokens - 1:  dsk_forward(m, tokens[i], pos0 + i, DSK_MODE_HYDRATE_KV_CACHE, NULL)
 *     dsk_forward(m, tokens[n_tokens - 1], pos0 + n_tokens - 1, mode, host_logits)
 * and runs it as batched launches - every weight matrix read once per chunk of up to "hydrate_chunk" tokens (option,
 * default 512), the Q2_K row products as i8 GEMMs on the matrix pipe - when the model qualifies: Q2_K weights with the experts'
 * matrices stored as tile records (option "q2k_tiles" >= 1: the default), MHA or MLA attention with a q latent, one GPU, and
 * positions below the ring wrap (rs_original_max_position_embeddings).
 *   "q2k_tiles" = 2 (every matrix as tile records): KV-cache rows, routing and logits are BIT-identical to the loop
 *       (tests/test_hydrate_gpu.py), at every context length below the wrap.
 *   "q2k_tiles" = 1 (default): the first batched call makes tile-record copies of the plane-layout matrices it multiplies (~4-5 GB
 *       for DeepSeek-V3; option "hydrate_tile_copies" = 0 refuses them and the call runs the loop; get_info "hydrate_tile_copy_mb")
 *       and runs the same kernels on them: the rows are the "q2k_tiles" = 2 engine's, and the default engine's loop within the float
 *       association of its plane kernels - audited against the reference's arithmetic block by block like decode
 *       (tests/test_teacher_forced_gpu.py), not bit-identical to a loop that associates differently.
 * Anything else - other models, the tokens at and past the wrap, no room for buffers or copies - runs the loop itself, so the call
 * is always valid.  dsk_hydrate_why_not: "" when the batched path applies to this model, else the reason.
 * dsk_model_get_info "hydrate_batched_tokens" / "hydrate_looped_tokens" count what ran where. */
int dsk_hydrate(dsk_model* m, const int32_t* tokens, int n_tokens, int pos0, int mode, float* host_logits);
const char* dsk_hydrate_why_not(dsk_model* m);
/* Parity harness: residual stream of token `index` of the last batched chunk after block `layer`
 * (dsk_model_set_trace(m, 1) before the first dsk_hydrate call); the per-token counterpart is dsk_model_get_trace_x. */
int dsk_hydrate_get_trace_x(dsk_model* m, int layer, int index, float* x_out);
/* Parity harness: rows [row0, row0 + rows) - one row per token - of a named intermediate of the LAST batched chunk.  With option
 * "hydrate_tap_layer" = l, block l's stages are copied aside while a chunk runs (the chunk itself is unchanged) under the names
 * dsk_model_get_stage uses for the per-token block ("q8.x_attn.qs" / ".d", "q_a", "kv_a", "att_out", "x_mid", "router_logits",
 * "route_e", "route_w", "hb", "eout", ...: hydrate.cpp lists them); tests/test_hydrate_gpu.py hands them to the oracle-side block
 * audit.  Without a tap: "x" (the residual stream after the last block), "route_e" (the last MoE block's).  `bytes` must be
 * rows x the row size of that name. */
int dsk_hydrate_get_buffer(dsk_model* m, const char* name, int row0, int rows, void* out, size_t bytes);
/* Enable/disable replaying the token step from a captured hipGraph (default on). */
/* The engine's own pinned host buffer of vocab_size floats (the D2H target of every OUTPUT_LOGITS step).  Passing it
   as `host_logits` to dsk_forward skips the extra host copy: a host application can make InferenceState::logits()
   (src/model.h:137) point here.  NULL before finalize. */
float* dsk_model_host_logits(dsk_model* m);

/* Greedy decode step (the sampler's temperature == 0 branch): like dsk_forward(OUTPUT_LOGITS) but the
   argmax of the logits is taken on the device with Sampler::sample_argmax's tie rule (strict >: the lowest index
   among equal maxima, src/sampler.cpp:28-39) and only the token id crosses PCIe (4 bytes instead of vocab * 4). */
int dsk_forward_argmax(dsk_model* m, int token, int pos, int32_t* next_token);
/* The sampler's other branch on the device (Sampler::sample, src/sampler.cpp:41-75): softmax with `temperature`, then
   the first token in vocabulary order whose cumulative probability reaches coin * top_p (vocab_size - 1 if none);
   temperature == 0 is dsk_forward_argmax.  `coin` is the caller's std::rand() / (float)RAND_MAX: the random stream
   stays the host's, only the 129 280-wide softmax and the scan move to the GPU (the reference spends two expf passes
   over the vocabulary per token on one core here).  The reference adds the vocabulary left to right in f32; the
   device sums in a fixed tree, so the token equals the reference's unless coin * top_p falls within ~1e-5 of a
   boundary of the cumulative distribution. */
int dsk_forward_sample(dsk_model* m, int token, int pos, float temperature, float top_p, float coin, int32_t* next_token);
/* Sampler::sample_prob (src/sampler.cpp:12-26) on the device: softmax probability of `index` under this step's logits,
   i.e. what run_perplexity sums per token (src/main.cpp:386-401); 4 bytes cross PCIe. */
int dsk_forward_prob(dsk_model* m, int token, int pos, int index, float* prob);
int dsk_model_set_graph(dsk_model* m, int enable);

/* Algorithmic HBM bytes one forward at `pos` must touch, with true block sizes
 * (the corrected analogue of Model::active_bytes, src/model.cpp:885-901). */
double dsk_model_active_bytes(const dsk_model* m, int pos);
/* Bytes of device memory held by the model (weights + caches + scratch). */
double dsk_model_device_bytes(const dsk_model* m);

/* ---- observation taps for the parity harness (the reference's dead DEBUG_MODEL
 * hooks, src/infer.cpp:10-119, show the intended mechanism) ------------------- */
/* Record the router decision of every MoE layer of the LAST forward.
 * experts: n_layers*n_active_routed ints (-1 for dense layers); weights likewise. */
int dsk_model_get_routing(dsk_model* m, int32_t* experts, float* weights);
/* Copy the residual stream x (dim floats) as it was after `layer` in the last forward;
 * requires dsk_model_set_trace(m, 1) before the forward (disables the graph). */
/* The per-slot expert outputs W2_k (act(W1_k x) * W3_k x) of the LAST MoE layer of the last forward:
   (n_active_routed [+1 if shared experts]) x dim floats, slot order = k order, shared expert last. */
int dsk_model_get_slot_outputs(dsk_model* m, float* out);
int dsk_model_set_trace(dsk_model* m, int enable);
int dsk_model_get_trace_x(dsk_model* m, int layer, float* x_out);

/* Teacher-forced execution of ONE block for the parity harness: the residual stream x_in (dim floats) goes in, the
 * block `layer` runs eagerly at position `pos` (KV caches as the previous calls left them; row kv_pos is written),
 * x_out receives the block's output.  Every Q8_K staging point of the block (quantize_row_q8_K_ref call sites,
 * src/infer.cpp:325-336 under :823-931) is tapped: dsk_model_get_stage returns, by name, the int8 codes and block
 * scales each launch actually staged ("q8.<point>.qs" / ".d": x_attn, q_a, kv_a, att, latent, x_ffn, x_ffn_tap,
 * x_ffn_shared, hb, x_final) and the float intermediates between the launches (x_mid, q_a, kv_a, att_out, vb_out,
 * latent_out, q_c, q_rope, router_logits, gate_scores, route_e, route_w, hb, eout) plus the cache rows
 * [0, kv_len) of the block (k_cache, v_cache, nope_cache, rope_cache).  `bytes` must not exceed the stage's size. */
int dsk_model_run_block(dsk_model* m, int layer, const float* x_in, int pos, float* x_out);
/* final norm + classifier on a given residual stream (src/infer.cpp:1292-1316); taps "q8.x_final.*" */
int dsk_model_run_head(dsk_model* m, const float* x_in, float* logits);
int dsk_model_get_stage(dsk_model* m, const char* name, void* out, size_t bytes);
/* Parity harness: overwrite rows [row0, row0 + nrows) of one KV cache of `layer` ("k_cache" / "v_cache": n_heads * head_dim /
 * n_heads * v_head_dim f16 per row; "nope_cache" / "rope_cache": kv_lora_rank / qk_rope_head_dim) with the caller's f16
 * bits, so that ONE block can be audited at a long context (the attention regimes that start at 320 / 1024 cached
 * positions) without decoding thousands of tokens first. */
int dsk_model_set_cache_rows(dsk_model* m, int layer, const char* cache, int row0, int nrows, const uint16_t* rows);
/* ... and the read side: the same rows as f16 bits (what a prompt left in the cache). */
int dsk_model_get_cache_rows(dsk_model* m, int layer, const char* cache, int row0, int nrows, uint16_t* rows);

/* Per-kernel-class device time of ONE eager forward bracketed by HIP events on the
 * engine stream.  names: up to max_classes pointers to static strings. */
typedef struct dsk_kernel_time {
  const char* name;
  int32_t launches;
  float total_ms;
  double algo_bytes;   /* algorithmic HBM bytes those launches move */
} dsk_kernel_time;
int dsk_profile_forward(dsk_model* m, int token, int pos, dsk_kernel_time* out, int max_classes, int* n_classes);

/* ---- op-level entry points (host buffers in/out; mirror the reference's
 * "exposed for tests" set, src/model.h:503-535, plus the static helpers) ------ */
/* quantize_row_q8_K_ref (src/quant.cpp:616-653): n % 256 == 0.
 * qs: n int8; d: n/256 floats; bsums: n/16 int16. */
int dsk_q8k_quantize(dsk_ctx* ctx, const float* x, int n, int8_t* qs, float* d, int16_t* bsums);
/* matmul (src/infer.cpp:381-417): out[d] = W(d,n) . x(n).  w: reference byte layout.
 * scale/block_size: F8E5M2 block scales or NULL. */
int dsk_gemv(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale,
             const int32_t block_size[2], int d, int n, const float* x, float* out);
/* matmul_expert (src/infer.cpp:423-469): slice `expert` of a stacked (E,d,n) tensor. */
int dsk_gemv_expert(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale,
                    const int32_t block_size[2], int n_experts, int expert, int d, int n,
                    const float* x, float* out);
/* dequantize one row of an embedding table (Model::_copy_embedding, src/infer.cpp:1217-1263). */
int dsk_embed_row(dsk_ctx* ctx, int quant, const void* w, size_t w_bytes, const float* scale,
                  const int32_t block_size[2], int vocab, int dim, int token, float* out);
/* rmsnorm (src/infer.cpp:601-611) */
int dsk_rmsnorm(dsk_ctx* ctx, const float* x, const float* weight, int size, float eps, float* out);
/* Sampler::sample / sample_argmax (src/sampler.cpp:28-75) on a host logits vector; see dsk_forward_sample */
int dsk_sample(dsk_ctx* ctx, const float* logits, int vocab_size, float temperature, float top_p, float coin, int32_t* token);
/* moe_gate (src/infer.cpp:493-599): scores are the raw router logits (modified in place in the
 * reference; here read-only).  bias may be NULL. */
int dsk_moe_gate(dsk_ctx* ctx, const float* scores, const float* bias, int n_routed, int n_active,
                 int norm_topk_prob, float routed_scaling_factor, int scoring_func, int topk_method,
                 int n_group, int topk_group, int32_t* active_experts, float* active_weights);
/* rope / rope_v3 (src/infer.cpp:648-685) on `n_heads` vectors of length d (= rotary dim), in place. */
int dsk_rope(dsk_ctx* ctx, float* vec, int n_heads, int d, int pos, float theta, int is_v3);
/* attn over all heads (mha_cpu, src/infer.cpp:1143-1164): kb (kv_len, n_heads*head_dim) f16,
 * vb (kv_len, n_heads*v_head_dim) f16, q (n_heads*head_dim) -> out (n_heads*v_head_dim). */
int dsk_attn_mha(dsk_ctx* ctx, const float* q, const uint16_t* kb, const uint16_t* vb, int n_heads,
                 int head_dim, int v_head_dim, int kv_len, float* out);
/* attn_mla over all heads (src/infer.cpp:766-804): ckv (kv_len, kv_lora_rank) f16,
 * krope (kv_len, rope) f16, q_c (n_heads*kv_lora_rank), q_rope (n_heads*rope)
 * -> out (n_heads*kv_lora_rank); softmax scale 1/sqrt(head_dim). */
int dsk_attn_mla(dsk_ctx* ctx, const float* q_c, const float* q_rope, const uint16_t* ckv,
                 const uint16_t* krope, int n_heads, int head_dim, int kv_lora_rank, int rope_dim,
                 int kv_len, float* out);

/* The MoE router as the model runs it (F32 GEMV over rmsnorm(x, norm_w) -- src/infer.cpp:839,847 -- in the router
 * kernel's own summation tree): raw logits (n_routed floats, before scoring).  norm_w may be NULL (x used as is). */
int dsk_router_logits(dsk_ctx* ctx, const float* w, const float* x, const float* norm_w, float eps, int n_routed, int dim,
                      float* logits);

/* Measured streaming-read bandwidth of this GPU (GB/s): grid-stride dwordx4 sum over `bytes`
 * (SURVEY 8d "measured roofline" denominator).  Best of `iters`. */
int dsk_measure_read_bw(dsk_ctx* ctx, size_t bytes, int iters, double* gbps_out);

/* Duration of one kernel class (a name reported by dsk_profile_forward, e.g. "gemv_experts_w13") inside
   the model: its launches of a whole token are enqueued back to back `reps` times between two HIP events
   on the engine stream -- no events between launches, so the figure matches rocprofv3 --kernel-trace plus
   the same-stream kernel boundary.  The activations / KV slot of `pos` are left undefined. */
int dsk_time_kernel_class(dsk_model* m, const char* name, int pos, int reps, double* us_per_launch,
                          double* bytes_per_launch, int* launches_per_token);

/* Diagnostics, host only (no GPU needed): the geometry the launch planner picks for n_tasks equal (rows x n) matrices of
 * one activation group; kind 0 plain, 1 GLU pair (w1 / w3), 2 / 3 tasks with the fused MoE combine; act_mode 0 ready
 * Q8_K, 1 f32, 2 f32 + rmsnorm.  out[8] = lanes per row, rows per lane group (R), column steps in flight (U), waves per
 * workgroup, grid, LDS bytes, activation groups, rows per workgroup step.  kind | 0x100: the same launch on Q2_K weights in the
 * tiled layout (option "q2k_tiles"): out[0..2] are placeholders, out[7] = item partials a round of strips holds in LDS.
 * tests/test_host_logic.py pins the choices for the DeepSeek-V3 shapes. */
int dsk_plan_gemv(int quant, int rows, int n, int n_tasks, int kind, int act_mode, int target_wgs, int* out);
/* ... and which of the round-6 "weights requested ahead of the staging" kernels the engine runs for that launch at its default
 * options (kernels_gemv.hip): 0 none, 1 gemv_ahead_kernel (f32 + rmsnorm vector, rows of <= 2 column steps), 2 gemv_ahead_q8_kernel
 * (ready Q8_K vector), 3 gemv_kvwrite_ahead_kernel (kvwrite != 0: the MLA second-stage launch with its cache-write rider).
 * tests/test_host_logic.py pins the DeepSeek-V3 / V2-Lite launches that qualify. */
int dsk_plan_gemv_ahead(int quant, int rows, int n, int n_tasks, int kind, int act_mode, int kvwrite, int* ahead_kind);

/* Diagnostics: time the GEMV kernel on device-resident synthetic weights (rotated through > 512 MB
 * so that the Infinity Cache cannot serve them).  kind: 0 plain, 1 GLU pair, 2 MoE accumulate over
 * n_tasks slots; act_mode: 0 ready Q8_K, 1 f32 (quantised in the prologue), 2 f32 + RMSNorm.
 * force_* > 0 override the launch planner (lanes per row, rows per lane group, column steps). */
int dsk_bench_gemv(dsk_ctx* ctx, int quant, int rows, int n, int n_tasks, int kind, int act_mode,
                   int force_lpr, int force_R, int force_U, int target_wgs, int iters,
                   double* us_per_launch, double* bytes_per_launch);

/* Diagnostics: 8 wall-clock stamps (100 MHz ticks: entry, staged, phase A done, hand-off passed, hidden vectors staged,
   rows done, exit, unused) per workgroup of the LAST fused routed-expert launch; needs the option "timeline". */
int dsk_model_get_timeline(dsk_model* m, int kind, unsigned long long* out, int n_wgs);  /* kind: 0 first-stage projections, 1 per-head attention, 2 wo, 3 shared expert w1/w3 (rider), 4 fused routed experts, 5 router, 6 MLA long-context scores / values; needs the option "timeline" (dsk_model_set_option); a region holds the first 1024 workgroups of a launch; tools/timeline.py names the stamps */
int dsk_model_get_moe_timeline(dsk_model* m, unsigned long long* out, int n_wgs);

/* Router (F32 GEMV + rmsnorm prologue) + moe_gate micro-benchmark on synthetic weights; flags: 0 = the
   production kernel, 1 = skip the gate, 2 = skip the weight stream, 4 = skip the norm (bitwise or). */
int dsk_bench_router(dsk_ctx* ctx, int n_routed, int dim, int ksplit, int flags, int iters, double* us_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* DSK_H */
