// kernels_gemv.hip -- batch-1 weight-streaming GEMV kernels for gfx950 (wave64).
//
// Replaces the reference's five _matmul overloads and matmul_expert
// (src/infer.cpp:121-379, 423-469) and the vec_dot kernels (src/quant.cpp:434-614, 666-783).
//
// K-quants are W2A8 / W3A8 exactly like the reference: activations are Q8_K
// (int8 + per-256 scale + per-16 sums, src/quant.cpp:616-653), the sub-block dot products
// are integer (v_dot4_i32_i8) and only the per-super-block scaling is float.
//
// Structure (HBM-bound, no MFMA -- one token, so there is no N dimension):
//   * one launch = up to GEMV_MAX_TASKS "tasks" (matrix x vector jobs: e.g. wq_a || wkv_a, or the
//     8 routed expert slots + the shared expert), each owning a contiguous range of workgroups;
//   * workgroups are persistent: the grid is sized to the machine and every workgroup walks
//     its task's row groups with a grid stride, so the per-workgroup prologue is amortised;
//   * prologue = stage the activation vector in LDS once: copy a ready Q8_K vector, or quantise
//     an f32 vector, or RMSNorm (src/infer.cpp:601-611) + quantise -- this fuses the reference's
//     rmsnorm() and quantize_row_q8_K_ref() calls into their consumer;
//   * the unit of work ("item") is 16 contiguous bytes of the qs plane = 64 weights: one
//     global_load_dwordx4 per lane, adjacent lanes adjacent 16 B => whole 128-B lines;
//     LPR lanes cooperate on one row, a wave holds 64/LPR rows x R row groups, and U column
//     steps are issued back to back so every lane keeps R*U 16-byte loads in flight;
//   * epilogues: store, residual add (src/infer.cpp:832-834), SiLU/GELU-GLU pair
//     (src/infer.cpp:859-872) and the MoE accumulate x += w_k * (W2_k . h_k) over the routed
//     slots in k order, then the shared expert (src/infer.cpp:873-878, 899-903);
//   * all reductions are fixed-order shuffles: no atomics, bit-reproducible run to run.
#include "dsk_internal.h"
#include "gemv_device.h"
#include "tile_gemv.h"

#ifndef GEMV_EXACT
#define GEMV_EXACT 2  // 0: only the generic chunk loop in gemv_body; 1: straight-line form for the 7168-wide GLU rows; 2: also for the
                      // plain 7168-wide rows at 16 lanes (the classifier) and for wo's 16384-wide rows at 64 lanes (A/B builds)
#endif
#ifndef HEAD_EXACT
#define HEAD_EXACT 1  // -DHEAD_EXACT=0: the generic row loop in the per-head attention kernel (A/B builds)
#endif

// ------------------------------------------------------------------------------------
// the kernel.  A workgroup belongs to one activation group (tasks sharing an input vector), stages that
// vector once, and walks its even share of the group's concatenated rows.
// ------------------------------------------------------------------------------------
// (a device function: router_shared_kernel below runs it for the shared expert next to the router's workgroups;
// bid = the workgroup's index within THIS gemv's grid, h_pre_scale > 0 = a known rmsnorm scale for the hinted staging)
template <int QT, int R, int U, bool GLU, int NW>
DEV void gemv_body(const GemvLaunch* __restrict__ Lp, const void* h_a0, const void* h_a1, const void* h_a2, int h_n, int h_mode,
                   float h_eps, int h_gwgs, int h_gstride, const int bid, const float h_pre_scale) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ bool comb_last;
  const GemvLaunch& L = *Lp;
  constexpr bool KQ = QT == DSK_QUANT_Q2_K || QT == DSK_QUANT_Q3_K;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned long long t_entry = wall_clock64();
  // two descriptor fields are requested NOW (cold lines: the group table and the launch geometry), so that a
  // hinted launch waits for them and for its activation vector at the same time
  const int n_groups = L.n_groups;
  const int lpr_log2 = L.lpr_log2;
  const bool hinted = h_n > 0;  // one activation group: its source came with the kernel arguments
  if (hinted) {
    ActSrc S;
    S.act_mode = h_mode; S.n = h_n; S.eps = h_eps; S.pre_scale = h_pre_scale;
    S.a_qs = static_cast<const int8_t*>(h_a0); S.a_d = static_cast<const float*>(h_a1); S.a_bsums = static_cast<const int16_t*>(h_a2);
    S.a_f32 = static_cast<const float*>(h_a0); S.norm_w = static_cast<const float*>(h_a1);
    if (h_gwgs > 0) S.a_f32 += (size_t)(bid / h_gwgs) * h_gstride;  // equal groups, equally spaced f32 vectors
    if (KQ) stage_q8<QT == DSK_QUANT_Q2_K, NW>(S, smem, tid, scratch);
    else stage_f32<NW>(S, reinterpret_cast<float*>(smem), tid, scratch);
  }
  const int RPW = 64 >> lpr_log2;
  const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
  const int RG = NW * RPW * R;  // rows per workgroup step
  unsigned long long* tl = L.timeline && bid < DSK_TL_WGS ? L.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = t_entry;

  int t0 = 0, t1 = 1, wi, nwg, head = 0, grp_idx = 0;
  const bool bd = L.bd_heads > 0;
  if (bd) {  // block-diagonal stack: workgroup -> head
    head = bid / L.bd_wgs;
    wi = bid - head * L.bd_wgs;
    nwg = L.bd_wgs;
  } else {
    int g = 0, wg0 = 0;
#pragma unroll
    for (int k = 0; k < GEMV_MAX_TASKS - 1; ++k)
      if (k + 1 < n_groups && bid >= L.grp_wg_end[k]) { g = k + 1; wg0 = L.grp_wg_end[k]; }
    t0 = L.grp_t0[g];
    t1 = L.grp_t0[g + 1];
    wi = bid - wg0;
    nwg = L.grp_wg_end[g] - wg0;
    grp_idx = g;
  }
  auto task_of = [&](int ti) {
    GemvTask T = L.t[ti];
    if (bd) {  // re-base the one task to this workgroup's head
      const size_t per = (size_t)T.rows * T.n;
      if (QT == DSK_QUANT_Q2_K) {
        T.qs += head * (per / 256 * 64); T.sc += head * (per / 256 * 16); T.dm += head * (per / 256 * 4);
      } else if (QT == DSK_QUANT_Q3_K) {
        T.qs += head * (per / 256 * 64); T.hm += head * (per / 256 * 32); T.sc += head * (per / 256 * 12); T.dm += head * (per / 256 * 2);
      } else {
        T.qs += head * per * FTraits<QT>::ESZ;
        // reference indexing: expert_index * cdiv(d,b0)*cdiv(n,b1) (src/infer.cpp:437-438)
        if (T.scale) T.scale += (size_t)head * ((T.rows + L.b0 - 1) / L.b0) * ((T.n + L.b1 - 1) / L.b1);
      }
      T.a_f32 += (size_t)head * T.n;
      T.out += (size_t)head * T.rows;
      T.vrow_begin = 0;
      T.vrow_end = T.rows;
    }
    return T;
  };
  if (!hinted) {
    const GemvTask Ta = task_of(t0);
    if (tl && tid == 0) tl[6] = wall_clock64();
    if (KQ) stage_q8<QT == DSK_QUANT_Q2_K, NW>(Ta, smem, tid, scratch, tl);
    else stage_f32<NW>(Ta, reinterpret_cast<float*>(smem), tid, scratch);
  } else if (tl && tid == 0) {
    tl[6] = wall_clock64();
  }
  if (tl && tid == 0) tl[7] = wall_clock64();
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();
#ifndef DSK_NO_TAPS
  if constexpr (KQ) {
    if (L.tap_qs && !bd && wi == 0)  // parity tap: what this activation group staged
      dump_staged_q8<QT == DSK_QUANT_Q2_K>(smem, L.t[t0].n, L.tap_qs + (size_t)grp_idx * L.tap_stride,
                                          L.tap_d + (size_t)grp_idx * (L.tap_stride >> 8), tid, NW * 64);
  }
#endif

  // this workgroup's share of the group's virtual rows, in multiples of part_unit
  // sharded experts: only the present tasks' rows count.  GLU instantiations only (the sharded w1/w3 launch is one): in the
  // plain ones the code below folds away (with it in, wo and the first-stage projections were 0.25 us slower each)
  const bool compact = GLU && L.compact_absent && !bd && L.comb_x == nullptr;
  int vtotal = bd ? L.t[0].rows : L.t[t1 - 1].vrow_end;
  int c_rows = 0, c_base = 0;  // lane k: rows of the group's task k if its expert lives here, and their first compacted row
  if (compact) {
    // lane k looks at task k: every expert id is read in ONE round trip (task by task it is a dependent load each: the
    // compacted launch took 21 us instead of 18)
    if (lane < t1 - t0) {
      const GemvTask& Tk = L.t[t0 + lane];
      bool here = true;
      if (Tk.e_qs != 0) {
        const int le = (Tk.expert_ids ? Tk.expert_ids[Tk.slot] : Tk.slot) - Tk.expert_base;
        here = le >= 0 && le < Tk.local_experts;
      }
      c_rows = here ? Tk.vrow_end - Tk.vrow_begin : 0;
    }
    vtotal = 0;
#pragma unroll
    for (int j = 0; j < GEMV_MAX_TASKS; ++j) {
      const int rj = __builtin_amdgcn_readlane(c_rows, j);
      if (lane == j) c_base = vtotal;
      vtotal += rj;
    }
  }
  const int unit = L.part_unit;
  const long long units = (vtotal + unit - 1) / unit;
  const int r_lo = (int)(units * wi / nwg) * unit;
  int r_hi = (int)(units * (wi + 1) / nwg) * unit;
  if (r_hi > vtotal) r_hi = vtotal;
  const bool comb = !GLU && L.comb_x != nullptr;
  const uint8_t* lds_lane = smem + sub * ITEM_LDS;
  const int q = sub & 3;
  bool first = true;

  for (int ti = t0; ti < t1; ++ti) {
    int vb, ve;
    if (compact) {  // (before the task's descriptor is touched: an absent task costs two lane reads)
      vb = __builtin_amdgcn_readlane(c_base, ti - t0);
      ve = vb + __builtin_amdgcn_readlane(c_rows, ti - t0);
      if (r_lo >= ve || r_hi <= vb) continue;
    }
    const GemvTask T = task_of(ti);
    if (!compact) { vb = T.vrow_begin; ve = T.vrow_end; }
    const int lo = (r_lo > vb ? r_lo : vb) - vb, hi = (r_hi < ve ? r_hi : ve) - vb;
    if (lo >= hi) continue;
    const WPtr P = resolve(T);
    if (!P.present) {  // the expert lives on another GPU
      if (L.zero_absent && !GLU)
        for (int rr = lo + tid; rr < hi; rr += NW * 64) T.out[rr] = 0.f;
      continue;
    }
    const KQRsrc B = kq_rsrc<QT, GLU>(P);
    const int nb = T.n >> 8;
    // (wo of DeepSeek-V3 - 28 rows per workgroup = two steps of 16 - with BOTH row groups requested at once and multiplied
    // as they arrive, straight-line: 12.2 -> 12.7 us; like every deeper burst tried, slower;
    // round 3, the two groups as ONE two-buffer stream - group 2's first buffer requested after group 1's first has been
    // multiplied, never more than two in flight -: 11.8 -> 11.9-12.0 us, 128 VGPRs; the drain between the groups is not what
    // the launch waits for)
    for (int base = lo; base < hi; base += RG) {
      int row[R];
      bool valid[R];
      const int row0 = base + wave * (RPW * R);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int rr = row0 + r * RPW + rloc;
        valid[r] = rr < hi;
        row[r] = valid[r] ? rr : hi - 1;
      }
      const bool has_rows = row0 < hi;
      float acc[R], acc2[R];
      if (has_rows) {
        if constexpr (KQ) {
          int rowblk[R];
#pragma unroll
          for (int r = 0; r < R; ++r) rowblk[r] = row[r] * nb + (sub >> 2);
          // (the plain forms are compiled into the ONE variant that runs each shape - U = 8: the classifier, U = 4: wo - :
          // instantiated in every plain variant they cost the first-stage projections' U = 2 kernel 0.85 us per launch)
          if constexpr (GEMV_EXACT > 1 && QT == DSK_QUANT_Q2_K && !GLU && R == 1 && NW == 16 && U == 4) {
            if (lpr_log2 == 6 && nb == 64) rows_dot_kq_exact<QT, 1, GLU, 4, 6>(B, sub, q, rowblk, lds_lane, acc, acc2);
            else rows_dot_kq<QT, R, U, GLU>(B, nb * 4, sub, lpr_log2, q, rowblk, lds_lane, acc, acc2);
          } else if constexpr (GEMV_EXACT && QT == DSK_QUANT_Q2_K && (GLU || (GEMV_EXACT > 1 && U == 8)) && R == 1 && NW == 16) {
            // 7168-wide rows at 16 lanes each (dense w1/w3, the shared expert's rider, the two-launch experts): the
            // software-pipelined straight-line form, same bits (gemv_device.h rows_dot_kq_exact)
            // (round 3: the generic loop costs ~10 of its ~72 VALU instructions per item on bounds, ragged-step selects and
            // 64-bit record addressing, plus ~14 scalar instructions and two branches per step; a launch whose SIMDs are busy
            // issuing - the classifier: SQ_ACTIVE_INST_VALU 68 % of wave time, profiles/r03_pmc_gemv.txt - gets faster without
            // them: lm_head 59.4 -> 55.9 us alone.  Column steps are consumed in the same order: same bits.)
            if (lpr_log2 == 4 && nb == 28) rows_dot_kq_exact<QT, 1, GLU, 7, 4>(B, sub, q, rowblk, lds_lane, acc, acc2);
            else rows_dot_kq<QT, R, U, GLU>(B, nb * 4, sub, lpr_log2, q, rowblk, lds_lane, acc, acc2);
          } else {
            rows_dot_kq<QT, R, U, GLU>(B, nb * 4, sub, lpr_log2, q, rowblk, lds_lane, acc, acc2);
          }
        } else {
          rows_dot_f<QT, R, U, GLU>(P, T.n, L.b0, L.b1, lpr_log2, lane, row, smem, acc, acc2);
        }
      }
      if (tl && tid == 0 && first) tl[2] = wall_clock64();
      first = false;
      if (comb) {
        // ---- fused MoE combine: slot vectors go out write-through (sc1), then one arrival per task ----
        const int g = base / RG;  // part_unit == RG: a row group is computed by exactly one workgroup per task
        if (has_rows && sub == 0) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (valid[r]) __hip_atomic_store(T.out + row[r], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          const unsigned old = __hip_atomic_fetch_add(L.comb_counter + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          comb_last = old == (unsigned)L.n_tasks - 1;
          if (comb_last) {
            FINISHER_ACQUIRE();
            __hip_atomic_store(L.comb_counter + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
          }
        }
        __syncthreads();
        if (comb_last) {  // all slots of these rows have landed: x += w_k * out_k (k order), then + shared
          for (int rr = base + tid; rr < min(T.rows, base + RG); rr += NW * 64) {
            float xv = L.comb_x[rr];
            for (int tj = 0; tj < L.n_tasks; ++tj) {
              const float v = __hip_atomic_load(L.t[tj].out + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (L.t[tj].accum_w) xv = fmaf(v, *L.t[tj].accum_w, xv);  // src/infer.cpp:874-877
              else xv += v;                                             // src/infer.cpp:900-903
            }
            L.comb_x[rr] = xv;
          }
        }
        continue;
      }
      if (!has_rows || sub != 0) continue;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!valid[r]) continue;
        float* o = T.out + row[r];
        if (GLU) *o = act_fn(acc[r], L.act) * acc2[r];  // src/infer.cpp:859-872
        else if (T.epilogue == EPI_ADD) *o += acc[r];   // residual add, src/infer.cpp:832-834,928-930
        else *o = acc[r];
      }
    }
  }
  if (tl && tid == 0) tl[3] = wall_clock64();
}

template <int QT, int R, int U, bool GLU, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(const GemvLaunch* __restrict__ Lp, const void* h_a0, const void* h_a1,
                                                         const void* h_a2, int h_n, int h_mode, float h_eps, int h_gwgs,
                                                         int h_gstride) {
  gemv_body<QT, R, U, GLU, NW>(Lp, h_a0, h_a1, h_a2, h_n, h_mode, h_eps, h_gwgs, h_gstride, (int)blockIdx.x, 0.f);
}


// ------------------------------------------------------------------------------------
// Round 6: a small plain Q2_K launch whose every workgroup stages the SAME rmsnorm'ed vector (the first-stage projections: wq_a and
// wkv_a, 2112 rows of 7168 over 256 workgroups = 19 KB of weights each) with ALL of a workgroup's weights requested AHEAD of the
// staging.  Round 1 measured that as slower twice - the requests left BEFORE the loads of x (a CU returns its vector memory
// operations in order: x queued behind 19 KB from HBM), and the staging's barriers were fences (`__syncthreads()` waits for every
// outstanding load, i.e. for the weights).  Here the order is: x and the norm weights first, then the descriptor (warm in L2: the
// previous launch's spare workgroups read it), then the weights - at most two (task, row group) pairs per wave - then the staging
// with LDS-only barriers (gemv_device.h lds_barrier), then the multiplies on registers that have been travelling since entry.
// Same device functions, same column-step order, same lane trees as gemv_body: same bits.  The planner selects it (GemvLaunch::
// ahead) for one activation group of <= 2 plain tasks, rows of <= 2 column steps, a share of <= one row group per workgroup.
// ------------------------------------------------------------------------------------
// (Straight-line on purpose: row length and lanes per row are template constants and a (task, row group) pair that does not exist
// for a wave reads through a NULL buffer descriptor - zeros, no memory traffic - instead of being branched around.  With a load
// inside a branch hipcc's waitcnt pass must assume either path at the join and waits for EVERYTHING before x is touched.)
template <int ITEMS, int LL>
__global__ __launch_bounds__(1024) void gemv_ahead_kernel(const GemvLaunch* __restrict__ Lp, const float* __restrict__ x, const float* __restrict__ norm_w,
                                                         float eps) {
  constexpr int QT = DSK_QUANT_Q2_K, NW = 16, KB1 = 32 / NW, nb = ITEMS / 4, n = nb * 256, ITS = (ITEMS + (1 << LL) - 1) >> LL;
  constexpr int RPW = 64 >> LL;
  static_assert(ITS <= 2 && nb <= 32, "gemv_ahead_kernel: rows of at most two column steps");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  const GemvLaunch& L = *Lp;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, bid = blockIdx.x;
  const unsigned long long t_entry = wall_clock64();
  // ---- 1. this wave's blocks of x and of the norm weights (stage_q8's single-pass path, first half) ----
  f32x4 t[KB1], wv[KB1];
#pragma unroll
  for (int k = 0; k < KB1; ++k) {
    const int b = wave + NW * k < nb ? wave + NW * k : nb - 1;  // (a wave without a k-th block re-reads the last one and drops it)
    t[k] = *reinterpret_cast<const f32x4*>(x + b * 256 + lane * 4);
    wv[k] = *reinterpret_cast<const f32x4*>(norm_w + b * 256 + lane * 4);
  }
  // ---- 2. the descriptor, this workgroup's share, and the weights of (at most) two (task, row group) pairs ----
  const int rloc = lane >> LL, sub = lane & ((1 << LL) - 1), q = sub & 3;
  unsigned long long* tl = L.timeline && bid < DSK_TL_WGS ? L.timeline + (size_t)bid * 8 : nullptr;
  // (the stamps are kept in registers and stored at the end: a store in flight makes hipcc wait for every outstanding operation at
  // the next use of a loaded register - it would wait for the weights before it touched x)
  unsigned long long t_staged = 0, t_bar = 0, t_rows = 0;
  const int n_tasks = L.grp_t0[1], nwg = L.grp_wg_end[0];
  const int vtotal = L.t[n_tasks - 1].vrow_end, unit = L.part_unit;
  // (gemv_body's partition in 32-bit arithmetic: units * workgroups < 2^31 for a launch this small, so the quotients are the same)
  const unsigned units = (unsigned)((vtotal + unit - 1) / unit);
  const int r_lo = __builtin_amdgcn_readfirstlane((int)(units * (unsigned)bid / (unsigned)nwg) * unit);
  int r_hi = __builtin_amdgcn_readfirstlane((int)(units * (unsigned)(bid + 1) / (unsigned)nwg) * unit);
  if (r_hi > vtotal) r_hi = vtotal;
  ChunkKQ<QT, 1, 2, false> ch[2];
  int p_row[2];
  bool p_valid[2];
  float* p_out[2];
  int p_epi[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const GemvTask& T = L.t[j < n_tasks ? j : 0];
    const int vb = T.vrow_begin, ve = T.vrow_end;
    const int lo = (r_lo > vb ? r_lo : vb) - vb, hi = (r_hi < ve ? r_hi : ve) - vb;
    // (the planner guarantees hi - lo <= RG: one row group per task and workgroup)
    const int row0 = lo + wave * RPW, rr = row0 + rloc;
    const bool has = j < n_tasks && row0 < hi;  // wave-uniform
    p_valid[j] = has && rr < hi;
    p_row[j] = p_valid[j] ? rr : (hi > 0 ? hi - 1 : 0);
    p_out[j] = T.out;
    p_epi[j] = T.epilogue;
    KQRsrc B;
    B.qs = make_rsrc_n(T.qs, has);
    B.sc = make_rsrc_n(T.sc, has);
    B.dm = make_rsrc_n(T.dm, has);
    B.hm = B.qs2 = B.sc2 = B.dm2 = B.hm2 = B.qs;
    const int rowblk[1] = {has ? p_row[j] * nb + (sub >> 2) : 0};
    load_chunk_kq<QT, 1, 2, false>(ch[j], B, ITS, ITEMS, sub, LL, q, rowblk, 0);
  }
  // ---- 3. rmsnorm (src/infer.cpp:601-611) + Q8_K of x into LDS: stage_q8's single-pass path, second half, LDS-only barriers ----
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < KB1; ++k) {
    if (wave + NW * k < nb) {
      ss = fmaf(t[k].x, t[k].x, ss);
      ss = fmaf(t[k].y, t[k].y, ss);
      ss = fmaf(t[k].z, t[k].z, ss);
      ss = fmaf(t[k].w, t[k].w, ss);
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) scratch[wave] = ss;
  lds_barrier();
  const float total = scratch_total<NW>(scratch);
  const float scale = 1.0f / sqrtf(total / (float)n + eps);
#pragma unroll
  for (int k = 0; k < KB1; ++k) {
    const int b = wave + NW * k;
    if (b < nb) {
      float v[4] = {t[k].x * scale * wv[k].x, t[k].y * scale * wv[k].y, t[k].z * scale * wv[k].z, t[k].w * scale * wv[k].w};
      q8k_block_lds<LAY_Q2>(v, lane, smem + (size_t)b * 4 * ITEM_LDS);
    }
  }
  if (tl) t_staged = wall_clock64();
  lds_barrier();
  if (tl) t_bar = wall_clock64();
#ifndef DSK_NO_TAPS
  if (L.tap_qs && bid == 0) {  // parity tap: what this launch staged (taps are armed for eager block runs only)
    dump_staged_q8<LAY_Q2>(smem, n, L.tap_qs, L.tap_d, tid, NW * 64);
  }
#endif
  // ---- 4. the multiplies (registers requested in 2.), the lane tree, the store ----
  const uint8_t* lds_lane = smem + sub * ITEM_LDS;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float acc[1] = {0.f}, acc2[1] = {0.f};
    compute_chunk_kq<QT, 1, 2, false>(ch[j], ITS, ITEMS, sub, LL, q, 0, lds_lane, acc, acc2);
    const float r = lanes_sum(acc[0], LL);
    if (tl && j == 0) t_rows = wall_clock64();
    if (p_valid[j] && sub == 0) {
      float* o = p_out[j] + p_row[j];
      if (p_epi[j] == EPI_ADD) *o += r;  // residual add, src/infer.cpp:832-834
      else *o = r;
    }
  }
  if (tl && tid == 0) {
    tl[0] = t_entry; tl[6] = t_staged; tl[7] = tl[1] = t_bar; tl[2] = t_rows; tl[3] = wall_clock64();
  }
}
// The same for a plain one-task launch on a READY Q8_K vector with rows of four column steps (wo: 7168 rows of 16384 over 256
// workgroups = two row groups of 16 each): the first row group's first two steps (43 KB per workgroup: what a CU keeps in flight)
// leave behind the loads of the vector and ahead of its copy into LDS records; the second half of the group is requested when the
// records are in place, the remaining groups go through rows_dot_kq_exact like before.  Column steps in order: same bits.
template <int ITEMS, int LL>
__global__ __launch_bounds__(1024) void gemv_ahead_q8_kernel(const GemvLaunch* __restrict__ Lp, const int8_t* __restrict__ a_qs, const float* __restrict__ a_d,
                                                            const int16_t* __restrict__ a_bsums) {
  constexpr int QT = DSK_QUANT_Q2_K, NW = 16, nb = ITEMS / 4, n = nb * 256, ITS = ITEMS >> LL, RPW = 64 >> LL, RG = NW * RPW;
  static_assert((ITS == 4 || ITS == 1) && (ITEMS & ((1 << LL) - 1)) == 0 && (n >> 4) <= NW * 64, "gemv_ahead_q8_kernel: four (or one) exact column steps, one 16-byte run per thread");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const GemvLaunch& L = *Lp;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, bid = blockIdx.x;
  const unsigned long long t_entry = wall_clock64();
  // ---- 1. the vector: one 16-byte run (a sub-block), its sum, and for the first nb * 4 threads a block scale ----
  // (every thread loads its block's scale and every thread stores it: the four sub-block threads of a record write the same two
  // floats.  A load whose only use sits in a branch is sunk INTO the branch by hipcc, behind the weights' requests)
  const int ri = tid < (n >> 4) ? tid : 0;
  const u32x4 run = reinterpret_cast<const u32x4*>(a_qs)[ri];
  const int bs = a_bsums[ri];
  const float dv = a_d[ri >> 4];
  // ---- 2. the descriptor, this workgroup's share, the first row group's steps 0 and 1 ----
  const int rloc = lane >> LL, sub = lane & ((1 << LL) - 1), q = sub & 3;
  unsigned long long* tl = L.timeline && bid < DSK_TL_WGS ? L.timeline + (size_t)bid * 8 : nullptr;
  unsigned long long t_staged = 0, t_bar = 0, t_rows = 0;
  const GemvTask& T = L.t[0];
  const int nwg = L.grp_wg_end[0], vtotal = T.vrow_end, unit = L.part_unit;
  const unsigned units = (unsigned)((vtotal + unit - 1) / unit);
  const int lo = __builtin_amdgcn_readfirstlane((int)(units * (unsigned)bid / (unsigned)nwg) * unit);
  int hi = __builtin_amdgcn_readfirstlane((int)(units * (unsigned)(bid + 1) / (unsigned)nwg) * unit);
  if (hi > vtotal) hi = vtotal;
  WPtr P;
  P.qs = T.qs; P.sc = T.sc; P.dm = T.dm; P.hm = nullptr; P.qs2 = P.sc2 = P.hm2 = P.dm2 = nullptr; P.scale = P.scale2 = nullptr; P.present = true;
  const KQRsrc B = kq_rsrc<QT, false>(P);
  ChunkKQ<QT, 1, 2, false> ca, cb;
  int row, rowblk[1];
  bool valid;
  {
    const int rr = lo + wave * RPW + rloc;
    const bool has = lo + wave * RPW < hi;
    valid = rr < hi;
    row = valid ? rr : (hi > 0 ? hi - 1 : 0);
    rowblk[0] = row * nb + (sub >> 2);
    KQRsrc B0 = B;
    B0.qs = make_rsrc_n(T.qs, has); B0.sc = make_rsrc_n(T.sc, has); B0.dm = make_rsrc_n(T.dm, has);
    load_chunk_kq<QT, 1, 2, false>(ca, B0, ITS, ITEMS, sub, LL, q, rowblk, 0);
  }
  // ---- 3. the vector into item records (stage_q8, the ready-Q8_K path for LAY_Q2) ----
  if (tid < (n >> 4)) {
    const int b = tid >> 4, j = tid & 15, h = j >> 3, sidx = (j >> 1) & 3, lh = j & 1;
    uint8_t* rec = smem + (size_t)(b * 4 + 2 * h + lh) * ITEM_LDS;
    *reinterpret_cast<u32x4*>(rec + sidx * 16) = run;
    rec[64 + sidx] = (uint8_t)(bs >> 8);
    rec[68 + sidx] = (uint8_t)(bs & 0xff);
    float* mrec = reinterpret_cast<float*>(rec + 72);
    mrec[0] = dv * 0.0625f;
    mrec[1] = dv;
  }
  if (tl) t_staged = wall_clock64();
  lds_barrier();
  if (tl) t_bar = wall_clock64();
  // ---- 4. the first row group: steps 2 and 3 requested now, then the four steps multiplied in order ----
  const uint8_t* lds_lane = smem + sub * ITEM_LDS;
  float* const out = T.out;
  const int epi = T.epilogue;
  {
    const bool has = lo + wave * RPW < hi;
    KQRsrc B0 = B;
    B0.qs = make_rsrc_n(T.qs, has); B0.sc = make_rsrc_n(T.sc, has); B0.dm = make_rsrc_n(T.dm, has);
    load_chunk_kq<QT, 1, 2, false>(cb, B0, ITS, ITEMS, sub, LL, q, rowblk, 2);
    float acc[1] = {0.f}, acc2[1] = {0.f};
    compute_chunk_kq<QT, 1, 2, false>(ca, ITS, ITEMS, sub, LL, q, 0, lds_lane, acc, acc2);
    compute_chunk_kq<QT, 1, 2, false>(cb, ITS, ITEMS, sub, LL, q, 2, lds_lane, acc, acc2);
    const float r = lanes_sum(acc[0], LL);
    if (tl) t_rows = wall_clock64();
    if (valid && sub == 0) {
      float* o = out + row;
      if (epi == EPI_ADD) *o += r;  // residual add, src/infer.cpp:832-834
      else *o = r;
    }
  }
  // ---- 5. the remaining row groups ----
  for (int base = lo + RG; base < hi; base += RG) {
    const int row0 = base + wave * RPW, rr = row0 + rloc;
    if (row0 >= hi) continue;
    const bool v2 = rr < hi;
    const int rw = v2 ? rr : hi - 1;
    const int rb[1] = {rw * nb + (sub >> 2)};
    float acc[1], acc2[1];
    rows_dot_kq_exact<QT, 1, false, ITS, LL>(B, sub, q, rb, lds_lane, acc, acc2);
    if (v2 && sub == 0) {
      float* o = out + rw;
      if (epi == EPI_ADD) *o += acc[0];
      else *o = acc[0];
    }
  }
  if (tl && tid == 0) {
    tl[0] = t_entry; tl[6] = t_staged; tl[7] = tl[1] = t_bar; tl[2] = t_rows; tl[3] = wall_clock64();
  }
}
static bool gemv_ahead_q8_ok(const GemvLaunch& h) {
  if (h.tiled || h.quant != DSK_QUANT_Q2_K || h.glu || h.R != 1 || h.NW != 16 || h.n_groups != 1 || h.bd_heads != 0 || h.n_tasks != 1) return false;
  if (h.comb_x || h.comb_geometry || h.compact_absent || h.zero_absent || h.tap_qs) return false;
  const GemvTask& T = h.t[0];
  // the instantiations: DeepSeek-V3's wo (rows of 16384: four steps at 64 lanes per row) and V2-Lite's (rows of 2048: one step at 32)
  const bool v3 = T.n == 16384 && h.lpr_log2 == 6 && h.U == 4, v2l = T.n == 2048 && h.lpr_log2 == 5 && h.U == 1;
  return T.act_mode == ACT_Q8 && (v3 || v2l) && T.e_qs == 0 && !T.accum_w && h.grid == h.grp_wg_end[0] && T.vrow_begin == 0;
}
// The MLA path's second-stage launch (wq_rope_b || wc on rmsnorm(q_a): 73 728 rows of 1536 over 255 workgroups = 2-3 row groups of
// 128 rows each, + the latent's cache write as the last workgroup) in the same order: q_a and its norm weights, the descriptor, the
// FIRST (task, row group) pair's rows - whole rows: three column steps in one chunk -, the staging with LDS-only barriers, that
// pair's multiplies; the remaining pairs go through rows_dot_kq like in gemv_body.  Same functions, same step order: same bits.
template <int ITEMS, int LL>
__global__ __launch_bounds__(1024) void gemv_kvwrite_ahead_kernel(const GemvLaunch* __restrict__ Lp, const float* __restrict__ x, const float* __restrict__ norm_w,
                                                                 float eps, const MlaKvArgs kv, const StepParams* __restrict__ sp) {
  if (blockIdx.x == gridDim.x - 1) {
    rd::mla_kv_write_body(kv, sp, threadIdx.x, 1024);
    return;
  }
  constexpr int QT = DSK_QUANT_Q2_K, NW = 16, KB1 = 32 / NW, nb = ITEMS / 4, n = nb * 256, ITS = (ITEMS + (1 << LL) - 1) >> LL;
  constexpr int RPW = 64 >> LL, RG = NW * RPW;
  static_assert(ITS <= 4 && nb <= 32, "gemv_kvwrite_ahead_kernel: whole rows in one chunk of four column steps");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  const GemvLaunch& L = *Lp;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, bid = blockIdx.x;
  // ---- 1. this wave's blocks of the vector and of the norm weights ----
  f32x4 t[KB1], wv[KB1];
#pragma unroll
  for (int k = 0; k < KB1; ++k) {
    const int b = wave + NW * k < nb ? wave + NW * k : nb - 1;
    t[k] = *reinterpret_cast<const f32x4*>(x + b * 256 + lane * 4);
    wv[k] = *reinterpret_cast<const f32x4*>(norm_w + b * 256 + lane * 4);
  }
  // ---- 2. the descriptor, this workgroup's share, the first pair's rows ----
  const int rloc = lane >> LL, sub = lane & ((1 << LL) - 1), q = sub & 3;
  const int n_tasks = L.grp_t0[1], nwg = L.grp_wg_end[0];
  const int vtotal = L.t[n_tasks - 1].vrow_end, unit = L.part_unit;
  const unsigned units = (unsigned)((vtotal + unit - 1) / unit);
  const int r_lo = __builtin_amdgcn_readfirstlane((int)(units * (unsigned)bid / (unsigned)nwg) * unit);
  int r_hi = __builtin_amdgcn_readfirstlane((int)(units * (unsigned)(bid + 1) / (unsigned)nwg) * unit);
  if (r_hi > vtotal) r_hi = vtotal;
  const int ti0 = n_tasks > 1 && r_lo >= L.t[1].vrow_begin ? 1 : 0;  // the first task this workgroup's share touches (<= 2 tasks: the planner's check)
  ChunkKQ<QT, 1, 4, false> c0;
  int row_a;
  bool valid_a;
  {
    const GemvTask& T = L.t[ti0];
    const int vb = T.vrow_begin, ve = T.vrow_end;
    const int lo = (r_lo > vb ? r_lo : vb) - vb, hi = (r_hi < ve ? r_hi : ve) - vb;
    const int row0 = lo + wave * RPW, rr = row0 + rloc;
    const bool has = row0 < hi;
    valid_a = has && rr < hi;
    row_a = valid_a ? rr : (hi > 0 ? hi - 1 : 0);
    KQRsrc B;
    B.qs = make_rsrc_n(T.qs, has); B.sc = make_rsrc_n(T.sc, has); B.dm = make_rsrc_n(T.dm, has);
    B.hm = B.qs2 = B.sc2 = B.dm2 = B.hm2 = B.qs;
    const int rowblk[1] = {has ? row_a * nb + (sub >> 2) : 0};
    load_chunk_kq<QT, 1, 4, false>(c0, B, ITS, ITEMS, sub, LL, q, rowblk, 0);
  }
  // ---- 3. rmsnorm + Q8_K into LDS (stage_q8's single-pass path), LDS-only barriers ----
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < KB1; ++k) {
    if (wave + NW * k < nb) {
      ss = fmaf(t[k].x, t[k].x, ss);
      ss = fmaf(t[k].y, t[k].y, ss);
      ss = fmaf(t[k].z, t[k].z, ss);
      ss = fmaf(t[k].w, t[k].w, ss);
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) scratch[wave] = ss;
  lds_barrier();
  const float total = scratch_total<NW>(scratch);
  const float scale = 1.0f / sqrtf(total / (float)n + eps);
#pragma unroll
  for (int k = 0; k < KB1; ++k) {
    const int b = wave + NW * k;
    if (b < nb) {
      float v[4] = {t[k].x * scale * wv[k].x, t[k].y * scale * wv[k].y, t[k].z * scale * wv[k].z, t[k].w * scale * wv[k].w};
      q8k_block_lds<LAY_Q2>(v, lane, smem + (size_t)b * 4 * ITEM_LDS);
    }
  }
  lds_barrier();
#ifndef DSK_NO_TAPS
  if (L.tap_qs && bid == 0) {
    dump_staged_q8<LAY_Q2>(smem, n, L.tap_qs, L.tap_d, tid, NW * 64);
  }
#endif
  // ---- 4. the first pair's multiplies ----
  const uint8_t* lds_lane = smem + sub * ITEM_LDS;
  {
    float acc[1] = {0.f}, acc2[1] = {0.f};
    compute_chunk_kq<QT, 1, 4, false>(c0, ITS, ITEMS, sub, LL, q, 0, lds_lane, acc, acc2);
    const float r = lanes_sum(acc[0], LL);
    if (valid_a && sub == 0) {
      float* o = L.t[ti0].out + row_a;
      if (L.t[ti0].epilogue == EPI_ADD) *o += r;
      else *o = r;
    }
  }
  // ---- 5. the remaining pairs, like gemv_body ----
  for (int ti = ti0; ti < n_tasks; ++ti) {
    const GemvTask& T = L.t[ti];
    const int vb = T.vrow_begin, ve = T.vrow_end;
    const int lo = (r_lo > vb ? r_lo : vb) - vb, hi = (r_hi < ve ? r_hi : ve) - vb;
    if (lo >= hi) continue;
    WPtr P;
    P.qs = T.qs; P.sc = T.sc; P.dm = T.dm; P.hm = nullptr; P.qs2 = P.sc2 = P.hm2 = P.dm2 = nullptr; P.scale = P.scale2 = nullptr; P.present = true;
    const KQRsrc B = kq_rsrc<QT, false>(P);
    for (int base = ti == ti0 ? lo + RG : lo; base < hi; base += RG) {
      const int row0 = base + wave * RPW, rr = row0 + rloc;
      if (row0 >= hi) continue;
      const bool v2 = rr < hi;
      const int rw = v2 ? rr : hi - 1;
      const int rb[1] = {rw * nb + (sub >> 2)};
      float acc[1], acc2[1];
      rows_dot_kq<QT, 1, 4, false>(B, ITEMS, sub, LL, q, rb, lds_lane, acc, acc2);
      if (v2 && sub == 0) {
        float* o = T.out + rw;
        if (T.epilogue == EPI_ADD) *o += acc[0];
        else *o = acc[0];
      }
    }
  }
}
static bool gemv_kvwrite_ahead_ok(const GemvLaunch& h) {
  if (h.tiled || h.quant != DSK_QUANT_Q2_K || h.glu || h.R != 1 || h.U != 4 || h.NW != 16 || h.n_groups != 1 || h.bd_heads != 0 || h.timeline) return false;
  if (h.comb_x || h.comb_geometry || h.compact_absent || h.zero_absent || !(h.ahead & 1) || h.n_tasks > 2) return false;
  const GemvTask& T0 = h.t[0];
  if (T0.act_mode != ACT_F32_NORM || T0.n != 1536 || h.lpr_log2 != 3) return false;  // (the one instantiation: 24 items at 8 lanes per row)
  for (int i = 0; i < h.n_tasks; ++i) {
    const GemvTask& T = h.t[i];
    if (T.n != T0.n || T.a_f32 != T0.a_f32 || T.norm_w != T0.norm_w || T.act_mode != ACT_F32_NORM || T.e_qs != 0 || T.accum_w) return false;
  }
  return h.grid == h.grp_wg_end[0];
}
// the plans gemv_ahead_kernel runs (set by gemv_plan: GemvLaunch::ahead)
static bool gemv_ahead_ok(const GemvLaunch& h) {
  if (h.tiled || h.quant != DSK_QUANT_Q2_K || h.glu || h.R != 1 || h.NW != 16 || h.n_groups != 1 || h.bd_heads != 0) return false;
  if (h.comb_x || h.comb_geometry || h.compact_absent || h.zero_absent || h.n_tasks > 2) return false;
  const GemvTask& T0 = h.t[0];
  // the instantiations: DeepSeek-V3 (rows of 7168: 112 items at 64 lanes per row, two column steps) and V2-Lite (rows of 2048: 32 items
  // at 32 lanes per row, one step)
  const bool v3 = T0.n == 7168 && h.lpr_log2 == 6 && h.U == 2, v2l = T0.n == 2048 && h.lpr_log2 == 5 && h.U == 1;
  if (T0.act_mode != ACT_F32_NORM || !(v3 || v2l)) return false;
  for (int i = 0; i < h.n_tasks; ++i) {
    const GemvTask& T = h.t[i];
    if (T.n != T0.n || T.a_f32 != T0.a_f32 || T.norm_w != T0.norm_w || T.act_mode != ACT_F32_NORM || T.e_qs != 0 || T.accum_w) return false;
  }
  const int RG = 16 * (64 >> h.lpr_log2);
  const long vtotal = h.t[h.n_tasks - 1].vrow_end, units = (vtotal + h.part_unit - 1) / h.part_unit;
  const long per = (units + h.grid - 1) / h.grid * h.part_unit;  // the largest share of a workgroup
  return h.grid == h.grp_wg_end[0] && per <= RG;
}

// ------------------------------------------------------------------------------------
// The router launch with the shared expert riding along (K-quant models, 1 GPU).  The router keeps E / 2 = 128 CUs busy
// for ~10 us (a latency chain: norm, 7 MB of F32 rows, arrival, gate); the shared expert's w1/w3 GLU depends only on
// the FFN-normed x - not on the routing - so its 64 workgroups run HERE, on CUs that would idle, instead of being the
// ninth task of the experts' w1/w3 launch.  Workgroups [0, n_router) run router_body, the others gemv_body over the
// one-task GLU descriptor; they normalise x with the router's own scale routine (same bits as the Q8_K vector the router
// leaves for the routed experts), so the result is bit-identical to the unfused path.
// ------------------------------------------------------------------------------------
// The MLA path's second-stage projection launch (wq_rope_b || wc) with the latent's cache write riding along: the one
// small workgroup of mla_kv_write_kernel needs kv_a only (like this launch needs q_a only), so it runs as the LAST
// workgroup of this launch instead of as a launch of its own (-1 launch per MLA layer).
template <int QT, int U>
__global__ __launch_bounds__(1024) void gemv_kvwrite_kernel(const GemvLaunch* __restrict__ Lp, const void* h_a0, const void* h_a1,
                                                           const void* h_a2, int h_n, int h_mode, float h_eps, const MlaKvArgs kv,
                                                           const StepParams* __restrict__ sp) {
  if (blockIdx.x == gridDim.x - 1) {
    rd::mla_kv_write_body(kv, sp, threadIdx.x, 1024);
    return;
  }
  gemv_body<QT, 1, U, false, 16>(Lp, h_a0, h_a1, h_a2, h_n, h_mode, h_eps, 0, 0, (int)blockIdx.x, 0.f);
}

template <int QT, int U>
__global__ __launch_bounds__(1024) void router_shared_kernel(const RouterArgs a, int n_router, const GemvLaunch* __restrict__ Lp) {
  if ((int)blockIdx.x < n_router) {
    rd::router_body<2>(a, (int)blockIdx.x, n_router);
    return;
  }
  __shared__ float nscratch[16];
  const float scale = rd::router_norm_scale(a, threadIdx.x, nscratch);
  gemv_body<QT, 1, U, true, 16>(Lp, a.x, a.norm_w, nullptr, a.dim, ACT_F32_NORM, a.eps, 0, 0, (int)blockIdx.x - n_router, scale);
}

// the same two launches for Q2_K weights in the tiled layout (tile_gemv.h)
__global__ __launch_bounds__(1024) void gemv_kvwrite_tile_kernel(const GemvLaunch* __restrict__ Lp, const void* h_a0, const void* h_a1,
                                                                const void* h_a2, int h_n, int h_mode, float h_eps, const MlaKvArgs kv,
                                                                const StepParams* __restrict__ sp) {
  if (blockIdx.x == gridDim.x - 1) {
    rd::mla_kv_write_body(kv, sp, threadIdx.x, 1024);
    return;
  }
  gemv_tile_body<false, 16>(Lp, h_a0, h_a1, h_a2, h_n, h_mode, h_eps, 0, 0, (int)blockIdx.x, 0.f);
}
__global__ __launch_bounds__(1024) void router_shared_tile_kernel(const RouterArgs a, int n_router, const GemvLaunch* __restrict__ Lp) {
  if ((int)blockIdx.x < n_router) {
    rd::router_body<2>(a, (int)blockIdx.x, n_router);
    return;
  }
  __shared__ float nscratch[16];
  const float scale = rd::router_norm_scale(a, threadIdx.x, nscratch);
  gemv_tile_body<true, 16>(Lp, a.x, a.norm_w, nullptr, a.dim, ACT_F32_NORM, a.eps, 0, 0, (int)blockIdx.x - n_router, scale);
}

// ------------------------------------------------------------------------------------
// Second-stage projections + attention, per head (MHA path, src/infer.cpp:976-1049).
// q = wq_b . norm(q_a) and kv_b = wkv_b . norm(kv_a) are consumed head by head: head h needs exactly
// rows [h*head_dim, +head_dim) of wq_b and [h*(nope+v), +(nope+v)) of wkv_b.  One 16-wave workgroup per
// head computes those rows straight into LDS and goes on with RoPE, the cache write and attention: q and
// kv_b never travel through HBM, and a whole launch (boundary, descriptor, prologue, tail) disappears.
// Only n_heads CUs stream weights here, but this stage is 18 MB of a 200 MB layer.
// ------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(1024) void head_attn_kernel(const HeadAttnArgs A, const StepParams* __restrict__ sp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ __attribute__((aligned(16))) float q_s[256];
  __shared__ __attribute__((aligned(16))) float kvb_s[512];
  __shared__ __attribute__((aligned(16))) float part[4096];
  __shared__ int last_flag;
  __shared__ float ml_s[2];
  constexpr bool KQ = QT == DSK_QUANT_Q2_K || QT == DSK_QUANT_Q3_K;
  constexpr int NW = 16;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int S = A.n_split;  // workgroups per head (1: the whole context here)
  if ((int)blockIdx.x >= A.a.n_heads * (S > 1 ? S : 1)) {  // prefetch workgroups (HeadAttnArgs::pf_wgs)
    // (one further workgroup per head reading the head's cached K / V rows into its XCD's L2: attention 16.96 -> 16.77 us, token unchanged)
    tail_prefetch(A.pf_p, A.pf_n, tid, 1024);
    return;
  }
  const int h = S > 1 ? (int)blockIdx.x / S : (int)blockIdx.x, split = S > 1 ? (int)blockIdx.x - h * S : 0;
  const AttnMhaArgs& a = A.a;
  unsigned long long* tl = A.timeline && blockIdx.x < DSK_TL_WGS ? A.timeline + (size_t)blockIdx.x * 8 : nullptr;  // (split contexts: H * n_split workgroups)
  if (tl && tid == 0) tl[0] = wall_clock64();
  uint8_t* act_q = smem;
  uint8_t* act_kv = smem + A.lds_q;
  float* att = reinterpret_cast<float*>(smem + A.lds_q + A.lds_kv);
  const int nbq = A.has_q ? A.tq.n >> 8 : 0, nbkv = A.tkv.n >> 8;
  const bool one_pass = KQ && nbq + nbkv <= NW && A.tkv.act_mode == ACT_F32_NORM && (!A.has_q || A.tq.act_mode == ACT_F32_NORM);
  auto tap_staged = [&]() {
#ifndef DSK_NO_TAPS
    if constexpr (KQ) {
      if (A.tap_qs && blockIdx.x == 0) {  // parity tap
        if (QT == DSK_QUANT_Q2_K && A.tiled) {
          if (A.has_q) dump_staged_q8<LAY_TILE>(act_q, A.tq.n, A.tap_qs, A.tap_d, tid, 1024);
          dump_staged_q8<LAY_TILE>(act_kv, A.tkv.n, A.tap_qs + A.tap_stride, A.tap_d + (A.tap_stride >> 8), tid, 1024);
        } else {
          if (A.has_q) dump_staged_q8<QT == DSK_QUANT_Q2_K>(act_q, A.tq.n, A.tap_qs, A.tap_d, tid, 1024);
          dump_staged_q8<QT == DSK_QUANT_Q2_K>(act_kv, A.tkv.n, A.tap_qs + A.tap_stride, A.tap_d + (A.tap_stride >> 8), tid, 1024);
        }
      }
    }
#endif
  };
  // both latents (q_a: 6 blocks, kv_a: 2 blocks for DeepSeek-V3) normed + quantised in ONE pass: wave w owns block w of
  // the concatenation; two sums of squares share one barrier (src/infer.cpp:601-611, quant.cpp:616-653).  Branch-free
  // loads (waves without a block re-read block 0 and drop it), LDS-only barriers.
  struct Latent { f32x4 t, wv; bool mine, is_q; int b; };
  auto latent_request = [&]() {
    Latent Z;
    Z.mine = wave < nbq + nbkv; Z.is_q = wave < nbq;
    const GemvTask& T = Z.is_q ? A.tq : A.tkv;
    Z.b = Z.mine ? (Z.is_q ? wave : wave - nbq) : 0;
    Z.t = *reinterpret_cast<const f32x4*>(T.a_f32 + Z.b * 256 + lane * 4);
    Z.wv = *reinterpret_cast<const f32x4*>(T.norm_w + Z.b * 256 + lane * 4);
    return Z;
  };
  auto latent_finish = [&](const Latent& Z) {
    const GemvTask& T = Z.is_q ? A.tq : A.tkv;
    const f32x4 t = Z.t, wv = Z.wv;
    float ss = fmaf(t.x, t.x, fmaf(t.y, t.y, fmaf(t.z, t.z, t.w * t.w)));
    ss = wave_sum(ss);
    if (lane == 0) scratch[wave] = Z.mine ? ss : 0.f;
    lds_barrier();
    float total = 0.f;
    if (Z.is_q) for (int i = 0; i < nbq; ++i) total += scratch[i];
    else for (int i = nbq; i < nbq + nbkv; ++i) total += scratch[i];
    if (Z.mine) {
      const float scale = 1.0f / sqrtf(total / (float)T.n + T.eps);
      float v[4] = {t.x * scale * wv.x, t.y * scale * wv.y, t.z * scale * wv.z, t.w * scale * wv.w};
      uint8_t* dst = (Z.is_q ? act_q : act_kv) + (size_t)Z.b * 4 * ITEM_LDS;  // (a block is 320 bytes in every staging layout)
      if (QT == DSK_QUANT_Q2_K && A.tiled) q8k_block_lds<LAY_TILE>(v, lane, dst);
      else q8k_block_lds<QT == DSK_QUANT_Q2_K>(v, lane, dst);
    }
    lds_barrier();
  };
  // head h's rows of one projection: 64/LPR rows per wave and step.  (Dealing both projections' rows to the
  // waves as one unit list, or 2 row sets per lane, measured slower: this stage is VALU-bound on its one CU.)
  auto head_rows = [&](const GemvTask& T, const uint8_t* act, int lpr_log2, int nrows, float* out_lds) {
    const int RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    const WPtr P = resolve(T);
    const KQRsrc B = kq_rsrc<QT, false>(P);
    const int nb = T.n >> 8;
    for (int base = wave * RPW; base < nrows; base += NW * RPW) {  // wave-uniform bounds, no barrier inside
      const int lr = base + rloc;
      const bool valid = lr < nrows;
      int row[1] = {h * nrows + (valid ? lr : nrows - 1)};
      float acc[1], acc2[1];
      if constexpr (KQ) {
        int rowblk[1] = {row[0] * nb + (sub >> 2)};
        rows_dot_kq<QT, 1, 4, false>(B, nb * 4, sub, lpr_log2, sub & 3, rowblk, act + sub * ITEM_LDS, acc, acc2);
      } else {
        rows_dot_f<QT, 1, 4, false>(P, T.n, A.b0, A.b1, lpr_log2, lane, row, act, acc, acc2);
      }
      if (sub == 0 && valid) out_lds[lr] = acc[0];
    }
  };
  bool exact = false;
  if constexpr (QT == DSK_QUANT_Q2_K) {
    // Tiled weights (tile_device.h): head h's strips of both projections - head_dim / 16 of wq_b, (nope + v) / 16 of wkv_b, rows of
    // <= 8 blocks: one item per block - as ONE list of steps (strip, block), dealt to the 16 waves as contiguous ranges; a wave
    // requests ALL its steps at once (<= 8), multiplies them as they arrive (matrix pipe: ~42 VALU per step where the dot4 form
    // spent 4.3 us of arithmetic per head on 4 SIMDs), a barrier, then one wave per strip adds the partials (the association
    // of tile_device.h) into q_s / kvb_s.
    if (A.tiled) {
      exact = true;
      latent_finish(latent_request());
      tap_staged();
      if (tl && tid == 0) tl[1] = wall_clock64();
      const TLane TL = tlane_init(lane);
      float* red = reinterpret_cast<float*>(smem + A.lds_q + A.lds_kv) + A.red_off;
      const int nbq_ = A.tq.n >> 8, nbk_ = A.tkv.n >> 8;
      const int sq = a.head_dim >> 4, sk = (a.nope + a.v_dim) >> 4;  // strips per head
      const int JQ = sq * nbq_, J = JQ + sk * nbk_;
      const int j0 = (int)((long long)J * wave / NW), j1 = (int)((long long)J * (wave + 1) / NW);
      const rsrc_t Wq = make_rsrc(A.tq.qs), Wk = make_rsrc(A.tkv.qs);
      constexpr int HS = 7;  // steps a wave holds at once (DeepSeek-V3: 104 steps per head, 6 or 7 per wave)
      TStep S[HS];
      const uint8_t* recv[HS];
      auto locate = [&](int j, bool& isq, int& soff, const uint8_t*& rec) {
        isq = j < JQ;
        if (isq) {
          const int st = j / nbq_, b = j - st * nbq_;
          soff = ((h * sq + st) * nbq_ + b) * TILE_B;
          rec = act_q + (size_t)b * TREC;
        } else {
          const int jj = j - JQ, st = jj / nbk_, b = jj - st * nbk_;
          soff = ((h * sk + st) * nbk_ + b) * TILE_B;
          rec = act_kv + (size_t)b * TREC;
        }
      };
      for (int jb = j0; jb < j1; jb += HS) {
        const int cnt = j1 - jb < HS ? j1 - jb : HS;
        if (cnt == 7) {
#pragma unroll
          for (int u = 0; u < 7; ++u) { bool isq; int soff; locate(jb + u, isq, soff, recv[u]); tstep_load(S[u], isq ? Wq : Wk, TL, soff); }
#pragma unroll
          for (int u = 0; u < 7; ++u) { float ad_ = 0.f, am_ = 0.f; tstep_mac(S[u], recv[u], TL, ad_, am_); red[(size_t)(jb + u) * 64 + lane] = titem_value(ad_, am_, TL); }
        } else if (cnt == 6) {
#pragma unroll
          for (int u = 0; u < 6; ++u) { bool isq; int soff; locate(jb + u, isq, soff, recv[u]); tstep_load(S[u], isq ? Wq : Wk, TL, soff); }
#pragma unroll
          for (int u = 0; u < 6; ++u) { float ad_ = 0.f, am_ = 0.f; tstep_mac(S[u], recv[u], TL, ad_, am_); red[(size_t)(jb + u) * 64 + lane] = titem_value(ad_, am_, TL); }
        } else {
#pragma unroll
          for (int u = 0; u < HS; ++u)
            if (u < cnt) { bool isq; int soff; locate(jb + u, isq, soff, recv[u]); tstep_load(S[u], isq ? Wq : Wk, TL, soff); }
#pragma unroll
          for (int u = 0; u < HS; ++u)
            if (u < cnt) { float ad_ = 0.f, am_ = 0.f; tstep_mac(S[u], recv[u], TL, ad_, am_); red[(size_t)(jb + u) * 64 + lane] = titem_value(ad_, am_, TL); }
        }
      }
      __syncthreads();
      for (int st = wave; st < sq + sk; st += NW) {
        const bool isq = st < sq;
        const float v = isq ? tile_strip_value(red + (size_t)st * nbq_ * 64, nbq_, lane)
                            : tile_strip_value(red + (size_t)(JQ + (st - sq) * nbk_) * 64, nbk_, lane);
        if (lane < 16) {
          if (isq) q_s[st * 16 + lane] = v;
          else kvb_s[(st - sq) * 16 + lane] = v;
        }
      }
    }
  }
#if HEAD_EXACT
  if constexpr (QT == DSK_QUANT_Q2_K) if (!exact) {
    // DeepSeek-V3 shapes (192 q rows of 1536, 256 kv rows of 512, 8 lanes per row: 3 and 1 column steps, all known at
    // compile time): every wave requests ALL its rows of both projections at once - q rows w*8.., kv rows w*8.. and
    // 128 + w*8.., and for waves 0-7 q rows 128 + w*8.. - and multiplies them as they arrive, instead of four dependent
    // load -> wait -> multiply round trips.  Straight-line code: hipcc counts the outstanding loads exactly (vmcnt(N) ladders).  Same
    // lanes per row and column order as head_rows: same bits.
    exact = one_pass && A.has_q && A.lq_log2 == 3 && A.lkv_log2 == 3 && A.tq.n == 1536 && A.tkv.n == 512 && a.head_dim == 192 && a.nope + a.v_dim == 256;
    if (exact) {
      const int rloc = lane >> 3, sub = lane & 7, q = sub & 3;
      // (Requesting the weights BEFORE the latents are normed and quantised - right after the latents' loads, or after
      // they have arrived - was measured: the prologue then ends at 6.0 us instead of 1.9 and the launch takes 15.2 us
      // instead of 12.9.  A wave cannot run ahead of its own requests: issue stalls once the CU's ~48 KB of reads in
      // flight are taken, so the norm / quantise math waits until most of the 140 KB have arrived and the multiplies,
      // which need the staged latents, start late.  Small launches are bound by that per-CU window, not by latency.)
      // What does pay is a request that stays INSIDE that window: the latents' loads first, then the wave's two row groups of wkv_b
      // (43 KB per CU), then the norm / quantise math - the kv rows stream while the latents are staged, the q rows are requested
      // behind the staging and arrive while the kv rows are multiplied.  The multiplies follow the arrival order (kv, kv, q, q);
      // every row is its own chain: same bits.  Worth 0.2 us of 16.3 (the staging slows down by most of what the head start gains: a
      // CU's memory pipeline is a queue; the same requests issued once the latents have ARRIVED, or half of them, measure the same).
      const Latent Z = latent_request();
      const KQRsrc Bq = kq_rsrc<QT, false>(resolve(A.tq)), Bk = kq_rsrc<QT, false>(resolve(A.tkv));
      const int lr0 = wave * 8 + rloc, lr1 = 128 + lr0;
      const int rbq0[1] = {(h * 192 + lr0) * 6 + (sub >> 2)}, rbq1[1] = {(h * 192 + (lr1 < 192 ? lr1 : 191)) * 6 + (sub >> 2)};
      const int rbk0[1] = {(h * 256 + lr0) * 2 + (sub >> 2)}, rbk1[1] = {(h * 256 + lr1) * 2 + (sub >> 2)};
      ChunkKQ<QT, 1, 3, false> cq0, cq1;
      ChunkKQ<QT, 1, 1, false> ck0, ck1;
      load_chunk_kq<QT, 1, 1, false>(ck0, Bk, 1, 8, sub, 3, q, rbk0, 0);
      load_chunk_kq<QT, 1, 1, false>(ck1, Bk, 1, 8, sub, 3, q, rbk1, 0);
      latent_finish(Z);
      tap_staged();
      if (tl && tid == 0) tl[1] = wall_clock64();
      load_chunk_kq<QT, 1, 3, false>(cq0, Bq, 3, 24, sub, 3, q, rbq0, 0);
      if (wave < 8) load_chunk_kq<QT, 1, 3, false>(cq1, Bq, 3, 24, sub, 3, q, rbq1, 0);
      float acc[1] = {0.f}, dummy[1] = {0.f};
      compute_chunk_kq<QT, 1, 1, false>(ck0, 1, 8, sub, 3, q, 0, act_kv + sub * ITEM_LDS, acc, dummy);
      float v = lanes_sum(acc[0], 3);
      if (sub == 0) kvb_s[lr0] = v;
      acc[0] = 0.f;
      compute_chunk_kq<QT, 1, 1, false>(ck1, 1, 8, sub, 3, q, 0, act_kv + sub * ITEM_LDS, acc, dummy);
      v = lanes_sum(acc[0], 3);
      if (sub == 0) kvb_s[lr1] = v;
      acc[0] = 0.f;
      compute_chunk_kq<QT, 1, 3, false>(cq0, 3, 24, sub, 3, q, 0, act_q + sub * ITEM_LDS, acc, dummy);
      v = lanes_sum(acc[0], 3);
      if (sub == 0) q_s[lr0] = v;
      if (wave < 8) {  // (a zero-record descriptor instead of this branch, every wave multiplying 8 steps: rows 5.4 -> 6.4 us, VALU-bound)
        acc[0] = 0.f;
        compute_chunk_kq<QT, 1, 3, false>(cq1, 3, 24, sub, 3, q, 0, act_q + sub * ITEM_LDS, acc, dummy);
        v = lanes_sum(acc[0], 3);
        if (sub == 0) q_s[lr1] = v;
      }
    }
  }
#endif
  if (!exact) {
    if (one_pass) {
      latent_finish(latent_request());
    } else {
      if (A.has_q) {
        if (KQ) stage_q8<QT == DSK_QUANT_Q2_K, NW>(A.tq, act_q, tid, scratch);
        else stage_f32<NW>(A.tq, reinterpret_cast<float*>(act_q), tid, scratch);
        __syncthreads();  // scratch is reused by the second staging
      }
      if (KQ) stage_q8<QT == DSK_QUANT_Q2_K, NW>(A.tkv, act_kv, tid, scratch);
      else stage_f32<NW>(A.tkv, reinterpret_cast<float*>(act_kv), tid, scratch);
      __syncthreads();
    }
    tap_staged();
    if (tl && tid == 0) tl[1] = wall_clock64();
    if (A.has_q) head_rows(A.tq, act_q, A.lq_log2, a.head_dim, q_s);
    else
      for (int i = tid; i < a.head_dim; i += 1024) q_s[i] = a.q[(size_t)h * a.head_dim + i];
    head_rows(A.tkv, act_kv, A.lkv_log2, a.nope + a.v_dim, kvb_s);
  }
  __syncthreads();
  if (tl && tid == 0) tl[2] = wall_clock64();
  ad::rope_kv_from_lds<1024>(a, sp, h, tid, q_s, kvb_s, split == 0);
  __syncthreads();  // the rotated q (LDS) and this position's k / v (global, same CU) are read by other threads below
  if (tl && tid == 0) tl[3] = wall_clock64();
  if (S <= 1) {
    const float o = ad::attn_mha_body<1024>(a, q_s, 0, sp->kv_len, h, tid, att, scratch, part);
    if (tl && tid == 0) tl[4] = wall_clock64();
    ad::attn_out_q8(a, h, tid, o, &last_flag);
    if (tl && tid == 0) tl[5] = wall_clock64();
    return;
  }
  // ---- split context: this workgroup's share of the positions (every split wrote the same k / v row above)
  const int kv_len = sp->kv_len, vd = a.v_dim;
  const int t_lo = (int)((long long)kv_len * split / S), t_hi = (int)((long long)kv_len * (split + 1) / S);
  float o = ad::attn_mha_body<1024>(a, q_s, t_lo, t_hi, h, tid, att, scratch, part, ml_s);
  float* P = A.split_part + ((size_t)h * S + split) * (vd + 2);
  if (tid < vd) __hip_atomic_store(P + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();  // ml_s
  if (tid < 2) __hip_atomic_store(P + vd + tid, ml_s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(A.split_counter + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = old == (unsigned)(S - 1);
    if (last_flag) {
      FINISHER_ACQUIRE();
      __hip_atomic_store(A.split_counter + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    }
  }
  __syncthreads();
  if (!last_flag) return;
  // the head's last split merges: O = sum_s O_s e^{m_s - M} / sum_s l_s e^{m_s - M}, splits in order
  const float* P0 = A.split_part + (size_t)h * S * (vd + 2);
  float M = -INFINITY;
  for (int j = 0; j < S; ++j) M = fmaxf(M, __hip_atomic_load(P0 + (size_t)j * (vd + 2) + vd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  float Lsum = 0.f;
  o = 0.f;
  for (int j = 0; j < S; ++j) {
    const float* Pj = P0 + (size_t)j * (vd + 2);
    const float w = expf(__hip_atomic_load(Pj + vd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - M);
    Lsum = fmaf(__hip_atomic_load(Pj + vd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w, Lsum);
    if (tid < vd) o = fmaf(__hip_atomic_load(Pj + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w, o);
  }
  o /= Lsum;
  __syncthreads();  // last_flag is reused by the finisher
  ad::attn_out_q8(a, h, tid, o, &last_flag);
}

static int head_lpr_log2(int quant, int n) {
  const bool kq = quant == DSK_QUANT_Q2_K || quant == DSK_QUANT_Q3_K;
  const int epi = kq ? 64 : (quant == DSK_QUANT_F32 ? 4 : (quant == DSK_QUANT_F16 ? 8 : 16));
  int lpr = 64;
  while (lpr > 1 && (n / epi) % lpr) lpr >>= 1;
  int l = 0;
  while ((1 << (l + 1)) <= lpr) ++l;
  return l;
}
int head_attn_plan(HeadAttnArgs& A) {
  const bool kq = A.quant == DSK_QUANT_Q2_K || A.quant == DSK_QUANT_Q3_K;
  const int epi = kq ? 64 : (A.quant == DSK_QUANT_F32 ? 4 : (A.quant == DSK_QUANT_F16 ? 8 : 16));
  if (A.a.head_dim > 256 || A.a.head_dim % 4 || A.a.v_dim > 256 || A.a.v_dim % 4 || A.a.nope + A.a.v_dim > 512)
    DSK_FAIL(DSK_ERR_UNSUPPORTED, "attn: head_dim %d / nope %d / v_head_dim %d", A.a.head_dim, A.a.nope, A.a.v_dim);
  if (A.a.rope > 128 || (A.a.rope & 1)) DSK_FAIL(DSK_ERR_UNSUPPORTED, "rope dim %d (max 128, even)", A.a.rope);
  if (A.a.q_qs && (A.a.n_heads * A.a.v_dim) % 256) DSK_FAIL(DSK_ERR_INVALID, "attn: n_heads * v_head_dim = %d is not a multiple of 256", A.a.n_heads * A.a.v_dim);
  for (const GemvTask* T : {&A.tq, &A.tkv}) {
    if (T == &A.tq && !A.has_q) continue;
    if (T->n % (kq ? 256 : epi)) DSK_FAIL(DSK_ERR_INVALID, "head projections: n=%d", T->n);
  }
  A.lq_log2 = A.has_q ? head_lpr_log2(A.quant, A.tq.n) : 0;
  A.lkv_log2 = head_lpr_log2(A.quant, A.tkv.n);
  auto lds = [&](int n) { return (int)(((kq ? (size_t)(n / 64) * ITEM_LDS : (size_t)n * 4) + 15) & ~(size_t)15); };
  A.lds_q = A.has_q ? lds(A.tq.n) : 0;
  A.lds_kv = lds(A.tkv.n);
  if (A.tiled) {
    const int nbq = A.tq.n >> 8, nbk = A.tkv.n >> 8;
    if (A.quant != DSK_QUANT_Q2_K || !A.has_q || nbq > 8 || nbk > 8 || nbq + nbk > 16 || A.a.head_dim % 16 || (A.a.nope + A.a.v_dim) % 16 ||
        A.tq.act_mode != ACT_F32_NORM || A.tkv.act_mode != ACT_F32_NORM)
      DSK_FAIL(DSK_ERR_UNSUPPORTED, "head_attn: tiled projections need Q2_K, a q latent, rows of <= 8 blocks and heads of whole 16-row strips");
    A.red_bytes = ((A.a.head_dim >> 4) * nbq + ((A.a.nope + A.a.v_dim) >> 4) * nbk) * 256;
  }
  if (A.b0 < 1) A.b0 = 1;
  if (A.b1 < 1) A.b1 = 1;
  return DSK_OK;
}
template <int QT>
static int launch_head_attn_q(hipStream_t st, const HeadAttnArgs& A, const StepParams* sp, size_t lds) {
  auto k = head_attn_kernel<QT>;
  if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(A.a.n_heads * (A.n_split > 1 ? A.n_split : 1) + (A.pf_wgs > 0 ? A.pf_wgs : 0)), dim3(1024), lds, st, A, sp);
  return DSK_OK;
}
int launch_head_attn(hipStream_t st, const HeadAttnArgs& A0, const StepParams* sp, int max_kv, int n_split) {
  HeadAttnArgs A = A0;
  A.n_split = (n_split > 1 && A0.split_part && A0.split_counter) ? (n_split > MHA_SPLIT_MAX ? MHA_SPLIT_MAX : n_split) : 1;
  size_t lds = (size_t)A.lds_q + A.lds_kv + (size_t)max_kv * 4;
  if (A.tiled) {
    A.red_off = (max_kv + 15) & ~15;
    lds = (size_t)A.lds_q + A.lds_kv + (size_t)A.red_off * 4 + A.red_bytes;
  }
  if (lds > 120 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "attn: kv_len %d does not fit LDS", max_kv);
  switch (A.quant) {
    case DSK_QUANT_F32: return launch_head_attn_q<DSK_QUANT_F32>(st, A, sp, lds);
    case DSK_QUANT_F16: return launch_head_attn_q<DSK_QUANT_F16>(st, A, sp, lds);
    case DSK_QUANT_F8E5M2: return launch_head_attn_q<DSK_QUANT_F8E5M2>(st, A, sp, lds);
    case DSK_QUANT_Q2_K: return launch_head_attn_q<DSK_QUANT_Q2_K>(st, A, sp, lds);
    case DSK_QUANT_Q3_K: return launch_head_attn_q<DSK_QUANT_Q3_K>(st, A, sp, lds);
  }
  DSK_FAIL(DSK_ERR_INVALID, "head_attn: bad quant %d", A.quant);
}

// ------------------------------------------------------------------------------------
// MLA attention, per head (src/infer.cpp:1072-1141, attn_mla :766-804): one 16-wave workgroup rotates q_rope,
// scores the shared latent cache (every head reads the same (kv_len x 576) f16 rows: L2 / Infinity-Cache hits),
// softmax, mixes the latent values, applies the head's wv_b rows (128 x 512) and leaves the Q8_K copy for wo.
// ------------------------------------------------------------------------------------
// (Round 3, measured and rejected: the second-stage projections FOLDED into this launch the way the MHA launch consumes
// wq_b || wkv_b - head h normalises / quantises q_a itself and computes its 64 + 512 rows of wq_rope_b || wc straight into
// LDS, the latent's cache write as workgroup 0 of the same launch publishing a per-step tag, the heads scoring the rows written
// during the launch last (write-through stores + sc1 loads; an agent acquire in 2048 waves cost 14 us).  Bit-identical to
// the separate launch, and 24.6 us against 11.0 + 11.5: 290 KB of rows per head at the ~26 GB/s one CU streams are 11 us on
// 128 CUs - exactly what the 256-CU launch and its boundary cost.  scratch/exp_mla_fold/ in the builder's tree)
template <int QT>
__global__ __launch_bounds__(1024) void mla_head_kernel(const MlaHeadArgs A, const StepParams* __restrict__ sp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ __attribute__((aligned(16))) float q_s[768];    // q_c | rotated q_rope
  __shared__ __attribute__((aligned(16))) float o_s[512];    // latent output of this head
  __shared__ __attribute__((aligned(16))) float out_s[256];  // wv_b rows of this head
  __shared__ __attribute__((aligned(16))) float part[4096];
  __shared__ int last_flag;
  constexpr bool KQ = QT == DSK_QUANT_Q2_K || QT == DSK_QUANT_Q3_K;
  constexpr int NT = 1024, NW = 16;
  typedef ad::f16x4 h4;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = lane >> 4, sl = lane & 15;
  const int h = blockIdx.x;
  if (h >= A.a.n_heads) {  // prefetch workgroups (MlaHeadArgs::pf_wgs)
    tail_prefetch(A.pf_p, A.pf_n, tid, 1024);
    return;
  }
  const AttnMlaArgs& a = A.a;
  const int lora = a.lora, rope = a.rope, kv_len = sp->kv_len;
  uint8_t* act = smem;
  float* att = reinterpret_cast<float*>(smem + A.lds_act);
  unsigned long long* tl = A.timeline && h < DSK_TL_WGS ? A.timeline + (size_t)h * 8 : nullptr;
  if (tl && tid == 0) tl[0] = wall_clock64();
  // (This head's wv_b rows - 21 KB, one row group - requested here, ahead of everything, and multiplied at the end: no
  // gain, 11.1 -> 12.3 us in the bench.  The tail after the attention is the Q8_K of the latent output, two barriers and
  // one column step of arithmetic, 2.0 us with the weights already in registers; the 3.9 us in front of it are the two
  // dependent reads of the cache rows, `python tools/timeline.py --attn mla`.)
  // ---- the cache rows of the first scoring step (8 positions per wave, 128 per workgroup: the whole context of a short
  // one) are requested NOW: they were written by earlier launches and depend on nothing here, so their HBM latency runs
  // under the staging of q instead of after it ----
  const int nj = lora >> 6;  // f16x4 loads per lane for the latent part (8 for lora 512)
  h4 kc[2][8], kr[2];
  auto request_rows = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = t0 + u * 4 + grp;
      if (t < kv_len) {
        const uint16_t* c = a.nope_cache + (size_t)t * lora;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nj) kc[u][j] = *reinterpret_cast<const h4*>(c + 64 * j + sl * 4);
        if (sl * 4 < rope) kr[u] = *reinterpret_cast<const h4*>(a.rope_cache + (size_t)t * rope + sl * 4);
      }
    }
  };
  const bool merged = A.flash_thresh > 0 && kv_len >= A.flash_thresh;  // uniform: mla_flash_kernel ran before us
  if (!merged) request_rows(wave * 8);
  // ---- q: latent part as is, rope part rotated (src/infer.cpp:1075-1084) ----
  for (int i = tid; i < lora; i += NT) q_s[i] = a.q_c[(size_t)h * lora + i];
  if (tid < rope / 2) {
    const float* qr = a.q_rope + (size_t)h * rope;
    const float v0 = qr[2 * tid], v1 = qr[2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    float re, im;
    ad::rope_rot(v0, v1, c, s, re, im);
    if (a.is_v3) {
      q_s[lora + 2 * tid] = re;
      q_s[lora + 2 * tid + 1] = im;
    } else {
      q_s[lora + tid] = re;
      q_s[lora + tid + rope / 2] = im;
    }
  }
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();
  if (merged) {
    // out = sum_c e^(m_c - M) O_c / sum_c e^(m_c - M) l_c over the chunk partials of this head
    const int fcl = A.fl_chunk_len > 0 ? A.fl_chunk_len : MLA_FL_CHUNK(kv_len, A.fl_n_chunks);
    const int nc = min(A.fl_n_chunks, (kv_len + fcl - 1) / fcl);
    const int H = a.n_heads;
    ad::mla_merge_partials(A.fl_part_ml, A.fl_part_o, H, h, lora, nc, tid, part, o_s);
  } else {
  // ---- scores: 16 lanes per position, 2 positions per group and step ----
  float qv[8][4], qr4[4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) qv[j][i] = j < nj ? q_s[64 * j + sl * 4 + i] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) qr4[i] = sl * 4 + i < rope ? q_s[lora + sl * 4 + i] : 0.f;
  const float inv = sqrtf((float)a.head_dim);
  for (int t0 = wave * 8; t0 < kv_len; t0 += NW * 8) {
    if (t0 != wave * 8) request_rows(t0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = t0 + u * 4 + grp;
      float p = 0.f;
      if (t < kv_len) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nj) {
            p = fmaf(qv[j][0], (float)kc[u][j].x, p);
            p = fmaf(qv[j][1], (float)kc[u][j].y, p);
            p = fmaf(qv[j][2], (float)kc[u][j].z, p);
            p = fmaf(qv[j][3], (float)kc[u][j].w, p);
          }
        if (sl * 4 < rope) {
          p = fmaf(qr4[0], (float)kr[u].x, p);
          p = fmaf(qr4[1], (float)kr[u].y, p);
          p = fmaf(qr4[2], (float)kr[u].z, p);
          p = fmaf(qr4[3], (float)kr[u].w, p);
        }
      }
      p = ad::row16_sum(p);
      if (sl == 0 && t < kv_len) att[t] = p / inv;
    }
  }
  __syncthreads();
  // ---- softmax (src/infer.cpp:472-487) ----
  // (the latent rows of the first value step requested before it: no gain - every barrier of the softmax is a fence that
  // waits for them)
  if (kv_len <= 64) {
    // short contexts: every position lives in wave 0's lanes, so the block-wide max / sum of the general path below ARE wave 0's
    // DPP trees (the other waves contribute -inf and exact zeros): same bits, one barrier instead of five
    if (wave == 0) {
      const float a0 = lane < kv_len ? att[lane] : -INFINITY;
      const float mx0 = ad::wave_max_dpp(a0);
      const float e0 = lane < kv_len ? expf(a0 - mx0) : 0.f;
      const float sum0 = ad::wave_sum_dpp(e0);
      if (lane < kv_len) att[lane] = e0 / sum0;
    }
    __syncthreads();
  } else {
  float mx = -INFINITY;
  for (int t = tid; t < kv_len; t += NT) mx = fmaxf(mx, att[t]);
  mx = ad::block_max(mx, scratch, tid, NT);
  float sum = 0.f;
  for (int t = tid; t < kv_len; t += NT) {
    const float e = expf(att[t] - mx);
    att[t] = e;
    sum += e;
  }
  sum = ad::block_sum(sum, scratch, tid, NT);
  for (int t = tid; t < kv_len; t += NT) att[t] = att[t] / sum;
  __syncthreads();
  }
  // ---- latent values: lora / 4 threads per position, NT / (lora / 4) positions in flight, 4 rows per thread ----
  {
    const int tpp = lora >> 2, TG = NT / tpp;
    const int g = tid / tpp, i4 = tid - g * tpp;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < TG) {
      for (int t0 = g; t0 < kv_len; t0 += 4 * TG) {
        h4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t0 + k * TG < kv_len) v[k] = *reinterpret_cast<const h4*>(a.nope_cache + (size_t)(t0 + k * TG) * lora + i4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t0 + k * TG < kv_len) {
            const float w = att[t0 + k * TG];
            acc[0] = fmaf(w, (float)v[k].x, acc[0]);
            acc[1] = fmaf(w, (float)v[k].y, acc[1]);
            acc[2] = fmaf(w, (float)v[k].z, acc[2]);
            acc[3] = fmaf(w, (float)v[k].w, acc[3]);
          }
      }
      *reinterpret_cast<f32x4*>(part + (size_t)g * lora + i4 * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
    __syncthreads();
    if (tid < lora) {
      float o = 0.f;
      for (int gg = 0; gg < TG; ++gg) o += part[gg * lora + tid];
      o_s[tid] = o;
    }
    __syncthreads();
  }
  }  // !merged
  if (tl && tid == 0) tl[2] = wall_clock64();
  // ---- this head's wv_b rows on the latent output (src/infer.cpp:1134-1137) ----
  const int vd = A.fin.v_dim;
  const bool wv_tiled = QT == DSK_QUANT_Q2_K && A.tiled;
  if constexpr (KQ) {
    if (wave < (lora >> 8)) {  // Q8_K of o_s: one 256-block per wave
      float v[4] = {o_s[wave * 256 + lane * 4], o_s[wave * 256 + lane * 4 + 1], o_s[wave * 256 + lane * 4 + 2], o_s[wave * 256 + lane * 4 + 3]};
      if (wv_tiled) q8k_block_lds<LAY_TILE>(v, lane, act + (size_t)wave * TREC);
      else q8k_block_lds<QT == DSK_QUANT_Q2_K>(v, lane, act + (size_t)wave * 4 * ITEM_LDS);
    }
  } else {
    for (int i = tid; i < lora; i += NT) reinterpret_cast<float*>(act)[i] = o_s[i];
  }
  __syncthreads();
#ifndef DSK_NO_TAPS
  if (A.tap_o) {  // parity tap: this head's latent output and the Q8_K vector staged for wv_b
    for (int i = tid; i < lora; i += NT) A.tap_o[(size_t)h * lora + i] = o_s[i];
    if constexpr (KQ) {
      if (wv_tiled) dump_staged_q8<LAY_TILE>(act, lora, A.tap_qs + (size_t)h * lora, A.tap_d + (size_t)h * (lora >> 8), tid, NT);
      else dump_staged_q8<QT == DSK_QUANT_Q2_K>(act, lora, A.tap_qs + (size_t)h * lora, A.tap_d + (size_t)h * (lora >> 8), tid, NT);
    }
  }
#endif
  if (wv_tiled) {
    // Tiled wv_b (tile_device.h): head h's v_dim / 16 strips x lora / 256 blocks as one list of steps (rows of <= 8 blocks: one item per
    // block), one or more per wave, partials in LDS (`part` is free again), one wave per strip adds them in the fixed order
    if constexpr (QT == DSK_QUANT_Q2_K) {
      const TLane TL = tlane_init(lane);
      const int nb = lora >> 8, strips = vd >> 4, J = strips * nb;
      const rsrc_t Wv = make_rsrc(A.twv.qs);
      for (int j = wave; j < J; j += NW) {
        const int st = j / nb, b = j - st * nb;
        TStep S;
        tstep_load(S, Wv, TL, ((h * strips + st) * nb + b) * TILE_B);
        float ad_ = 0.f, am_ = 0.f;
        tstep_mac(S, act + (size_t)b * TREC, TL, ad_, am_);
        part[(size_t)j * 64 + lane] = titem_value(ad_, am_, TL);
      }
      __syncthreads();
      for (int st = wave; st < strips; st += NW) {
        const float v = tile_strip_value(part + (size_t)st * nb * 64, nb, lane);
        if (lane < 16) out_s[st * 16 + lane] = v;
      }
    }
  } else {
    const int lpr_log2 = A.lpr_log2, RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    const WPtr P = resolve(A.twv);
    const KQRsrc B = kq_rsrc<QT, false>(P);
    const int nb = lora >> 8;
    for (int base = wave * RPW; base < vd; base += NW * RPW) {
      const int lr = base + rloc;
      const bool valid = lr < vd;
      int row[1] = {h * vd + (valid ? lr : vd - 1)};
      float acc[1], acc2[1];
      if constexpr (KQ) {
        int rowblk[1] = {row[0] * nb + (sub >> 2)};
        rows_dot_kq<QT, 1, 4, false>(B, nb * 4, sub, lpr_log2, sub & 3, rowblk, act + sub * ITEM_LDS, acc, acc2);
      } else {
        // reference indexing of the F8 block scales for this stack: expert_index = head (src/infer.cpp:437-438)
        WPtr Ph = P;
        if (Ph.scale) Ph.scale += (size_t)h * ((vd + A.b0 - 1) / A.b0) * ((lora + A.b1 - 1) / A.b1);
        int lrow[1] = {valid ? lr : vd - 1};
        Ph.qs += (size_t)h * vd * lora * FTraits<QT>::ESZ;
        rows_dot_f<QT, 1, 4, false>(Ph, lora, A.b0, A.b1, lpr_log2, lane, lrow, act, acc, acc2);
      }
      if (sub == 0 && valid) out_s[lr] = acc[0];
    }
  }
  __syncthreads();
  if (tl && tid == 0) tl[3] = wall_clock64();
  ad::attn_out_q8(A.fin, h, tid, tid < vd ? out_s[tid] : 0.f, &last_flag);
  if (tl && tid == 0) tl[4] = wall_clock64();
}

int mla_head_plan(MlaHeadArgs& A) {
  const bool kq = A.quant == DSK_QUANT_Q2_K || A.quant == DSK_QUANT_Q3_K;
  if (A.a.lora > 512 || A.a.lora % 64 || A.a.rope > 64 || A.a.rope % 4 || A.fin.v_dim > 256)
    DSK_FAIL(DSK_ERR_UNSUPPORTED, "mla: kv_lora_rank %d / rope %d / v_head_dim %d", A.a.lora, A.a.rope, A.fin.v_dim);
  if (kq && A.a.lora % 256) DSK_FAIL(DSK_ERR_INVALID, "mla: kv_lora_rank %d is not a multiple of 256", A.a.lora);
  if (A.fin.q_qs && (A.fin.n_heads * A.fin.v_dim) % 256) DSK_FAIL(DSK_ERR_INVALID, "mla: n_heads * v_head_dim = %d is not a multiple of 256", A.fin.n_heads * A.fin.v_dim);
  A.lpr_log2 = head_lpr_log2(A.quant, A.a.lora);
  A.lds_act = (int)(((kq ? (size_t)(A.a.lora / 64) * ITEM_LDS : (size_t)A.a.lora * 4) + 15) & ~(size_t)15);
  if (A.b0 < 1) A.b0 = 1;
  if (A.b1 < 1) A.b1 = 1;
  return DSK_OK;
}
template <int QT>
static int launch_mla_head_q(hipStream_t st, const MlaHeadArgs& A, const StepParams* sp, size_t lds) {
  auto k = mla_head_kernel<QT>;
  if (lds > 32 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(A.a.n_heads + (A.pf_wgs > 0 ? A.pf_wgs : 0)), dim3(1024), lds, st, A, sp);
  return DSK_OK;
}
int launch_mla_head(hipStream_t st, const MlaHeadArgs& A, const StepParams* sp, int max_kv) {
  const size_t lds = (size_t)A.lds_act + (size_t)max_kv * 4;
  if (lds > 100 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "mla: kv_len %d does not fit LDS", max_kv);
  switch (A.quant) {
    case DSK_QUANT_F32: return launch_mla_head_q<DSK_QUANT_F32>(st, A, sp, lds);
    case DSK_QUANT_F16: return launch_mla_head_q<DSK_QUANT_F16>(st, A, sp, lds);
    case DSK_QUANT_F8E5M2: return launch_mla_head_q<DSK_QUANT_F8E5M2>(st, A, sp, lds);
    case DSK_QUANT_Q2_K: return launch_mla_head_q<DSK_QUANT_Q2_K>(st, A, sp, lds);
    case DSK_QUANT_Q3_K: return launch_mla_head_q<DSK_QUANT_Q3_K>(st, A, sp, lds);
  }
  DSK_FAIL(DSK_ERR_INVALID, "mla_head: bad quant %d", A.quant);
}

// ------------------------------------------------------------------------------------
// host side: choose the launch geometry and fill the descriptor
// ------------------------------------------------------------------------------------
static int ilog2(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}

thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

template <int QT, int R, int U, int NW>
static void launch_one(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h) {
  dim3 grid(h.grid), block(NW * 64);
  // activation hint (see ActSrc): only when every workgroup stages the same vector
  const void *a0 = nullptr, *a1 = nullptr, *a2 = nullptr;
  int hn = 0, hm = 0, gw = 0, gs = 0;
  float he = 0.f;
  static const bool no_hint = dsk_ab_env("DSK_NO_HINT") != nullptr;  // A/B knob of tools/kbench.py
  if (h.bd_heads == 0 && h.n_groups == 1 && !no_hint) {
    const GemvTask& T = h.t[h.grp_t0[0]];
    hn = T.n; hm = T.act_mode; he = T.eps;
    if (T.act_mode == ACT_Q8) { a0 = T.a_qs; a1 = T.a_d; a2 = T.a_bsums; }
    else { a0 = T.a_f32; a1 = T.norm_w; }
  } else if (h.bd_heads == 0 && h.n_groups > 1 && !no_hint) {
    // e.g. the experts' W2 launch: one group per slot, f32 hidden vectors k * stride apart
    const GemvTask& T0 = h.t[h.grp_t0[0]];
    const int w0 = h.grp_wg_end[0];
    const ptrdiff_t st = h.t[h.grp_t0[1]].a_f32 - T0.a_f32;
    bool ok = T0.act_mode == ACT_F32 && st > 0 && st < (1 << 30);
    for (int g = 0; ok && g < h.n_groups; ++g) {
      const GemvTask& T = h.t[h.grp_t0[g]];
      ok = T.act_mode == ACT_F32 && T.n == T0.n && T.a_f32 == T0.a_f32 + (ptrdiff_t)g * st && h.grp_wg_end[g] == (g + 1) * w0;
    }
    if (ok) { a0 = T0.a_f32; hn = T0.n; hm = ACT_F32; gw = w0; gs = (int)st; }
  }
  if (h.glu) {
    auto k = gemv_kernel<QT, R, U, true, NW>;
    if (h.lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_bytes);
    if (g_prof_start && g_prof_stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)h.lds_bytes, st, g_prof_start, g_prof_stop, 0u, dev, a0, a1, a2, hn, hm, he, gw, gs);
    else hipLaunchKernelGGL(k, grid, block, h.lds_bytes, st, dev, a0, a1, a2, hn, hm, he, gw, gs);
  } else if (QT == DSK_QUANT_Q2_K && R == 1 && (U == 2 || U == 1) && NW == 16 && (h.ahead & 1) && hn > 0 && hm == ACT_F32_NORM && gemv_ahead_ok(h)) {
    auto k = U == 2 ? gemv_ahead_kernel<112, 6> : gemv_ahead_kernel<32, 5>;
    if (h.lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_bytes);
    if (g_prof_start && g_prof_stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)h.lds_bytes, st, g_prof_start, g_prof_stop, 0u, dev, (const float*)a0, (const float*)a1, he);
    else hipLaunchKernelGGL(k, grid, block, h.lds_bytes, st, dev, (const float*)a0, (const float*)a1, he);
  } else if (QT == DSK_QUANT_Q2_K && R == 1 && (U == 4 || U == 1) && NW == 16 && (h.ahead & 2) && hn > 0 && hm == ACT_Q8 && gemv_ahead_q8_ok(h)) {
    auto k = U == 4 ? gemv_ahead_q8_kernel<256, 6> : gemv_ahead_q8_kernel<32, 5>;
    if (h.lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_bytes);
    if (g_prof_start && g_prof_stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)h.lds_bytes, st, g_prof_start, g_prof_stop, 0u, dev, (const int8_t*)a0, (const float*)a1, (const int16_t*)a2);
    else hipLaunchKernelGGL(k, grid, block, h.lds_bytes, st, dev, (const int8_t*)a0, (const float*)a1, (const int16_t*)a2);
  } else {
    auto k = gemv_kernel<QT, R, U, false, NW>;
    if (h.lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_bytes);
    if (g_prof_start && g_prof_stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)h.lds_bytes, st, g_prof_start, g_prof_stop, 0u, dev, a0, a1, a2, hn, hm, he, gw, gs);
    else hipLaunchKernelGGL(k, grid, block, h.lds_bytes, st, dev, a0, a1, a2, hn, hm, he, gw, gs);
  }
}
template <int QT, int NW>
static int launch_q(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h) {
  // (R, U) variants: R rows x U column steps = the 16-byte loads a lane keeps in flight
  switch (h.R * 16 + h.U) {
    case 1 * 16 + 8: launch_one<QT, 1, 8, NW>(st, dev, h); break;
    case 1 * 16 + 4: launch_one<QT, 1, 4, NW>(st, dev, h); break;
    case 1 * 16 + 2: launch_one<QT, 1, 2, NW>(st, dev, h); break;
    case 1 * 16 + 1: launch_one<QT, 1, 1, NW>(st, dev, h); break;
    case 2 * 16 + 4: launch_one<QT, 2, 4, NW>(st, dev, h); break;
    case 2 * 16 + 2: launch_one<QT, 2, 2, NW>(st, dev, h); break;
    case 2 * 16 + 1: launch_one<QT, 2, 1, NW>(st, dev, h); break;
    default: DSK_FAIL(DSK_ERR_INVALID, "gemv: no kernel variant R=%d U=%d", h.R, h.U);
  }
  return DSK_OK;
}
template <int QT>
static int launch_nw(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h) {
  return h.NW == 16 ? launch_q<QT, 16>(st, dev, h) : launch_q<QT, 4>(st, dev, h);
}

// router + shared expert in one launch; h = the planned one-task GLU descriptor of the shared expert's w1/w3
bool router_shared_supported(const RouterArgs& a, const GemvLaunch& h) {
  if (h.tiled)
    return a.ksplit >= 8 && a.norm_w && h.glu && h.n_tasks == 1 && h.n_groups == 1 && h.bd_heads == 0 && h.NW == 16 &&
           h.t[0].act_mode == ACT_F32_NORM && h.t[0].n == a.dim && (a.dim >> 8) <= 32 && !h.comb_x;
  const bool q2 = h.quant == DSK_QUANT_Q2_K, q3 = h.quant == DSK_QUANT_Q3_K;
  return (q2 || q3) && a.ksplit >= 8 && a.norm_w && h.glu && h.n_tasks == 1 && h.n_groups == 1 && h.bd_heads == 0 && h.NW == 16 && h.R == 1 &&
         (h.U == 2 || (q2 && h.U == 4)) && h.t[0].act_mode == ACT_F32_NORM && h.t[0].n == a.dim && (a.dim >> 8) <= 32 && !h.comb_x;
}
int launch_router_shared(hipStream_t st, const RouterArgs& a, const GemvLaunch* dev, const GemvLaunch& h) {
  if (!router_shared_supported(a, h)) DSK_FAIL(DSK_ERR_INVALID, "router_shared: unsupported plan");
  const int n_router = (a.n_routed + 1) / 2;
  dim3 grid(n_router + h.grid), block(1024);
  const size_t lds = h.lds_bytes;
  if (h.tiled) {
    auto k = router_shared_tile_kernel;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, st, a, n_router, dev);
    return DSK_OK;
  }
#define RS_LAUNCH(QT, U)                                                                                             \
  do {                                                                                                               \
    auto k = router_shared_kernel<QT, U>;                                                                            \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
    hipLaunchKernelGGL(k, grid, block, lds, st, a, n_router, dev);                                                   \
  } while (0)
  if (h.quant == DSK_QUANT_Q2_K) {
    if (h.U == 4) RS_LAUNCH(DSK_QUANT_Q2_K, 4);
    else RS_LAUNCH(DSK_QUANT_Q2_K, 2);
  } else {
    RS_LAUNCH(DSK_QUANT_Q3_K, 2);  // (4 column steps in flight spill at 16 waves, like the plain GLU variant)
  }
#undef RS_LAUNCH
  return DSK_OK;
}

bool gemv_kvwrite_supported(const GemvLaunch& h) {
  if (h.tiled) return !h.glu && h.n_groups == 1 && h.bd_heads == 0 && h.NW == 16 && !h.comb_x && !h.timeline;
  const bool kq = h.quant == DSK_QUANT_Q2_K || h.quant == DSK_QUANT_Q3_K;
  return kq && !h.glu && h.n_groups == 1 && h.bd_heads == 0 && h.NW == 16 && h.R == 1 && (h.U == 4 || h.U == 2) && !h.comb_x && !h.timeline;
}
// which of the round-6 "weights ahead of the staging" kernels a planned launch runs: 0 none, 1 gemv_ahead_kernel (first-stage
// projections), 2 gemv_ahead_q8_kernel (wo), 3 gemv_kvwrite_ahead_kernel (the MLA second stage; `kvwrite`: launched with the rider)
int gemv_ahead_kind(const GemvLaunch& h, bool kvwrite) {
  if (kvwrite) return gemv_kvwrite_supported(h) && gemv_kvwrite_ahead_ok(h) ? 3 : 0;
  if (h.NW != 16 || h.R != 1 || h.bd_heads != 0 || h.n_groups != 1) return 0;
  const GemvTask& T = h.t[h.grp_t0[0]];
  if ((h.ahead & 1) && T.act_mode == ACT_F32_NORM && gemv_ahead_ok(h)) return 1;
  if ((h.ahead & 2) && T.act_mode == ACT_Q8 && gemv_ahead_q8_ok(h)) return 2;
  return 0;
}
int launch_gemv_kvwrite(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h, const MlaKvArgs& kv, const StepParams* sp) {
  if (!gemv_kvwrite_supported(h)) DSK_FAIL(DSK_ERR_INVALID, "gemv_kvwrite: unsupported plan");
  if (kv.rope > 128 || (kv.rope & 1)) DSK_FAIL(DSK_ERR_UNSUPPORTED, "rope dim %d (max 128, even)", kv.rope);
  const GemvTask& T = h.t[h.grp_t0[0]];  // one activation group: the hint of launch_one
  const void *a0 = T.a_f32, *a1 = T.norm_w, *a2 = nullptr;
  if (T.act_mode == ACT_Q8) { a0 = T.a_qs; a1 = T.a_d; a2 = T.a_bsums; }
  dim3 grid(h.grid + 1), block(1024);
  const size_t lds = h.lds_bytes;
  if (h.tiled) {
    auto k = gemv_kvwrite_tile_kernel;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, st, dev, a0, a1, a2, T.n, T.act_mode, T.eps, kv, sp);
    return DSK_OK;
  }
  if (gemv_kvwrite_ahead_ok(h)) {
    auto k = gemv_kvwrite_ahead_kernel<24, 3>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, st, dev, (const float*)a0, (const float*)a1, T.eps, kv, sp);
    return DSK_OK;
  }
#define KV_LAUNCH(QT, U)                                                                                             \
  do {                                                                                                               \
    auto k = gemv_kvwrite_kernel<QT, U>;                                                                             \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
    hipLaunchKernelGGL(k, grid, block, lds, st, dev, a0, a1, a2, T.n, T.act_mode, T.eps, kv, sp);                    \
  } while (0)
  if (h.quant == DSK_QUANT_Q2_K) {
    if (h.U == 4) KV_LAUNCH(DSK_QUANT_Q2_K, 4);
    else KV_LAUNCH(DSK_QUANT_Q2_K, 2);
  } else {
    if (h.U == 4) KV_LAUNCH(DSK_QUANT_Q3_K, 4);
    else KV_LAUNCH(DSK_QUANT_Q3_K, 2);
  }
#undef KV_LAUNCH
  return DSK_OK;
}

int gemv_launch(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h) {
  if (h.tiled) return gemv_launch_tile(st, dev, h);
  switch (h.quant) {
    case DSK_QUANT_F32: return launch_nw<DSK_QUANT_F32>(st, dev, h);
    case DSK_QUANT_F16: return launch_nw<DSK_QUANT_F16>(st, dev, h);
    case DSK_QUANT_F8E5M2: return launch_nw<DSK_QUANT_F8E5M2>(st, dev, h);
    case DSK_QUANT_Q2_K: return launch_nw<DSK_QUANT_Q2_K>(st, dev, h);
    case DSK_QUANT_Q3_K: return launch_nw<DSK_QUANT_Q3_K>(st, dev, h);
  }
  DSK_FAIL(DSK_ERR_INVALID, "gemv: bad quant %d", h.quant);
}

// Decide lanes-per-row, rows-per-wave, grid and workgroup ranges.  `target_wgs` ~ a few per CU.
int gemv_plan(GemvLaunch& h, int target_wgs) {
  if (h.tiled) return gemv_plan_tile(h, target_wgs);
  const bool cg = h.comb_x != nullptr || h.comb_geometry != 0;  // fused-combine geometry
  if (h.n_tasks < 1 || h.n_tasks > GEMV_MAX_TASKS) DSK_FAIL(DSK_ERR_INVALID, "gemv: %d tasks", h.n_tasks);
  const bool kq = h.quant == DSK_QUANT_Q2_K || h.quant == DSK_QUANT_Q3_K;
  const int epi = kq ? 64 : (h.quant == DSK_QUANT_F32 ? 4 : (h.quant == DSK_QUANT_F16 ? 8 : 16));
  size_t lds_max = 0, lds_sum = 0;
  long total_rows = 0;
  double total_work = 0;
  int min_items = 1 << 30;
  for (int i = 0; i < h.n_tasks; ++i) {
    GemvTask& T = h.t[i];
    if (T.rows <= 0 || T.n <= 0) DSK_FAIL(DSK_ERR_INVALID, "gemv: empty shape %d x %d", T.rows, T.n);
    if (kq && T.n % QK_K) DSK_FAIL(DSK_ERR_INVALID, "k-quant gemv: n=%d is not a multiple of 256 (quantizer.cpp:8)", T.n);
    if (!kq && T.n % epi) DSK_FAIL(DSK_ERR_INVALID, "gemv: n=%d is not a multiple of %d (src/infer.cpp:169,246)", T.n, epi);
    const size_t lds = kq ? (size_t)(T.n / 64) * ITEM_LDS : (size_t)T.n * 4;
    lds_max = lds > lds_max ? lds : lds_max;
    lds_sum += (lds + 15) & ~(size_t)15;
    total_rows += T.rows;
    total_work += (double)T.rows * T.n * (h.glu ? 2 : 1);
    min_items = T.n / epi < min_items ? T.n / epi : min_items;
    if (cg && T.rows != h.t[0].rows) DSK_FAIL(DSK_ERR_INVALID, "gemv combine: tasks must share the row count");
  }
  h.lds_bytes = lds_max;
  (void)lds_sum;
  if (h.lds_bytes > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "gemv: activation vector(s) need %zu B of LDS", h.lds_bytes);
  if (h.b0 < 1) h.b0 = 1;
  if (h.b1 < 1) h.b1 = 1;
  // lanes per row: the largest power of two <= 64 dividing every task's item count
  // (K-quants need an exact fit: a lane's quarter must not change between column steps.  The float paths mask a
  // ragged last step, so they take the widest lane group that wastes <= 1/8 of its slots: V2-Lite's 10944-wide
  // F8 rows are 684 items = 4 x 171 -- 4 lanes per row would leave 32 workgroups for the whole matrix.)
  // Small plain launches (the first-stage projections: 2112 rows of 7168 = 112 items) are bound by the serial work of a
  // wave, not by bytes: 64 lanes per row with a ragged second step (1/8 of the lane slots idle) give every wave of a
  // workgroup one row and two column steps instead of four waves four rows and seven steps each (tools/kbench.py sweep:
  // wq_a 8.3 -> 6.5 us, wkv_a 8.0 -> 5.9 us; in the model 7.9 -> 7.15 us).  GLU launches keep the exact fit: the shared
  // expert's rider must produce the bits of the expert-sharded arrangement, where it is the ninth task of the experts'
  // launch.  (Requesting such a launch's one row group AHEAD of its prologue was measured too: the request needs the
  // launch descriptor, a cold 1.5 us read that otherwise hides under the staging of x, which the kernel arguments
  // describe - staged at 4.8 us instead of 2.9, the launch 8.3 us instead of 7.15.)
  const bool small_plain = kq && !h.glu && !cg && h.bd_heads <= 0 && total_rows <= 4096;
  int lpr = 64;
  while (lpr > 1) {
    bool ok = true;
    for (int i = 0; i < h.n_tasks; ++i) {
      const int items = h.t[i].n / epi;
      if (kq) ok = ok && (items % lpr == 0 || (lpr >= 32 && (items + lpr - 1) / lpr * lpr * (small_plain ? 7 : 8) <= items * (small_plain ? 8 : 9)));
      else ok = ok && ((items + lpr - 1) / lpr * lpr * 8 <= items * 9);
    }
    if (ok) break;
    lpr >>= 1;
  }
  if (h.force_lpr > 0) {
    for (int i = 0; i < h.n_tasks; ++i)
      if (kq && h.force_lpr < 4) DSK_FAIL(DSK_ERR_INVALID, "gemv: lanes-per-row %d does not divide the row", h.force_lpr);
    lpr = h.force_lpr;
  }
  const long rows_eff = h.bd_heads > 0 ? (long)h.t[0].rows * h.bd_heads : total_rows;
  // (fewer lanes per row so that a 16-wave workgroup's share fits ONE row group -- wo: 32 lanes, one round trip of 8
  // loads instead of two of 4 -- measured slower: 11.9 -> 12.9 us; two rows per lane likewise: 12.0 -> 13.1)
  h.lpr_log2 = ilog2(lpr);
  // (R, U) variant and grid, from the sweeps of tools/kbench.py on MI355X (DeepSeek-V3 shapes): the
  // geometry moves a launch by < 10 % -- U = 4 column steps in flight is right for plain and GLU
  // launches alike, big launches (> 64 MB) want ~8 workgroups per CU.  The fused MoE combine pays one
  // arrival (barrier + atomic round trip) per row group, so it wants FEW, tall groups: 8 lanes per row,
  // 2 row sets => 64 rows per group (experts_w2: 16.7 us vs 46 us at 32 lanes per row).
  // Float weights (F8E5M2 / F16 / F32: 16 bytes = 16 / 8 / 4 weights per item, so rows are 4-16x longer in items than
  // a K-quant row of the same width) want 16 lanes per row and ONE row set: V2-Lite F8 experts' W2 (7 x 2048 x 1408), sweep of
  // tools/kbench-style runs: 14.7 us with the K-quant geometry, 11.9-12.2 us at 16 lanes x 1 row set.
  const bool cg_float = cg && !kq;
  if (cg && h.force_lpr <= 0) {
    while (lpr > (cg_float ? 16 : 8)) lpr >>= 1;
    h.lpr_log2 = ilog2(lpr);
  }
  const int its = (min_items + lpr - 1) / lpr;
  h.R = cg && !cg_float ? 2 : 1;
  // one HBM round trip per row group when the row fits: U = column steps of a row, rounded up to a power
  // of two (<= 8; the GLU pair holds two matrices per step: <= 4); longer rows take chunks of 4
  h.U = 1;
  while (h.U < its && h.U < (h.glu ? 4 : 8)) h.U *= 2;
  if (its > h.U) h.U = 4;
  if (h.R > 1 && h.U > 4) h.U = 4;
  (void)rows_eff;
  if (h.force_R > 0) h.R = h.force_R;
  if (h.force_U > 0) h.U = h.force_U;
  const int RPW = 64 / lpr;
  // Workgroup size.  The per-workgroup prologue (staging + quantising the activation vector) is serial
  // work in front of every launch: ~4 us for rmsnorm + Q8_K of 7168 values on 4 waves.  Launches with
  // enough rows to give each of the 256 CUs a full 16-wave row group run 16-wave workgroups, one per CU:
  // a quarter of the blocks per wave and a quarter of the redundant prologues.
  h.NW = 4;
  if (!cg && h.bd_heads <= 0 && total_rows >= 192L * 16 * RPW * h.R) h.NW = 16;
  // small launches (the first-stage projections: 2112 rows of 7168): 16-wave workgroups quarter-filled with rows still
  // quantise the activation vector four times faster than 4-wave ones (2 instead of 7 blocks per wave)
  static const int small_nw16 = dsk_ab_env("DSK_SMALL_NW16") ? atoi(dsk_ab_env("DSK_SMALL_NW16")) : 1;  // measured: qkv_a 9.7 -> 7.9 us
  int fill_div = 2;
  if (small_nw16 && h.NW == 4 && !cg && h.bd_heads <= 0 && kq && h.force_NW <= 0 && total_rows >= 32L * 16 * RPW) { h.NW = 16; fill_div = 4; }
  if (h.force_NW == 4 || h.force_NW == 16) h.NW = h.force_NW;
  if (h.fill_div > 0) fill_div = h.fill_div;
  // Q3_K items hold three planes per step: under the 128-VGPR budget of a 16-wave workgroup the wide
  // variants spill inside the column loop (hipcc: 172..664 B of scratch), so they take fewer steps in flight
  if (h.quant == DSK_QUANT_Q3_K && h.NW == 16 && h.force_U <= 0) {
    const int cap = h.glu ? 2 : 4;
    if (h.U > cap) h.U = cap;
  }
  // (Round 3, measured and rejected - an L2 warm-up rider: 128 extra workgroups of the per-head attention launch (the CUs
  // it leaves idle) READ the first 32 / 64 / 112 KB of every wo workgroup's rows with cacheable loads, XCD-matched
  // (workgroup b runs on XCD b % 8): wo 12.3 -> 11.4 / 11.2 / 10.8 us, attention 16.8 -> 17.9 / 18.0 / 18.7, first-stage
  // projections 7.1 -> 7.9 / 8.0 / 8.3 (their x and norm weights leave the L2): 5.19 -> 5.34 / 5.35 / 5.45 ms.  A GEMV whose
  // weights sit in its XCD's L2 is ~10 % faster, no more: these launches are bound on the CU side, not by HBM latency.)
  // (Round 3, measured and rejected: for short rows whose per-workgroup share spans 2.25 row groups - the MLA path's
  // second-stage projections, 73 728 rows of 1536 - three row sets per lane with the whole share requested at once
  // (R = 3, U = 3, 114 VGPRs, no scratch) instead of three dependent round trips: the launch went 12.6 -> 13.5 us and the
  // MLA token 5.58 -> 5.75 ms.  Like every deeper burst before it: a CU's read window is full either way, and the
  // arithmetic of the first rows then waits for the whole burst.)
  const int RG = h.NW * RPW * h.R;
  h.part_unit = cg ? RG : 1;
  if (h.bd_heads > 0) {
    const int n_groups = (h.t[0].rows + RG - 1) / RG;
    int per_head = target_wgs / h.bd_heads;
    if (per_head < 1) per_head = 1;
    if (per_head > n_groups) per_head = n_groups;
    h.bd_wgs = per_head;
    h.grid = per_head * h.bd_heads;
    h.t[0].wg_begin = 0;
    h.t[0].wg_end = h.grid;
    h.t[0].vrow_begin = 0;
    h.t[0].vrow_end = h.t[0].rows;
    h.n_groups = 1; h.grp_wg_end[0] = h.grid; h.grp_t0[0] = 0; h.grp_t0[1] = 1;
    return DSK_OK;
  }
  // activation groups: consecutive tasks that read the same vector (the fused combine keeps one task per
  // group: its arrival counters are per task)
  const int W = h.NW == 16 ? 256 - (h.reserve_wgs > 0 && h.reserve_wgs < 128 ? h.reserve_wgs : 0) : target_wgs;
  int wg = 0;
  h.n_groups = 0;
  for (int i = 0; i < h.n_tasks;) {
    int j = i + 1;
    while (!cg && j < h.n_tasks && h.t[j].act_mode == h.t[i].act_mode && h.t[j].a_f32 == h.t[i].a_f32 &&
           h.t[j].a_qs == h.t[i].a_qs && h.t[j].norm_w == h.t[i].norm_w && h.t[j].n == h.t[i].n)
      ++j;
    long rows_g = 0;
    double work_g = 0;
    for (int k = i; k < j; ++k) {
      h.t[k].vrow_begin = (int)rows_g;
      rows_g += h.t[k].rows;
      h.t[k].vrow_end = (int)rows_g;
      work_g += (double)h.t[k].rows * h.t[k].n * (h.glu ? 2 : 1);
    }
    int share = (int)(W * (work_g / total_work) + 0.5);
    const long units = (rows_g + h.part_unit - 1) / h.part_unit;
    // no more workgroups than half-filled row groups (combine: than whole groups, evenly dealt)
    long cap = cg ? units : (fill_div * rows_g + RG - 1) / RG;
    if (share > cap) share = (int)cap;
    if (share < 1) share = 1;
    if (cg) share = (int)((units + (units + share - 1) / share - 1) / ((units + share - 1) / share));
    for (int k = i; k < j; ++k) { h.t[k].wg_begin = wg; h.t[k].wg_end = wg + share; }
    wg += share;
    h.grp_t0[h.n_groups] = i;
    h.grp_wg_end[h.n_groups] = wg;
    ++h.n_groups;
    i = j;
  }
  h.grp_t0[h.n_groups] = h.n_tasks;
  h.grid = wg;
  return DSK_OK;
}
