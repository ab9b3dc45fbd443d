// kernels_moe.hip -- the routed experts of one MoE block in ONE launch (K-quant models, one GPU).
//
// Replaces, for the selected experts e_0..e_{K-1} of a token (src/infer.cpp:853-878, 899-903):
//     h_k  = act(W1[e_k] xb) * (W3[e_k] xb)          matmul_expert x2 + GLU
//     o_k  = W2[e_k] h_k                              matmul_expert (h_k re-quantised to Q8_K, src/infer.cpp:325-336)
//     x   += w_k * o_k  (k order),  x += shared_out   (the shared expert's W2 runs here too)
// The two-launch form (gemv w1/w3 -> gemv w2 + combine) pays a kernel boundary and a full prologue between the halves
// although slot k's second half only needs slot k's hidden vector.  Here one resident grid of 16-wave workgroups (one
// per CU) runs both halves:
//   phase A  flat list of (slot, 64-row unit) pairs of the w1/w3 GLU; a finished unit publishes its rows of h_k
//            (write-through stores) and arrives on the slot's counter;
//   phase B  flat list of (slot, 256-row unit) pairs of W2 (shared expert = one more slot).  The unit's weight chunk is
//            REQUESTED FIRST, then the workgroup waits for its slot's counter (a local hand-off among the ~32
//            workgroups of the slot, not a grid barrier), stages h_k (sc1 loads -> Q8_K in LDS) and multiplies;
//   combine  as in the two-launch form: slot outputs go out write-through, one arrival per 256-row group, the last
//            arriver adds x += w_k o_k in k order, then the shared expert.
// Per-row arithmetic (lanes per row, column-step order, reduction trees, Q8_K staging, combine order) is the same code
// with the same parameters as the two-launch form: results are BIT-identical to it (tests/test_fused_moe_gpu.py).
// Every workgroup of the grid is resident (grid <= CUs), producers never wait, spins are bounded (err flag).
#include "dsk_internal.h"
#include "gemv_device.h"

namespace {

DEV KQRsrc expert_rsrc13(const MoeFfnArgs& a, int e) {
  WPtr p;
  p.present = true;
  p.qs = a.w1_qs + (size_t)e * a.e13_qs;
  p.sc = a.w1_sc + (size_t)e * a.e13_sc;
  p.hm = a.w1_hm ? a.w1_hm + (size_t)e * a.e13_hm : nullptr;
  p.dm = a.w1_dm + (size_t)e * a.e13_dm;
  p.qs2 = a.w3_qs + (size_t)e * a.e13_qs;
  p.sc2 = a.w3_sc + (size_t)e * a.e13_sc;
  p.hm2 = a.w3_hm ? a.w3_hm + (size_t)e * a.e13_hm : nullptr;
  p.dm2 = a.w3_dm + (size_t)e * a.e13_dm;
  p.scale = p.scale2 = nullptr;
  return p.hm ? kq_rsrc<DSK_QUANT_Q3_K, true>(p) : kq_rsrc<DSK_QUANT_Q2_K, true>(p);
}

template <int QT>
DEV KQRsrc w2_rsrc(const MoeFfnArgs& a, int slot, int e) {
  WPtr p;
  p.present = true;
  if (slot < a.K) {
    p.qs = a.w2_qs + (size_t)e * a.e2_qs;
    p.sc = a.w2_sc + (size_t)e * a.e2_sc;
    p.hm = a.w2_hm ? a.w2_hm + (size_t)e * a.e2_hm : nullptr;
    p.dm = a.w2_dm + (size_t)e * a.e2_dm;
  } else {  // the shared expert: a plain matrix
    p.qs = a.sw2_qs; p.sc = a.sw2_sc; p.hm = a.sw2_hm; p.dm = a.sw2_dm;
  }
  p.qs2 = p.sc2 = p.hm2 = p.dm2 = nullptr;
  p.scale = p.scale2 = nullptr;
  return kq_rsrc<QT, false>(p);
}

}  // namespace

template <int QT, int UA, int UB>
__global__ __launch_bounds__(1024) void moe_ffn_kernel(const MoeFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ int flag_s;
  constexpr int NW = 16;
  constexpr bool Q2 = QT == DSK_QUANT_Q2_K;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int bid = blockIdx.x, G = gridDim.x;
  uint8_t* actA = smem;
  uint8_t* actB = smem + a.lds_a;
  const int K = a.K, slots = K + (a.shared_n > 0 ? 1 : 0);

  // ---- prologue: the router left Q8_K(rmsnorm(x)) behind (previous launch): copy it into item records ----
  {
    ActSrc S;
    S.act_mode = ACT_Q8; S.n = a.dim; S.a_qs = a.a_qs; S.a_d = a.a_d; S.a_bsums = a.a_bsums;
    S.a_f32 = nullptr; S.norm_w = nullptr; S.eps = 0.f; S.pre_scale = 0.f;
    stage_q8<Q2, NW>(S, actA, tid, scratch);
  }
  __syncthreads();

  // ---- phase A: w1/w3 GLU units ----
  {
    const int lpr_log2 = a.lprA_log2, RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    const int RG = NW * RPW;  // rows per unit (R = 1)
    const int nb = a.dim >> 8;
    for (int t = bid; t < K * a.UA; t += G) {
      const int s = t / a.UA, u = t - s * a.UA;
      const int e = a.route_e[s];
      const KQRsrc B = expert_rsrc13(a, e);
      const int row0 = u * RG + wave * RPW;
      const int rr = row0 + rloc;
      const bool valid = rr < a.mi;
      if (row0 < a.mi) {  // wave-uniform
        int rowblk[1] = {(valid ? rr : a.mi - 1) * nb + (sub >> 2)};
        float acc[1], acc2[1];
        rows_dot_kq<QT, 1, UA, true>(B, nb * 4, sub, lpr_log2, sub & 3, rowblk, actA + sub * ITEM_LDS, acc, acc2);
        if (sub == 0 && valid)  // src/infer.cpp:859-872; write-through: the slot's consumers sit on other CUs
          __hip_atomic_store(a.hb + (size_t)s * a.hb_stride + rr, act_fn(acc[0], a.act) * acc2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.slot_ctr + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- phase B: W2 units (routed slots in k order, then the shared expert) + k-ordered combine ----
  {
    const int lpr_log2 = a.lprB_log2, RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    constexpr int R = 2;
    const int RG = NW * RPW * R;  // rows per unit
    for (int t = bid; t < slots * a.UB; t += G) {
      const int s = t / a.UB, g = t - s * a.UB;
      const bool routed = s < K;
      const int n = routed ? a.mi : a.shared_n;
      const int e = routed ? a.route_e[s] : 0;
      const KQRsrc B = w2_rsrc<QT>(a, s, e);
      const int nb = n >> 8, items = nb * 4;
      const int its = (items + (1 << lpr_log2) - 1) >> lpr_log2;
      const int base = g * RG;
      const int row0 = base + wave * (RPW * R);
      int row[R], rowblk[R];
      bool valid[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int rr = row0 + r * RPW + rloc;
        valid[r] = rr < a.dim;
        row[r] = valid[r] ? rr : a.dim - 1;
        rowblk[r] = row[r] * nb + (sub >> 2);
      }
      const bool has_rows = row0 < a.dim;
      // the unit's first weight chunk is requested BEFORE the hand-off: it streams while the slot's producers finish
      ChunkKQ<QT, R, UB, false> c;
      if (has_rows) load_chunk_kq<QT, R, UB, false>(c, B, its, items, sub, lpr_log2, sub & 3, rowblk, 0);
      if (routed) {  // wait until every phase-A unit of slot s has published its rows of h_s
        if (tid == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(a.slot_ctr + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.UA) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) { *a.err = 1u; break; }
          }
          // the slot's last consumer to pass re-arms both counters (every consumer has seen the full count by then)
          const unsigned old = __hip_atomic_fetch_add(a.slot_pass + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == (unsigned)a.UB - 1) {
            __hip_atomic_store(a.slot_ctr + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.slot_pass + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      __syncthreads();  // (also: every wave is done with actB of the previous unit)
      // stage h_s: sc1 loads (written by other CUs during THIS launch), Q8_K per 256-block (quantize_row_q8_K_ref)
      const float* hsrc = a.hb + (size_t)s * a.hb_stride;
      for (int b = wave; b < nb; b += NW) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __hip_atomic_load(hsrc + b * 256 + lane * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q8k_block_lds<Q2>(v, lane, actB + (size_t)b * 4 * ITEM_LDS);
      }
      __syncthreads();
      if (a.tap_qs && g == 0)  // parity tap: what this slot staged
        dump_staged_q8<Q2>(actB, n, a.tap_qs + (size_t)s * a.tap_stride, a.tap_d + (size_t)s * (a.tap_stride >> 8), tid, 1024);
      float acc[R], acc2[R];
      if (has_rows) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = acc2[r] = 0.f;
        const uint8_t* lds_lane = actB + sub * ITEM_LDS;
        for (int it0 = 0; it0 < its; it0 += UB) {
          if (it0 > 0) load_chunk_kq<QT, R, UB, false>(c, B, its, items, sub, lpr_log2, sub & 3, rowblk, it0);
          compute_chunk_kq<QT, R, UB, false>(c, its, items, sub, lpr_log2, sub & 3, it0, lds_lane, acc, acc2);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = lanes_sum(acc[r], lpr_log2);
        if (sub == 0) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (valid[r]) __hip_atomic_store(a.eout + (size_t)s * a.dim + row[r], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // ---- combine: the last slot to finish this row group adds x += w_k * o_k (k order), then the shared expert ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.comb_ctr + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_s = old == (unsigned)slots - 1;
        if (flag_s) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(a.comb_ctr + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
        }
      }
      __syncthreads();
      if (flag_s) {
        const int hi = min(a.dim, base + RG);
        for (int rr = base + tid; rr < hi; rr += 1024) {
          float xv = a.x[rr];
          for (int k = 0; k < K; ++k) {  // src/infer.cpp:874-877
            const float v = __hip_atomic_load(a.eout + (size_t)k * a.dim + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xv = fmaf(v, a.route_w[k], xv);
          }
          if (slots > K) xv += __hip_atomic_load(a.eout + (size_t)K * a.dim + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // :900-903
          a.x[rr] = xv;
        }
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------
int moe_ffn_plan(MoeFfnArgs& a, int n_cus) {
  const bool q3 = a.quant == DSK_QUANT_Q3_K;
  if (a.quant != DSK_QUANT_Q2_K && !q3) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: k-quants only");
  if (a.dim % 256 || a.mi % 256 || a.shared_n % 256) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: lengths must be multiples of 256");
  const int itemsA = a.dim / 64, itemsB = a.mi / 64, itemsS = a.shared_n / 64;
  const int lprA = 1 << a.lprA_log2, lprB = 1 << a.lprB_log2;
  if (lprA < 4 || lprB < 4 || itemsA % lprA || itemsB % lprB || (a.shared_n > 0 && itemsS % lprB))
    DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: lanes per row %d / %d do not divide the rows", lprA, lprB);
  const int RGA = 16 * (64 / lprA), RGB = 16 * (64 / lprB) * 2;
  a.UA = (a.mi + RGA - 1) / RGA;
  a.UB = (a.dim + RGB - 1) / RGB;
  if (a.UB > a.comb_ctr_cap) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: %d row groups", a.UB);
  a.lds_a = (int)(((size_t)(a.dim / 64) * ITEM_LDS + 15) & ~(size_t)15);
  const int nB = a.mi > a.shared_n ? a.mi : a.shared_n;
  a.lds_b = (int)(((size_t)(nB / 64) * ITEM_LDS + 15) & ~(size_t)15);
  if (a.lds_a + a.lds_b > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: activations do not fit LDS");
  // every workgroup must be resident at once (the slot hand-off spins): one 16-wave workgroup per CU
  int grid = a.K * a.UA;
  const int unitsB = (a.K + (a.shared_n > 0 ? 1 : 0)) * a.UB;
  if (unitsB > grid) grid = unitsB;
  if (grid > n_cus) grid = n_cus;
  a.grid = grid;
  return DSK_OK;
}

int launch_moe_ffn(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const size_t lds = (size_t)a.lds_a + a.lds_b;
  auto go = [&](auto k) {
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(k, dim3(a.grid), dim3(1024), (uint32_t)lds, st, ev_start, ev_stop, 0u, a);
    else hipLaunchKernelGGL(k, dim3(a.grid), dim3(1024), lds, st, a);
  };
  if (a.quant == DSK_QUANT_Q2_K) go(moe_ffn_kernel<DSK_QUANT_Q2_K, 4, 4>);
  else go(moe_ffn_kernel<DSK_QUANT_Q3_K, 2, 2>);  // (more column steps in flight spill at 16 waves x 128 VGPRs: gemv_plan's caps)
  return DSK_OK;
}
