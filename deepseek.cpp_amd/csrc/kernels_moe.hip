// kernels_moe.hip -- the routed experts of one MoE block in ONE launch (K-quant models, one GPU).
//
// Replaces, for the selected experts e_0..e_{K-1} of a token (src/infer.cpp:853-878, 899-903):
//     h_k  = act(W1[e_k] xb) * (W3[e_k] xb)          matmul_expert x2 + GLU
//     o_k  = W2[e_k] h_k                              matmul_expert (h_k re-quantised to Q8_K, src/infer.cpp:325-336)
//     x   += w_k * o_k  (k order),  x += shared_out   (the shared expert's W2 runs here too)
// The two-launch form (gemv w1/w3 -> gemv w2 + combine) pays a kernel boundary and a full prologue between the halves,
// and its second half is a chain of dependent memory round trips (stage, stream, publish, arrive, combine) on small
// workgroups.  Here one resident grid of 16-wave workgroups (one per CU) runs both halves:
//   phase A  flat list of (slot, 64-row unit) pairs of the w1/w3 GLU; a finished unit publishes its rows of h_k
//            (write-through stores) and arrives on the slot's counter (no-return atomics);
//   phase B  every workgroup owns a slice of the ROWS of x for ALL slots: its W2 rows of the K experts and of the
//            shared expert are REQUESTED FIRST (they depend on the routing only), then it waits until all slots have
//            arrived, stages the K+1 hidden vectors (sc1 loads -> Q8_K in LDS), multiplies, and adds
//            x += w_k o_k in k order, then the shared expert, for its own rows: no cross-workgroup combine at all.
// The slot counters are re-armed by the router launch of the same block (its gate workgroup; stream order makes that
// safe), so the hot path has no atomic with a return value.
// Per-row arithmetic (lanes per row, column-step order, reduction trees, Q8_K staging, combine order) is the same code
// with the same parameters as the two-launch form: results are BIT-identical to it (tests/test_fused_moe_gpu.py).
// Every workgroup of the grid is resident (grid <= CUs), producers never wait, the one spin is bounded (err flag).
#include "dsk_internal.h"
#include "gemv_device.h"

#ifndef MOE_EXACT_A
#define MOE_EXACT_A 1  // -DMOE_EXACT_A=0: the generic chunk loop everywhere (A/B builds)
#endif

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

// W2: ONE descriptor over the whole expert stack (a lane adds its expert's block offset to its row offset, so the rows
// of one wave may belong to different experts), or over the shared expert's plain matrix
template <int QT>
DEV KQRsrc w2_rsrc(const MoeFfnArgs& a, bool shared) {
  WPtr p;
  p.present = true;
  if (!shared) { p.qs = a.w2_qs; p.sc = a.w2_sc; p.hm = a.w2_hm; p.dm = a.w2_dm; }
  else { p.qs = a.sw2_qs; p.sc = a.sw2_sc; p.hm = a.sw2_hm; p.dm = a.sw2_dm; }
  p.qs2 = p.sc2 = p.hm2 = p.dm2 = nullptr;
  p.scale = p.scale2 = nullptr;
  return kq_rsrc<QT, false>(p);
}

}  // namespace

template <int QT, int UA, int UB>
__global__ __launch_bounds__(1024) void moe_ffn_kernel(const MoeFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  constexpr int NW = 16;
  constexpr bool Q2 = QT == DSK_QUANT_Q2_K;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int bid = blockIdx.x, G = a.grid;
  if (bid >= G) {  // tail prefetch workgroups (MoeFfnArgs::pf_wgs): run when the first workgroups of the grid have left
    tail_prefetch(a.pf_p, a.pf_n, tid, 1024);
    return;
  }
  uint8_t* actA = smem;
  uint8_t* actB = smem + a.lds_a;                                            // slots x lds_b item records
  float* o_s = reinterpret_cast<float*>(smem + a.lds_a + (size_t)a.lds_b * (a.K + 1));  // [slot][rows_wg] slot outputs
  const int K = a.K, slots = K + (a.shared_n > 0 ? 1 : 0);
  unsigned long long* tl = a.timeline && bid < DSK_TL_WGS ? a.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = wall_clock64();

  // ---- prologue: the router left Q8_K(rmsnorm(x)) behind (previous launch): copy it into item records ----
  {
    ActSrc S;
    S.act_mode = ACT_Q8; S.n = a.dim; S.a_qs = a.a_qs; S.a_d = a.a_d; S.a_bsums = a.a_bsums;
    S.a_f32 = nullptr; S.norm_w = nullptr; S.eps = 0.f; S.pre_scale = 0.f;
    stage_q8<Q2, NW>(S, actA, tid, scratch);
  }
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();

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
        // DeepSeek-V3 Q2_K (7168-wide rows, 16 lanes each = 7 column steps known at compile time): the software-pipelined
        // straight-line form (gemv_device.h rows_dot_kq_exact): phase A 17.5 -> 16.1 us, same bits
        if (QT == DSK_QUANT_Q2_K && MOE_EXACT_A && lpr_log2 == 4 && nb == 28)
          rows_dot_kq_exact<DSK_QUANT_Q2_K, 1, true, 7, 4>(B, sub, sub & 3, rowblk, actA + sub * ITEM_LDS, acc, acc2);
        else
          rows_dot_kq<QT, 1, UA, true>(B, nb * 4, sub, lpr_log2, sub & 3, rowblk, actA + sub * ITEM_LDS, acc, acc2);
        if (sub == 0 && valid)  // src/infer.cpp:859-872; write-through: the consumers sit on other CUs
          __hip_atomic_store(a.hb + (size_t)s * a.hb_stride + rr, act_fn(acc[0], a.act) * acc2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival
      __syncthreads();
      if (!a.hq_qs) {
        if (tid == 0) __hip_atomic_fetch_add(a.slot_ctr + s * MOE_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (wave == NW - 1) {
        // Q8_K hand-over.  The units of a 256-block of h_k (256 / RG of them) arrive on the block's counter; the LAST one
        // quantises the block - one wave, quantize_row_q8_K_ref on the f32 values every unit published (sc1 loads: the
        // other units ran on other CUs) -, publishes codes / sums / scale write-through, re-arms the block counter and
        // only then arrives on the slot's counter (which counts blocks).  Only THIS wave waits for the returning atomic:
        // the other 15 go on to phase B and request their W2 rows.
        const int upb = 256 / RG > 0 ? 256 / RG : 1;   // units per block (RG = 16 x rows per wave divides 256)
        const int blk = (u * RG) >> 8;                 // block of h_k this unit belongs to (RG > 256 never happens: 16 x 16)
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(a.blk_ctr + s * (a.mi >> 8) + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == (unsigned)upb - 1) {
          if (lane == 0) __hip_atomic_store(a.blk_ctr + s * (a.mi >> 8) + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
          // (ONE 16-byte sc1 load per lane: four dword sc1 loads are four fabric reads each)
          const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.hb), (int)(((size_t)s * a.hb_stride + blk * 256 + lane * 4) * 4), 0, 16);
          const u32 h0 = hv.x, h1 = hv.y, h2 = hv.z, h3 = hv.w;
          const float v[4] = {u2f(h0), u2f(h1), u2f(h2), u2f(h3)};
          const size_t e0 = (size_t)s * a.hb_stride + blk * 256;
          ad::q8k_block_wt(v, lane, a.hq_qs + e0, a.hq_d + (e0 >> 8), a.hq_bsums + (e0 >> 4));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(a.slot_ctr + s * MOE_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }

  if (tl && tid == 0) tl[2] = wall_clock64();
  // ---- phase B: this workgroup's rows [r_lo, r_hi) of x, for all slots ----
  {
    const int lpr_log2 = a.lprB_log2, RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    const int nrows_max = a.rows_wg;
    const int r_lo = bid * nrows_max;
    const int nrows = max(0, min(a.dim, r_lo + nrows_max) - r_lo);
    // jobs: a wave handles two RPW-row steps per job (two independent rows per lane).  Routed steps walk the virtual rows
    // (slot, row) slot-major; the shared expert's steps follow (their own descriptor: a job never mixes the two kinds).
    const int WR = (nrows * K + RPW - 1) / RPW, JR = (WR + 1) >> 1;
    const int WS = slots > K ? (nrows + RPW - 1) / RPW : 0, JS = (WS + 1) >> 1;
    const int nbR = a.mi >> 8, nbS = a.shared_n >> 8;
    const KQRsrc BR = w2_rsrc<QT>(a, false);
    const KQRsrc BS = w2_rsrc<QT>(a, slots > K);
    struct Job { bool shared; int n_items, its; bool valid[2]; int slot[2], rr[2]; int rowblk[2][1]; };
    auto make_job = [&](int j) {
      Job J;
      J.shared = j >= JR;
      const int nb = J.shared ? nbS : nbR;
      J.n_items = nb * 4;
      J.its = (J.n_items + (1 << lpr_log2) - 1) >> lpr_log2;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int step = (J.shared ? j - JR : j) * 2 + r;
        const int v = step * RPW + rloc;
        if (J.shared) {
          J.valid[r] = v < nrows;
          J.slot[r] = K;
          J.rr[r] = J.valid[r] ? v : 0;
          J.rowblk[r][0] = (r_lo + J.rr[r]) * nb + (sub >> 2);
        } else {
          J.valid[r] = v < nrows * K;
          const int vv = J.valid[r] ? v : 0;
          J.slot[r] = vv / (nrows > 0 ? nrows : 1);
          J.rr[r] = vv - J.slot[r] * nrows;
          const int e = a.route_e[J.slot[r]];
          J.rowblk[r][0] = (e * a.dim + r_lo + J.rr[r]) * nb + (sub >> 2);  // expert e's rows inside the stack
        }
      }
      return J;
    };
    const int n_jobs = nrows > 0 ? JR + JS : 0;
    float xv = 0.f;
    if (tid < nrows) xv = a.x[r_lo + tid];
    // wait until every phase-A unit of every slot has published its rows (lane k of wave 0 watches slot k)
    const unsigned slot_target = a.hq_qs ? (unsigned)(a.mi >> 8) : (unsigned)a.UA;  // arrivals per slot: blocks, or units
    auto wait_slots = [&]() {
      if (wave == 0) {
        unsigned spins = 0;
        // (an earlier launch of this token already gave up - a DEVICE word next to the counters says so; the host-visible word lives
      // in pinned host memory and must not be read here: 256 reads over PCIe cost the launch 15 us - : the host will re-run the token,
      // do not spin the limit out again in every layer)
        if (__hip_atomic_load(a.slot_ctr + MOE_GAVE_UP_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = (unsigned)a.spin_limit > 64u ? (unsigned)a.spin_limit - 64u : 0u;
        for (;;) {
          const bool ok = lane >= K || __hip_atomic_load(a.slot_ctr + lane * MOE_CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slot_target;
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(1);  // (8 or 32, or ONE counter for all slots: no difference in the time to pass)
          if (++spins > (unsigned)a.spin_limit) {
          if (lane == 0) { *a.err = 1u; __hip_atomic_store(a.slot_ctr + MOE_GAVE_UP_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        }
        if (a.spin_limit < 0 && bid == 0 && lane == 0) *a.err = 1u;  // fault injection (option "moe_spin_limit" < 0): tests/test_fused_moe_gpu.py
      }
      __syncthreads();
      if (tl && tid == 0) tl[3] = wall_clock64();
    };
    // stage the hidden vectors: 16-byte loads that bypass this CU's L1 (written by other CUs during THIS launch), all of
    // a wave's blocks requested before the first is quantised (one memory round trip, not one per block); Q8_K per
    // 256-block (quantize_row_q8_K_ref).  `more` runs between the requests and the quantisation: further weight requests
    // queue up behind the hidden vectors' and stream while the blocks are quantised.
    // hidden vectors handed over as Q8_K: routed slots are COPIED into item records (one 16-byte run + its sum per thread and
    // pass, sc1 loads: written by other CUs during this launch); the shared expert's f32 vector (the router launch wrote it)
    // is quantised here as before - it is staged BEFORE the hand-off, in the shadow of the W2 requests
    auto stage_routed_q8 = [&]() {
      const rsrc_t qr = make_rsrc(a.hq_qs), br = make_rsrc(a.hq_bsums), dr = make_rsrc(a.hq_d);
      const int runs_per_slot = a.mi >> 4, nruns = K * runs_per_slot;
      for (int i = tid; i < nruns; i += NW * 64) {
        const int s = i / runs_per_slot, r = i - s * runs_per_slot;    // run r of slot s = sub-block r
        const int b = r >> 4, j = r & 15, h = j >> 3, sidx = (j >> 1) & 3, lh = j & 1;
        const int e16 = s * (a.hb_stride >> 4) + r;                    // index of the run in the [slot][stride] arrays
        const u32x4 codes = __builtin_amdgcn_raw_buffer_load_b128(qr, e16 * 16, 0, 16);
        const int bs = (int)(short)__builtin_amdgcn_raw_buffer_load_b16(br, e16 * 2, 0, 16);
        uint8_t* rec = actB + (size_t)s * a.lds_b + (size_t)(b * 4 + 2 * h + lh) * ITEM_LDS;
        *reinterpret_cast<u32x4*>(rec + sidx * 16) = codes;
        if (Q2) {
          rec[64 + sidx] = (uint8_t)(bs >> 8);
          rec[68 + sidx] = (uint8_t)(bs & 0xff);
        } else {
          reinterpret_cast<short*>(rec + 64)[sidx] = (short)bs;
        }
      }
      for (int i = tid; i < K * nbR * 4; i += NW * 64) {  // block scales: one per quarter record
        const int s = i / (nbR * 4), q4 = i - s * (nbR * 4);
        const float d = u2f(__builtin_amdgcn_raw_buffer_load_b32(dr, (s * (a.hb_stride >> 8) + (q4 >> 2)) * 4, 0, 16));
        float* mrec = reinterpret_cast<float*>(actB + (size_t)s * a.lds_b + (size_t)q4 * ITEM_LDS + 72);
        if (Q2) { mrec[0] = d * 0.0625f; mrec[1] = d; }
        else mrec[0] = d;
      }
    };
    auto stage_hidden = [&](bool routed, bool shared, auto&& more) {
      const int nblk = (routed ? K * nbR : 0) + (shared ? nbS : 0);
      const int boff = routed ? 0 : K * nbR;  // (block index space: routed slots first, then the shared expert)
      const rsrc_t hr = make_rsrc(a.hb);
      constexpr int SB = 5;
      constexpr int HAUX = 16;  // sc1
      for (int b0 = wave; b0 < nblk; b0 += NW * SB) {
        u32x4 hv[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int b = b0 + k * NW + boff;
          if (b0 + k * NW < nblk) {
            const int s = b < K * nbR ? b / nbR : K, bb = b < K * nbR ? b - s * nbR : b - K * nbR;
            hv[k] = __builtin_amdgcn_raw_buffer_load_b128(hr, (s * a.hb_stride + bb * 256 + lane * 4) * 4, 0, HAUX);
          }
        }
        if (b0 == wave) more();
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int b = b0 + k * NW + boff;
          if (b0 + k * NW < nblk) {
            const int s = b < K * nbR ? b / nbR : K, bb = b < K * nbR ? b - s * nbR : b - K * nbR;
            const u32 w0 = hv[k].x, w1 = hv[k].y, w2 = hv[k].z, w3 = hv[k].w;
            const float v[4] = {u2f(w0), u2f(w1), u2f(w2), u2f(w3)};
            q8k_block_lds<Q2>(v, lane, actB + (size_t)s * a.lds_b + (size_t)bb * 4 * ITEM_LDS);
          }
        }
      }
    };
    auto staged = [&]() {
      __syncthreads();
      if (tl && tid == 0) tl[4] = wall_clock64();
      if (a.tap_qs && bid == 0)  // parity tap: what the slots staged
        for (int s = 0; s < slots; ++s)
          dump_staged_q8<Q2>(actB + (size_t)s * a.lds_b, s < K ? a.mi : a.shared_n, a.tap_qs + (size_t)s * a.tap_stride,
                             a.tap_d + (size_t)s * (a.tap_stride >> 8), tid, 1024);
    };
    auto finish_job = [&](const Job& J, float o0, float o1) {
      if (sub == 0) {
        if (J.valid[0]) { o_s[J.slot[0] * nrows_max + J.rr[0]] = o0; a.eout[(size_t)J.slot[0] * a.dim + r_lo + J.rr[0]] = o0; }
        if (J.valid[1]) { o_s[J.slot[1] * nrows_max + J.rr[1]] = o1; a.eout[(size_t)J.slot[1] * a.dim + r_lo + J.rr[1]] = o1; }
      }
    };
    {
      // (All of the workgroup's W2 rows in registers before the hand-off / right behind the hidden vectors' requests - two
      // jobs per wave as four 2-step chunks - was measured: 36.1 us against 35.7; so was reading the hidden vectors with
      // plain loads after an agent acquire, so that an XCD's L2 serves its 32 workgroups: 36.2.  From the end of phase A
      // to the exit this launch moves its 61 MB (W2 + 256 copies of the hidden vectors) at the same ~18 GB/s per CU as
      // phase A moves the w1/w3 rows.  With ALL W2 rows requested before the hand-off (8 us ahead of their use) the
      // multiplies after the staging still take 5.4 us: they are VALU-bound - 252 virtual rows x 32 items per workgroup at
      // ~70-75 wave instructions per item on 4 SIMDs - and nothing is left to stream underneath them.)
      // (Round 3, measured and rejected: the hand-over WITHOUT counters - every value an 8-byte granule {payload, tag}, the
      // block's last unit polls the block's granules, quantises, publishes tagged Q8_K granules, phase B polls those; no drain,
      // no barrier, no arrival.  On an idle chip a hop costs 2.1 us this way against 4.3 us (tools/ll_probe.hip).  In situ,
      // bit-identical and 36.9 us (all threads poll), 34.6 us (one wave polls the blocks' scale granules), 33-35 us with the
      // producers' W2 requests held back - against 33.8: a CU serves its waves' memory requests in order, so whatever is
      // polled queues up behind the 128 KB of W2 rows requested below, and a workgroup's exit is its last w1 / w3 row plus
      // ~270 KB of W2 rows and hidden vectors at ~20 GB/s per CU whichever way the hand-off is signalled.  The wait is not
      // idle time: it is W2 streaming.)
      // the first job's weights are requested BEFORE the hand-off: they stream while the slots' producers finish
      ChunkKQ<QT, 1, UB, false> c0, c1;
      Job J = make_job(wave < n_jobs ? wave : 0);
      // (Round 3, measured and rejected: the DeepSeek-V3 shape of this phase - 2048-wide rows, 8 lanes per row, ONE chunk of
      // 4 column steps - passed as compile-time constants, so that the bounds, ragged-step selects and record addressing of
      // the generic chunk code fold away (~10 of ~72 VALU instructions per item in a VALU-bound stretch): the launch went
      // 33.7 -> 37.9 us, every workgroup passing the hand-off 2 us later: hipcc schedules the now straight-line requests and
      // the staging differently.  Kept generic.)
      if (wave < n_jobs) {
        load_chunk_kq<QT, 1, UB, false>(c0, J.shared ? BS : BR, J.its, J.n_items, sub, lpr_log2, sub & 3, J.rowblk[0], 0);
        load_chunk_kq<QT, 1, UB, false>(c1, J.shared ? BS : BR, J.its, J.n_items, sub, lpr_log2, sub & 3, J.rowblk[1], 0);
      }
      if (a.hq_qs) {
        if (slots > K) stage_hidden(false, true, []() {});  // the shared expert's vector is ready since the router launch
        wait_slots();
        stage_routed_q8();
      } else {
        wait_slots();
        stage_hidden(true, slots > K, []() {});
      }
      staged();
      for (int j = wave; j < n_jobs; j += NW) {
        if (j != wave) {
          J = make_job(j);
          load_chunk_kq<QT, 1, UB, false>(c0, J.shared ? BS : BR, J.its, J.n_items, sub, lpr_log2, sub & 3, J.rowblk[0], 0);
          load_chunk_kq<QT, 1, UB, false>(c1, J.shared ? BS : BR, J.its, J.n_items, sub, lpr_log2, sub & 3, J.rowblk[1], 0);
        }
        const KQRsrc& B = J.shared ? BS : BR;
        float acc0[1] = {0.f}, acc1[1] = {0.f}, dummy[1] = {0.f};
        const uint8_t* l0 = actB + (size_t)J.slot[0] * a.lds_b + sub * ITEM_LDS;
        const uint8_t* l1 = actB + (size_t)J.slot[1] * a.lds_b + sub * ITEM_LDS;
        for (int it0 = 0; it0 < J.its; it0 += UB) {
          if (it0 > 0) {
            load_chunk_kq<QT, 1, UB, false>(c0, B, J.its, J.n_items, sub, lpr_log2, sub & 3, J.rowblk[0], it0);
            load_chunk_kq<QT, 1, UB, false>(c1, B, J.its, J.n_items, sub, lpr_log2, sub & 3, J.rowblk[1], it0);
          }
          compute_chunk_kq<QT, 1, UB, false>(c0, J.its, J.n_items, sub, lpr_log2, sub & 3, it0, l0, acc0, dummy);
          compute_chunk_kq<QT, 1, UB, false>(c1, J.its, J.n_items, sub, lpr_log2, sub & 3, it0, l1, acc1, dummy);
        }
        finish_job(J, lanes_sum(acc0[0], lpr_log2), lanes_sum(acc1[0], lpr_log2));
      }
    }
    __syncthreads();
    if (tl && tid == 0) tl[5] = wall_clock64();
    if (tid < nrows) {  // x += w_k * o_k in k order (src/infer.cpp:874-877), then the shared expert (:900-903)
      for (int k = 0; k < K; ++k) xv = fmaf(o_s[k * nrows_max + tid], a.route_w[k], xv);
      if (slots > K) xv += o_s[K * nrows_max + tid];
      a.x[r_lo + tid] = xv;
    }
    if (tl && tid == 0) tl[6] = wall_clock64();
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same launch for float weights (F8E5M2 with 128 x 128 block scales, F16, F32; src/infer.cpp:238-313, 423-469): f32
// activations in LDS instead of Q8_K item records, rows_dot_f (f32 FMAs like the reference) instead of the integer dots,
// and the shared expert's w1 / w3 as further phase-A units (these models have no rider in the router launch).  Same
// phases, same hand-off, same combine; lanes per row come from the two-launch plans, so the per-row sums are theirs.
// ------------------------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(1024) void moe_ffn_f_kernel(const MoeFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  constexpr int NW = 16;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int bid = blockIdx.x, G = a.grid;
  if (bid >= G) {  // tail prefetch workgroups (MoeFfnArgs::pf_wgs)
    tail_prefetch(a.pf_p, a.pf_n, tid, 1024);
    return;
  }
  float* actA = reinterpret_cast<float*>(smem);                                  // rmsnorm(x), dim floats
  float* actB = reinterpret_cast<float*>(smem + a.lds_a);                        // [slot][lds_b / 4] hidden vectors
  float* o_s = reinterpret_cast<float*>(smem + a.lds_a + (size_t)a.lds_b * (a.K + 1));  // [slot][rows_wg] slot outputs
  const int K = a.K, slots = K + (a.shared_n > 0 ? 1 : 0);
  unsigned long long* tl = a.timeline && bid < DSK_TL_WGS ? a.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = wall_clock64();
  {  // prologue: rmsnorm(x, ffn_norm) (src/infer.cpp:839, 601-611) into LDS, like the two-launch form's w1/w3 launch
    ActSrc S;
    S.act_mode = ACT_F32_NORM; S.n = a.dim; S.a_f32 = a.x; S.norm_w = a.norm_w; S.eps = a.eps; S.pre_scale = 0.f;
    S.a_qs = nullptr; S.a_d = nullptr; S.a_bsums = nullptr;
    stage_f32<NW>(S, actA, tid, scratch);
  }
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();
  // ---- phase A: the w1/w3 GLU rows of the K routed slots and of the shared expert, concatenated slot-major into one
  // virtual row space that the workgroups split EVENLY (to the row: V2-Lite's 11 264 rows are 44 per workgroup; dealing
  // 16-row units left some workgroups three units and others two).  A lane resolves its own (slot, row): the rows of one
  // wave may belong to different experts.  Arrivals count ROWS: each workgroup adds, per slot it touched, the rows it
  // finished; a slot is complete at mi (shared_n) rows. ----
  {
    const int lpr_log2 = a.lprA_log2, RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    const int RG = NW * RPW;
    const int vtotal = K * a.mi + a.shared_n;
    const int v_lo = (int)((long long)vtotal * bid / G), v_hi = (int)((long long)vtotal * (bid + 1) / G);
    for (int base = v_lo; base < v_hi; base += RG) {
      const int row0 = base + wave * RPW;
      if (row0 < v_hi) {  // wave-uniform
        const int vr = row0 + rloc;
        const bool valid = vr < v_hi;
        const int vv = valid ? vr : v_hi - 1;
        const bool sh = vv >= K * a.mi;
        const int s = sh ? K : vv / a.mi, rr = sh ? vv - K * a.mi : vv - s * a.mi;
        WPtr P;
        P.present = true;
        P.sc = P.hm = P.dm = P.sc2 = P.hm2 = P.dm2 = nullptr;
        if (sh) {
          P.qs = a.sw1_qs; P.qs2 = a.sw3_qs; P.scale = a.sw1_scale; P.scale2 = a.sw3_scale;
        } else {
          const int e = a.route_e[s];
          P.qs = a.w1_qs + (size_t)e * a.e13_qs; P.qs2 = a.w3_qs + (size_t)e * a.e13_qs;
          P.scale = a.w1_scale ? a.w1_scale + (size_t)e * a.e13_scale : nullptr;
          P.scale2 = a.w3_scale ? a.w3_scale + (size_t)e * a.e13_scale : nullptr;
        }
        int row[1] = {rr};
        float acc[1], acc2[1];
        rows_dot_f<QT, 1, 4, true>(P, a.dim, a.b0, a.b1, lpr_log2, lane, row, reinterpret_cast<const uint8_t*>(actA), acc, acc2);
        if (sub == 0 && valid)  // src/infer.cpp:859-872, 882-897; write-through: the consumers sit on other CUs
          __hip_atomic_store(a.hb + (size_t)s * a.hb_stride + rr, act_fn(acc[0], a.act) * acc2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrivals
    __syncthreads();
    if (tid < slots) {  // thread s: this workgroup's rows of slot s
      const int s_lo = tid * a.mi, s_hi = tid < K ? s_lo + a.mi : s_lo + a.shared_n;
      const int lo = v_lo > s_lo ? v_lo : s_lo, hi = v_hi < s_hi ? v_hi : s_hi;
      if (hi > lo) __hip_atomic_fetch_add(a.slot_ctr + tid * MOE_CTR_STRIDE, (unsigned)(hi - lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tl && tid == 0) tl[2] = wall_clock64();
  // ---- phase B: this workgroup's rows of x, for all slots ----
  {
    const int lpr_log2 = a.lprB_log2, RPW = 64 >> lpr_log2;
    const int rloc = lane >> lpr_log2, sub = lane & ((1 << lpr_log2) - 1);
    const int nrows_max = a.rows_wg;
    const int r_lo = bid * nrows_max;
    const int nrows = max(0, min(a.dim, r_lo + nrows_max) - r_lo);
    float xv = 0.f;
    if (tid < nrows) xv = a.x[r_lo + tid];
    // routed steps over the virtual rows (slot, row) slot-major, then the shared expert's steps (a step never mixes the two
    // kinds: the row lengths differ).  A lane resolves its own expert: the rows of one wave may belong to different slots.
    const int WR = (nrows * K + RPW - 1) / RPW, WS = slots > K ? (nrows + RPW - 1) / RPW : 0;
    struct StepF { WPtr P; int n, slot, rr; bool valid; };
    auto make_step = [&](int st) {
      StepF S;
      const bool sh = st >= WR;
      const int v = (sh ? st - WR : st) * RPW + rloc;
      S.valid = st < WR + WS && (sh ? v < nrows : v < nrows * K);
      const int vv = S.valid ? v : 0;
      S.slot = sh ? K : vv / (nrows > 0 ? nrows : 1);
      S.rr = sh ? vv : vv - S.slot * nrows;
      S.n = sh ? a.shared_n : a.mi;
      S.P.present = true;
      S.P.sc = S.P.hm = S.P.dm = S.P.qs2 = S.P.sc2 = S.P.hm2 = S.P.dm2 = nullptr;
      S.P.scale2 = nullptr;
      if (sh) {
        S.P.qs = a.sw2_qs; S.P.scale = a.sw2_scale;
      } else {
        const int e = a.route_e[S.slot];
        S.P.qs = a.w2_qs + (size_t)e * a.e2_qs;
        S.P.scale = a.w2_scale ? a.w2_scale + (size_t)e * a.e2_scale : nullptr;
      }
      return S;
    };
    // the first step's first 8 column steps are requested BEFORE the hand-off: they depend on the routing only
    constexpr int UB = 8;
    ChunkF<QT, 1, UB, false> c0;
    StepF S0 = make_step(wave);
    const int row0[1] = {r_lo + S0.rr};
    if (wave < WR + WS) load_chunk_f<QT, 1, UB, false>(c0, S0.P, S0.n, a.b0, a.b1, lpr_log2, lane, row0, 0);
    // hand-off: lane k of wave 0 watches slot k's counter (the shared expert's is slot K); bounded like the K-quant kernel's
    if (wave == 0) {
      unsigned spins = 0;
      // (an earlier launch of this token already gave up - a DEVICE word next to the counters says so; the host-visible word lives
      // in pinned host memory and must not be read here: 256 reads over PCIe cost the launch 15 us - : the host will re-run the token,
      // do not spin the limit out again in every layer)
      if (__hip_atomic_load(a.slot_ctr + MOE_GAVE_UP_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = (unsigned)a.spin_limit > 64u ? (unsigned)a.spin_limit - 64u : 0u;
      for (;;) {
        const unsigned want = lane < K ? (unsigned)a.mi : (unsigned)a.shared_n;
        const bool ok = lane >= slots || __hip_atomic_load(a.slot_ctr + lane * MOE_CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (unsigned)a.spin_limit) {
          if (lane == 0) { *a.err = 1u; __hip_atomic_store(a.slot_ctr + MOE_GAVE_UP_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
      }
      if (a.spin_limit < 0 && bid == 0 && lane == 0) *a.err = 1u;  // fault injection
    }
    __syncthreads();
    if (tl && tid == 0) tl[3] = wall_clock64();
    {  // the hidden vectors: 16-byte loads that bypass this CU's L1 (written by other CUs during this launch)
      const rsrc_t hr = make_rsrc(a.hb);
      const int per = a.lds_b >> 4;  // 16-byte pieces per slot in LDS
      for (int i = tid; i < slots * per; i += NW * 64) {
        const int s = i / per, j = i - s * per;
        const int n = s < K ? a.mi : a.shared_n;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (j * 4 < n) v = __builtin_amdgcn_raw_buffer_load_b128(hr, (s * a.hb_stride + j * 4) * 4, 0, 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(actB) + (size_t)s * a.lds_b + (size_t)j * 16) = v;
      }
    }
    __syncthreads();
    if (tl && tid == 0) tl[4] = wall_clock64();
    for (int st = wave; st < WR + WS; st += NW) {
      StepF S = st == wave ? S0 : make_step(st);
      const int row[1] = {r_lo + S.rr};
      const float* lx = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(actB) + (size_t)S.slot * a.lds_b);
      const int its = (S.n / FTraits<QT>::EPI + (1 << lpr_log2) - 1) >> lpr_log2;
      float acc[1] = {0.f}, acc2[1] = {0.f};
      for (int it0 = 0; it0 < its; it0 += UB) {  // column steps in order: the sums of rows_dot_f
        if (st != wave || it0 > 0) load_chunk_f<QT, 1, UB, false>(c0, S.P, S.n, a.b0, a.b1, lpr_log2, lane, row, it0);
        compute_chunk_f<QT, 1, UB, false>(c0, S.n, lpr_log2, it0, lx, acc, acc2);
      }
      const float o = lanes_sum(acc[0], lpr_log2);
      if (sub == 0 && S.valid) { o_s[S.slot * nrows_max + S.rr] = o; a.eout[(size_t)S.slot * a.dim + r_lo + S.rr] = o; }
    }
    __syncthreads();
    if (tl && tid == 0) tl[5] = wall_clock64();
    if (tid < nrows) {  // x += w_k * o_k in k order (src/infer.cpp:874-877), then the shared expert (:900-903)
      for (int k = 0; k < K; ++k) xv = fmaf(o_s[k * nrows_max + tid], a.route_w[k], xv);
      if (slots > K) xv += o_s[K * nrows_max + tid];
      a.x[r_lo + tid] = xv;
    }
    if (tl && tid == 0) tl[6] = wall_clock64();
  }
}

#ifndef MOE_UA
#define MOE_UA 4
#endif
// ---- host side ------------------------------------------------------------------------------------------
static int moe_ffn_plan_f(MoeFfnArgs& a, int n_cus) {
  const int epi = a.quant == DSK_QUANT_F32 ? 4 : (a.quant == DSK_QUANT_F16 ? 8 : 16);
  if (a.dim % 16 || a.mi % 4 || a.shared_n % 4 || a.dim % epi || a.mi % epi || a.shared_n % epi)
    DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn (float weights): row lengths %d / %d / %d", a.dim, a.mi, a.shared_n);
  const int lprA = 1 << a.lprA_log2;
  const int RG = 16 * (64 / lprA);
  a.UA = (a.mi + RG - 1) / RG;
  a.US = a.shared_n > 0 ? (a.shared_n + RG - 1) / RG : 0;
  int grid = a.K * a.UA + a.US;
  if (grid > n_cus) grid = n_cus;
  if (grid > a.dim) grid = a.dim;
  a.grid = grid;
  if (a.spin_limit == 0) a.spin_limit = 1 << 20;
  a.rows_wg = (a.dim + grid - 1) / grid;
  if (a.rows_wg > 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: %d rows per workgroup", a.rows_wg);
  a.lds_a = (a.dim * 4 + 15) & ~15;
  const int nB = a.mi > a.shared_n ? a.mi : a.shared_n;
  a.lds_b = (nB * 4 + 15) & ~15;
  a.lds_o = (a.K + 1) * a.rows_wg * 4;
  const size_t lds = (size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o;
  if (lds > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: activations do not fit LDS");
  int per_cu = 0;
  hipError_t e = hipErrorUnknown;
  auto occ = [&](auto k) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 1024, lds);
  };
  if (a.quant == DSK_QUANT_F32) occ(moe_ffn_f_kernel<DSK_QUANT_F32>);
  else if (a.quant == DSK_QUANT_F16) occ(moe_ffn_f_kernel<DSK_QUANT_F16>);
  else occ(moe_ffn_f_kernel<DSK_QUANT_F8E5M2>);
  if (e != hipSuccess || per_cu < 1) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: a workgroup is not resident on one CU (occupancy query: %d)", per_cu);
  return DSK_OK;
}

int moe_ffn_plan(MoeFfnArgs& a, int n_cus) {
  if (a.tiled) return moe_ffn_plan_tile(a, n_cus);
  const bool q3 = a.quant == DSK_QUANT_Q3_K;
  if (a.quant == DSK_QUANT_F32 || a.quant == DSK_QUANT_F16 || a.quant == DSK_QUANT_F8E5M2) return moe_ffn_plan_f(a, n_cus);
  if (a.quant != DSK_QUANT_Q2_K && !q3) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: bad quant %d", a.quant);
  if (a.dim % 256 || a.mi % 256 || a.shared_n % 256) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: lengths must be multiples of 256");
  const int itemsA = a.dim / 64, itemsB = a.mi / 64, itemsS = a.shared_n / 64;
  const int lprA = 1 << a.lprA_log2, lprB = 1 << a.lprB_log2;
  if (lprA < 4 || lprB < 4 || itemsA % lprA || itemsB % lprB || (a.shared_n > 0 && itemsS % lprB))
    DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: lanes per row %d / %d do not divide the rows", lprA, lprB);
  const int RGA = 16 * (64 / lprA), RGB = 16 * (64 / lprB) * 2;
  (void)RGB;
  a.UA = (a.mi + RGA - 1) / RGA;
  // the Q8_K hand-over counts the units of a 256-block: a unit must divide the block, and the block counters must exist
  if (a.hq_qs && (RGA > 256 || 256 % RGA != 0 || (long)a.K * (a.mi / 256) > MOE_BLK_CTRS)) a.hq_qs = nullptr;
  // one descriptor spans the W2 stack: a lane's byte offset (expert, row, block) must fit its 32-bit offset field
  if ((double)a.n_experts * a.dim * (a.mi / 256) * 64.0 >= 2147483648.0) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: the W2 stack exceeds the 31-bit offset of a buffer load");
  // every workgroup must be resident at once (phase B spins on the slot counters): one 16-wave workgroup per CU
  int grid = a.K * a.UA;
  if (grid > n_cus) grid = n_cus;
  if (grid > a.dim) grid = a.dim;
  a.grid = grid;
  if (a.spin_limit == 0) a.spin_limit = 1 << 20;
  a.rows_wg = (a.dim + grid - 1) / grid;  // rows of x a workgroup owns in phase B
  if (a.rows_wg > 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: %d rows per workgroup", a.rows_wg);
  a.lds_a = (int)(((size_t)(a.dim / 64) * ITEM_LDS + 15) & ~(size_t)15);
  const int nB = a.mi > a.shared_n ? a.mi : a.shared_n;
  a.lds_b = (int)(((size_t)(nB / 64) * ITEM_LDS + 15) & ~(size_t)15);
  a.lds_o = (a.K + 1) * a.rows_wg * 4;
  if ((size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: activations do not fit LDS");
  // the runtime's own view of residency: one 1024-thread workgroup with this much LDS must fit a CU.  (What the query
  // cannot see - a CU mask, another process on the GPU - is caught at run time: the bounded spin gives up and the model
  // falls back to the two-launch form, forward.cpp handoff_gave_up.)
  {
    const size_t lds = (size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o;
    int per_cu = 0;
    hipError_t e;
    if (q3) {
      auto k = moe_ffn_kernel<DSK_QUANT_Q3_K, 2, 2>;
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 1024, lds);
    } else {
      auto k = moe_ffn_kernel<DSK_QUANT_Q2_K, MOE_UA, 4>;
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 1024, lds);
    }
    if (e != hipSuccess || per_cu < 1) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: a workgroup is not resident on one CU (occupancy query: %d)", per_cu);
  }
  return DSK_OK;
}

int launch_moe_ffn(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (a.tiled) return launch_moe_ffn_tile(st, a, ev_start, ev_stop);
  const size_t lds = (size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o;
  auto go = [&](auto k) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), (uint32_t)lds, st, ev_start, ev_stop, 0u, a);
    else hipLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), lds, st, a);
  };
  if (a.quant == DSK_QUANT_Q2_K) go(moe_ffn_kernel<DSK_QUANT_Q2_K, MOE_UA, 4>);
  else if (a.quant == DSK_QUANT_F8E5M2) go(moe_ffn_f_kernel<DSK_QUANT_F8E5M2>);
  else if (a.quant == DSK_QUANT_F16) go(moe_ffn_f_kernel<DSK_QUANT_F16>);
  else if (a.quant == DSK_QUANT_F32) go(moe_ffn_f_kernel<DSK_QUANT_F32>);
  else go(moe_ffn_kernel<DSK_QUANT_Q3_K, 2, 2>);  // (more column steps in flight spill at 16 waves x 128 VGPRs: gemv_plan's caps)
  return DSK_OK;
}
