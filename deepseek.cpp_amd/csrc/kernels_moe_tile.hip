// kernels_moe_tile.hip -- the routed experts of one MoE block in ONE launch, Q2_K weights in the tiled layout (tile_device.h):
// the same two phases, hand-off and combine as moe_ffn_kernel (kernels_moe.hip; src/infer.cpp:853-878, 899-903), the row
// products on the matrix pipe.
//   phase A  units of 64 rows of one slot's w1 / w3 pair: 8 strips x (dim / 1024) four-block items, dealt to the 16 waves as
//            contiguous ranges; a barrier; waves 0..3 add the partials of one strip pair each (the association of
//            tile_device.h), apply the GLU and publish 16 rows of h_k write-through; then the arrival / Q8_K hand-over of
//            moe_ffn_kernel, unchanged.
//   phase B  the workgroup owns whole 16-row tiles of x for ALL slots: the steps (slot, tile, block) - one item each, W2's rows
//            are <= 8 blocks long - are dealt to the waves as contiguous ranges and the FIRST EIGHT steps of every wave (all of
//            them at DeepSeek-V3's shapes) are REQUESTED BEFORE the hand-off wait: they depend on the routing only.  After the
//            wait: the hidden vectors into block records, the multiplies (matrix pipe: ~9 steps x ~42 VALU per wave where the
//            dot4 form needed 5 us of pure arithmetic), one wave per (slot, tile) adds the partials, x += w_k o_k in k order,
//            then the shared expert - for the workgroup's own rows: no cross-workgroup combine.
// Results are BIT-identical to the two-launch form on tiled tensors (gemv_tile_kernel GLU + per-slot W2 with the last-arriver
// combine): both add a row's item partials in the one order tile_device.h defines.
#include "dsk_internal.h"
#include "tile_device.h"

#define MOE_T_PRE 8   // column steps of phase B a wave requests before the hand-off and HOLDS IN REGISTERS across it
#define MOE_T_PARK 2  // ... plus this many more, requested first and PARKED IN LDS once they have arrived: DeepSeek-V3 has
                      // 9 steps per wave (two tiles x 9 slots x 8 blocks = 144 per workgroup), and a ninth step requested behind the
                      // hand-off paid a whole memory round trip there; nine steps in registers spill (128 VGPRs at 16 waves)
#define MOE_PARK_B 1536  // bytes of a parked step: 16 + 4 + 4 per lane
// (Measured and rejected: the parked steps requested at kernel ENTRY as LDS-DMA (global_load_lds_dwordx4 / _dword straight into the
// park layout, no register in between), so that 2 of a wave's 9-10 W2 steps stream before phase A instead of behind it.  A CU's
// loads return in order: the staging of x (an L2 hit, 1.2 us) then waits behind the 43 KB of HBM reads - staged x 1.2 -> 3.0 us,
// phase A done 15.3 -> 17.6, exit 31.9 -> 33.0, the launch 34.0 -> 35.2 us.)

// LEAN: the instantiation for the shapes the model path runs at DeepSeek-V3 width - rows of x longer than 8 blocks (4-block items),
// Q8_K hand-over, every wave's phase-B steps inside its registers + park slots, no parity tap - without the code of the other cases
// (the launch is sensitive to its own size: EXPERIMENTS 5.4).  The host picks it (moe_tile_lean).
template <int LEAN>
__global__ __launch_bounds__(1024) void moe_ffn_tile_kernel(const MoeFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  constexpr int NW = 16;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int bid = blockIdx.x, G = a.grid;
  if (bid >= G) {  // tail prefetch workgroups (MoeFfnArgs::pf_wgs)
    tail_prefetch(a.pf_p, a.pf_n, tid, 1024);
    return;
  }
  uint8_t* actA = smem;
  uint8_t* actB = smem + a.lds_a;                                                        // slots x lds_b block records
  float* o_s = reinterpret_cast<float*>(smem + a.lds_a + (size_t)a.lds_b * (a.K + 1));  // [slot][rows_wg] slot outputs
  float* red = reinterpret_cast<float*>(smem + a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o);  // partials [item][64]
  const int K = a.K, slots = K + (a.shared_n > 0 ? 1 : 0);
  unsigned long long* tl = a.timeline && bid < DSK_TL_WGS ? a.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = wall_clock64();
  const TLane TL = tlane_init(lane);

  // ---- prologue: the router left Q8_K(rmsnorm(x)) behind (previous launch): copy it into block records ----
  {
    ActSrc S;
    S.act_mode = ACT_Q8; S.n = a.dim; S.a_qs = a.a_qs; S.a_d = a.a_d; S.a_bsums = a.a_bsums;
    S.a_f32 = nullptr; S.norm_w = nullptr; S.eps = 0.f; S.pre_scale = 0.f;
    stage_q8<LAY_TILE, NW>(S, actA, tid, scratch);
  }
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();

  // ---- phase A: w1/w3 GLU units of 4 strips (64 rows) ----
  {
    const int nb = a.dim >> 8, ips = tile_ips(nb);
    const int strips_e = (a.mi + 15) >> 4;  // strips per expert matrix
    for (int t = bid; t < K * a.UA; t += G) {
      const int s = t / a.UA, u = t - s * a.UA;
      const int e = a.route_e[s];
      const uint8_t* const W1 = a.w1_qs + (size_t)e * a.e13_qs;
      const uint8_t* const W3 = a.w3_qs + (size_t)e * a.e13_qs;
      const int tb = u * 4, nt = strips_e - tb < 4 ? strips_e - tb : 4;
      const int I = 2 * nt * ips;
      const int i0 = (int)((long long)I * wave / NW), i1 = (int)((long long)I * (wave + 1) / NW);
      auto strip_of = [&](int sidx, rsrc_t& W, int& soff0, const uint8_t*& act) {
        const bool m3 = sidx >= nt;
        W = make_rsrc(m3 ? W3 : W1);
        soff0 = (tb + (m3 ? sidx - nt : sidx)) * nb * TILE_B;
        act = actA;
      };
      if (LEAN || nb > 8) tile_items<4>(i0, i1, ips, nb, red, TL, lane, strip_of, [](int, int) {});
      else tile_items<1>(i0, i1, ips, nb, red, TL, lane, strip_of, [](int, int) {});
      __syncthreads();
      if (wave < nt) {  // src/infer.cpp:859-872; write-through: the consumers sit on other CUs
        const float v1 = tile_strip_value(red + (size_t)wave * ips * 64, ips, lane);
        const float v3 = tile_strip_value(red + (size_t)(nt + wave) * ips * 64, ips, lane);
        const int rr = (tb + wave) * 16 + lane;
        if (lane < 16 && rr < a.mi)
          __hip_atomic_store(a.hb + (size_t)s * a.hb_stride + rr, act_fn(v1, a.act) * v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival
      __syncthreads();
      if (!LEAN && !a.hq_qs) {
        if (tid == 0) __hip_atomic_fetch_add(a.slot_ctr + s * MOE_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (wave == NW - 1) {
        // Q8_K hand-over (kernels_moe.hip): the 4 units of a 256-block of h_k arrive on the block's counter; the LAST one
        // quantises the block (quantize_row_q8_K_ref on the published f32 values, sc1 loads), publishes codes / sums / scale
        // write-through, re-arms the block counter and only then arrives on the slot's counter (which counts blocks)
        const int blk = (u * 64) >> 8;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(a.blk_ctr + s * (a.mi >> 8) + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 3u) {
          if (lane == 0) __hip_atomic_store(a.blk_ctr + s * (a.mi >> 8) + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
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
  // ---- phase B: this workgroup's tiles [t_lo, t_hi) of x, for all slots ----
  {
    const int tiles_x = a.dim >> 4;
    const int t_lo = (int)((long long)tiles_x * bid / G), t_hi = (int)((long long)tiles_x * (bid + 1) / G);
    const int ntile = t_hi - t_lo, nrows = ntile * 16, r_lo = t_lo * 16;
    const int nbR = a.mi >> 8, nbS = a.shared_n >> 8;
    // steps: (slot k, tile, block) slot-major, the shared expert as slot K; all rows here are <= 8 blocks long (moe_ffn_plan):
    // one item per block, partial j of the list at red[j]
    const int R = K * ntile * nbR, SH = slots > K ? ntile * nbS : 0, J = R + SH;
    const int j0 = (int)((long long)J * wave / NW), j1 = (int)((long long)J * (wave + 1) / NW);
    const rsrc_t WR = make_rsrc(a.w2_qs), WS = make_rsrc(slots > K ? a.sw2_qs : a.w2_qs);
    // a wave's steps are consecutive in that list: a cursor (slot, tile, block) advanced step by step, no divisions after the first
    struct Cur { int k, t, b; bool sh; int ebase; };
    auto cur_at = [&](int j) {
      Cur c;
      c.sh = j >= R;
      if (!c.sh) {
        c.k = j / (ntile * nbR);
        const int rem = j - c.k * ntile * nbR;
        c.t = rem / nbR; c.b = rem - c.t * nbR;
        c.ebase = (int)((size_t)a.route_e[c.k < K ? c.k : 0] * a.e2_qs);  // (the stack is < 2^31 bytes: moe_ffn_plan_tile)
      } else {
        const int jj = j - R;
        c.k = K; c.t = nbS > 0 ? jj / nbS : 0; c.b = jj - c.t * nbS; c.ebase = 0;
      }
      return c;
    };
    auto cur_next = [&](Cur& c) {
      ++c.b;
      if (!c.sh) {
        if (c.b == nbR) {
          c.b = 0;
          if (++c.t == ntile) {
            c.t = 0;
            if (++c.k == K) { c.sh = true; c.ebase = 0; }
            else c.ebase = (int)((size_t)a.route_e[c.k] * a.e2_qs);
          }
        }
      } else if (c.b == nbS) { c.b = 0; ++c.t; }
    };
    auto cur_soff = [&](const Cur& c) { return c.ebase + ((t_lo + c.t) * (c.sh ? nbS : nbR) + c.b) * TILE_B; };
    auto cur_rec = [&](const Cur& c) { return actB + (size_t)c.k * a.lds_b + (size_t)c.b * TREC; };
    float xv = 0.f;
    if (tid < nrows) xv = a.x[r_lo + tid];
    // the first MOE_T_PRE + MOE_T_PARK steps of this wave are requested NOW: they stream while the slots' producers finish
    constexpr int NPRE = MOE_T_PRE + MOE_T_PARK;
    TStep S[MOE_T_PRE];
    int soffv[NPRE];
    const uint8_t* recv[NPRE];
    bool shv[NPRE];
    Cur cur = cur_at(j0 < J ? j0 : 0);
#pragma unroll
    for (int u = 0; u < NPRE; ++u)
      if (j0 + u < j1) {
        soffv[u] = cur_soff(cur); recv[u] = cur_rec(cur); shv[u] = cur.sh;
        cur_next(cur);
      }
    uint8_t* park = reinterpret_cast<uint8_t*>(red) + a.lds_red + (size_t)wave * MOE_T_PARK * MOE_PARK_B;
    {
      TStep P[MOE_T_PARK];
#pragma unroll
      for (int p = 0; p < MOE_T_PARK; ++p)
        if (j0 + MOE_T_PRE + p < j1) tstep_load(P[p], shv[MOE_T_PRE + p] ? WS : WR, TL, soffv[MOE_T_PRE + p]);
#pragma unroll
      for (int u = 0; u < MOE_T_PRE; ++u)
        if (j0 + u < j1) tstep_load(S[u], shv[u] ? WS : WR, TL, soffv[u]);
#pragma unroll
      for (int p = 0; p < MOE_T_PARK; ++p)
        if (j0 + MOE_T_PRE + p < j1) {
          uint8_t* q = park + p * MOE_PARK_B;
          *reinterpret_cast<u32x4*>(q + lane * 16) = P[p].w;
          *reinterpret_cast<u32*>(q + 1024 + lane * 4) = P[p].scw;
          *reinterpret_cast<u32*>(q + 1280 + lane * 4) = P[p].dm;
        }
    }
    // the shared expert's f32 hidden vector is ready since the router launch: quantise it in the shadow of the requests
    auto stage_shared = [&]() {
      const rsrc_t hr = make_rsrc(a.hb);
      for (int b = wave; b < nbS; b += NW) {
        const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(hr, (K * a.hb_stride + b * 256 + lane * 4) * 4, 0, 16);
        const u32 w0 = hv.x, w1 = hv.y, w2 = hv.z, w3 = hv.w;
        const float v[4] = {u2f(w0), u2f(w1), u2f(w2), u2f(w3)};
        q8k_block_lds<LAY_TILE>(v, lane, actB + (size_t)K * a.lds_b + (size_t)b * TREC);
      }
    };
    if ((LEAN || a.hq_qs) && slots > K) stage_shared();
    // wait until every phase-A unit of every slot has published (lane k of wave 0 watches slot k; bounded)
    const unsigned slot_target = (LEAN || a.hq_qs) ? (unsigned)(a.mi >> 8) : (unsigned)a.UA;
    if (wave == 0) {
      unsigned spins = 0;
      // (an earlier launch of this token already gave up - a DEVICE word next to the counters says so; the host-visible word lives
      // in pinned host memory and must not be read here: 256 reads over PCIe cost the launch 15 us - : the host will re-run the token,
      // do not spin the limit out again in every layer)
      if (__hip_atomic_load(a.slot_ctr + MOE_GAVE_UP_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = (unsigned)a.spin_limit > 64u ? (unsigned)a.spin_limit - 64u : 0u;
      for (;;) {
        const bool ok = lane >= K || __hip_atomic_load(a.slot_ctr + lane * MOE_CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slot_target;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (unsigned)a.spin_limit) {
          if (lane == 0) { *a.err = 1u; __hip_atomic_store(a.slot_ctr + MOE_GAVE_UP_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
      }
      if (a.spin_limit < 0 && bid == 0 && lane == 0) *a.err = 1u;  // fault injection (option "moe_spin_limit" < 0)
    }
    __syncthreads();
    if (tl && tid == 0) tl[3] = wall_clock64();
    if (LEAN || a.hq_qs) {  // routed slots: copies of the Q8_K blocks their producers left (sc1 loads: written during THIS launch)
      const rsrc_t qr = make_rsrc(a.hq_qs), br = make_rsrc(a.hq_bsums), dr = make_rsrc(a.hq_d);
      const int runs_per_slot = a.mi >> 4, nruns = K * runs_per_slot;
      for (int i = tid; i < nruns; i += NW * 64) {
        const int s = i / runs_per_slot, r = i - s * runs_per_slot, b = r >> 4, j = r & 15;
        const int e16 = s * (a.hb_stride >> 4) + r;
        const u32x4 codes = __builtin_amdgcn_raw_buffer_load_b128(qr, e16 * 16, 0, 16);
        const int bs = (int)(short)__builtin_amdgcn_raw_buffer_load_b16(br, e16 * 2, 0, 16);
        uint8_t* rec = actB + (size_t)s * a.lds_b + (size_t)b * TREC;
        *reinterpret_cast<u32x4*>(rec + j * 16) = codes;
        rec[TREC_BS + 8 * (j >> 2) + (j & 3)] = (uint8_t)(bs >> 8);
        rec[TREC_BS + 4 + 8 * (j >> 2) + (j & 3)] = (uint8_t)(bs & 0xff);
      }
      for (int i = tid; i < K * nbR; i += NW * 64) {
        const int s = i / nbR, b = i - s * nbR;
        uint8_t* rec = actB + (size_t)s * a.lds_b + (size_t)b * TREC;
        *reinterpret_cast<u32x4*>(rec + TREC_ZERO) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32*>(rec + TREC_D) = __builtin_amdgcn_raw_buffer_load_b32(dr, (s * (a.hb_stride >> 8) + b) * 4, 0, 16);
      }
    } else {  // f32 hidden vectors: quantize_row_q8_K_ref per block, here
      const rsrc_t hr = make_rsrc(a.hb);
      const int nblk = K * nbR + (slots > K ? nbS : 0);
      for (int bb = wave; bb < nblk; bb += NW) {
        const int s = bb < K * nbR ? bb / nbR : K, b = bb < K * nbR ? bb - s * nbR : bb - K * nbR;
        const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(hr, (s * a.hb_stride + b * 256 + lane * 4) * 4, 0, 16);
        const u32 w0 = hv.x, w1 = hv.y, w2 = hv.z, w3 = hv.w;
        const float v[4] = {u2f(w0), u2f(w1), u2f(w2), u2f(w3)};
        q8k_block_lds<LAY_TILE>(v, lane, actB + (size_t)s * a.lds_b + (size_t)b * TREC);
      }
    }
    __syncthreads();
    if (tl && tid == 0) tl[4] = wall_clock64();
    if (!LEAN && a.tap_qs && bid == 0)  // parity tap: what the slots staged
      for (int s = 0; s < slots; ++s)
        dump_staged_q8<LAY_TILE>(actB + (size_t)s * a.lds_b, s < K ? a.mi : a.shared_n, a.tap_qs + (size_t)s * a.tap_stride,
                                 a.tap_d + (size_t)s * (a.tap_stride >> 8), tid, 1024);
    // the multiplies: the steps held in registers (a full set runs straight-line: the LDS reads, matrix instructions and scale
    // arithmetic of neighbouring steps interleave), the parked ones, then (longer ranges: other shapes) the rest in groups
    if (j0 + MOE_T_PRE <= j1) {
#pragma unroll
      for (int u = 0; u < MOE_T_PRE; ++u) {
        float accd = 0.f, accm = 0.f;
        tstep_mac(S[u], recv[u], TL, accd, accm);
        red[(size_t)(j0 + u) * 64 + lane] = titem_value(accd, accm, TL);
      }
    } else {
#pragma unroll
      for (int u = 0; u < MOE_T_PRE; ++u)
        if (j0 + u < j1) {
          float accd = 0.f, accm = 0.f;
          tstep_mac(S[u], recv[u], TL, accd, accm);
          red[(size_t)(j0 + u) * 64 + lane] = titem_value(accd, accm, TL);
        }
    }
#pragma unroll
    for (int p = 0; p < MOE_T_PARK; ++p)
      if (j0 + MOE_T_PRE + p < j1) {
        const uint8_t* q = park + p * MOE_PARK_B;
        TStep P;
        P.w = *reinterpret_cast<const u32x4*>(q + lane * 16);
        P.scw = *reinterpret_cast<const u32*>(q + 1024 + lane * 4);
        P.dm = *reinterpret_cast<const u32*>(q + 1280 + lane * 4);
        float accd = 0.f, accm = 0.f;
        tstep_mac(P, recv[MOE_T_PRE + p], TL, accd, accm);
        red[(size_t)(j0 + MOE_T_PRE + p) * 64 + lane] = titem_value(accd, accm, TL);
      }
    if (!LEAN) for (int jb = j0 + NPRE; jb < j1; jb += MOE_T_PRE) {
#pragma unroll
      for (int u = 0; u < MOE_T_PRE; ++u)
        if (jb + u < j1) {
          tstep_load(S[u], cur.sh ? WS : WR, TL, cur_soff(cur));
          recv[u] = cur_rec(cur);
          cur_next(cur);
        }
#pragma unroll
      for (int u = 0; u < MOE_T_PRE; ++u)
        if (jb + u < j1) {
          float accd = 0.f, accm = 0.f;
          tstep_mac(S[u], recv[u], TL, accd, accm);
          red[(size_t)(jb + u) * 64 + lane] = titem_value(accd, accm, TL);
        }
    }
    __syncthreads();
    // one wave per (slot, tile): the rows' values (association of tile_device.h: one item per block)
    for (int q = wave; q < slots * ntile; q += NW) {
      const int s = q / ntile, tl_ = q - s * ntile;
      const int nbq = s < K ? nbR : nbS;
      const float* rs = red + (size_t)(s < K ? (s * ntile + tl_) * nbR : R + tl_ * nbS) * 64;
      const float v = tile_strip_value(rs, nbq, lane);
      if (lane < 16) {
        o_s[s * a.rows_wg + tl_ * 16 + lane] = v;
        a.eout[(size_t)s * a.dim + r_lo + tl_ * 16 + lane] = v;
      }
    }
    __syncthreads();
    if (tl && tid == 0) tl[5] = wall_clock64();
    if (tid < nrows) {  // x += w_k * o_k in k order (src/infer.cpp:874-877), then the shared expert (:900-903)
      for (int k = 0; k < K; ++k) xv = fmaf(o_s[k * a.rows_wg + tid], a.route_w[k], xv);
      if (slots > K) xv += o_s[K * a.rows_wg + tid];
      a.x[r_lo + tid] = xv;
    }
    if (tl && tid == 0) tl[6] = wall_clock64();
  }
}

// ---- host side ------------------------------------------------------------------------------------------
static size_t moe_tile_lds(const MoeFfnArgs& a) {
  return (size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o + (size_t)a.lds_red + (size_t)16 * MOE_T_PARK * MOE_PARK_B;
}

int moe_ffn_plan_tile(MoeFfnArgs& a, int n_cus) {
  if (a.quant != DSK_QUANT_Q2_K) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn (tiles): Q2_K only");
  if (a.dim % 256 || a.mi % 256 || a.shared_n % 256) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: lengths must be multiples of 256");
  const int nbA = a.dim >> 8, nbR = a.mi >> 8, nbS = a.shared_n >> 8;
  if (nbR > 8 || nbS > 8) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn (tiles): hidden vectors of more than 2048 values");
  if ((double)a.n_experts * (double)a.e2_qs >= 2147483648.0) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: the W2 stack exceeds the 31-bit offset of a buffer load");
  a.UA = (a.mi + 63) / 64;
  int grid = a.K * a.UA;
  if (grid > n_cus) grid = n_cus;
  if (grid > a.dim / 16) grid = a.dim / 16;
  a.grid = grid;
  if (a.spin_limit == 0) a.spin_limit = 1 << 20;
  const int tiles_x = a.dim / 16;
  const int ntile_max = (tiles_x + grid - 1) / grid;
  a.rows_wg = ntile_max * 16;
  if (a.rows_wg > 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: %d rows per workgroup", a.rows_wg);
  a.lds_a = (int)(((size_t)nbA * TREC + 15) & ~(size_t)15);
  a.lds_b = (int)(((size_t)(nbR > nbS ? nbR : nbS) * TREC + 15) & ~(size_t)15);
  a.lds_o = (a.K + 1) * a.rows_wg * 4;
  const int itemsA = 8 * tile_ips(nbA), itemsB = ntile_max * (a.K * nbR + nbS);
  a.lds_red = (itemsA > itemsB ? itemsA : itemsB) * 256;
  const size_t lds = moe_tile_lds(a);
  if (lds > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: activations do not fit LDS");
  int per_cu = 0;
  auto k = moe_ffn_tile_kernel<0>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 1024, lds);
  if (e != hipSuccess || per_cu < 1) DSK_FAIL(DSK_ERR_UNSUPPORTED, "moe_ffn: a workgroup is not resident on one CU (occupancy query: %d)", per_cu);
  return DSK_OK;
}

// the lean instantiation applies when none of the cases it leaves out can occur
static bool moe_tile_lean(const MoeFfnArgs& a) {
  const int nbR = a.mi >> 8, nbS = a.shared_n >> 8, slots_sh = a.shared_n > 0 ? 1 : 0;
  const int ntile_max = a.rows_wg / 16;
  const int J = ntile_max * (a.K * nbR + slots_sh * nbS);   // the largest step list of a workgroup
  return (a.dim >> 8) > 8 && a.hq_qs != nullptr && a.tap_qs == nullptr && (J + 15) / 16 <= MOE_T_PRE + MOE_T_PARK;
}
int launch_moe_ffn_tile(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (a.pipe && moe_pipe_applies(a)) return launch_moe_ffn_pipe(st, a, ev_start, ev_stop);
  const size_t lds = moe_tile_lds(a);
  auto k = moe_tile_lean(a) ? moe_ffn_tile_kernel<1> : moe_ffn_tile_kernel<0>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (ev_start && ev_stop) hipExtLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), (uint32_t)lds, st, ev_start, ev_stop, 0u, a);
  else hipLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), lds, st, a);
  return DSK_OK;
}
