// kernels_moe_pipe.hip -- the routed experts of one MoE block in ONE launch, PIPELINED BY SLOT HALVES (round 6).
//
// Same arithmetic, same bits as moe_ffn_tile_kernel (kernels_moe_tile.hip; src/infer.cpp:853-878, 899-903: w1 / w3 GLU per selected
// expert -> Q8_K -> W2 -> x += w_k o_k in k order, then the shared expert) - every row value is the association of tile_device.h,
// which does not depend on who computes which item -, another schedule.  The one-phase-after-the-other launch runs the chip in
// lock step: all 256 workgroups finish phase A together (~16 us), then all wait ~9 us for the hand-over of the hidden vectors while
// their W2 tiles stream, then all multiply for ~5 us with HBM idle (profiles/r05_timeline_moe.txt).  Here
//   * phase A is dealt as TWO units of 32 rows per workgroup: first a unit of a slot of the FIRST half (slots [0, K/2)), then one of the
//     second half.  The first half's hidden vectors are complete, quantised and handed over while the second half still streams;
//   * waves have ROLES: 14 PRODUCER waves stream and multiply (two 4-block items each per unit: an exact deal, no idle wave slots;
//     no workgroup barrier anywhere in phase A - a producer announces its partials on an LDS counter and goes on), 2 SERVICE waves do
//     everything that is a chain of round trips, and they do it through the SCALAR memory path - s_store_dword glc + s_dcache_wb,
//     s_atomic_add, s_load_dword glc - which is outside the CU's in-order vector-memory queue (tools/scalar_store_probe.hip,
//     profiles/r06_scalar_store_probe.txt: a publish costs 1.1 us while the CU streams where the vector form costs 10.6; the data
//     and the atomics are coherent across XCDs): each adds one strip's partials, applies the GLU and runs the Q8_K hand-over of its
//     16 values in two steps - publish the strip's (|max|, signed max) candidate and arrive on the 256-block's counter; when the block's
//     16 strips have arrived read the 16 candidates, pick the block's max the way quantize_row_q8_K_ref does (src/quant.cpp:622-629:
//     the first element with the largest magnitude), quantise ITS OWN 16 values, publish the 16 codes (+ the block scale) and arrive
//     on the slot's counter.  Nobody gathers 256 floats; the consumers recompute the sub-block sums from the codes.  The service waves
//     then poll the slot counters, copy the handed-over codes into LDS block records and raise an LDS flag;
//   * phase B is two stages per producer: the W2 steps of (shared expert + first half) and of the second half are requested together
//     the moment the producer's phase A is through; stage 1 multiplies as soon as its tiles are there (its hidden vectors were copied
//     during phase A), stage 2's tiles stream under those multiplies and its hidden vectors arrive meanwhile.
// What stays exposed behind phase A: stage 1's stream, stage 1's multiplies (~5/9 of them), stage 2's multiplies, the combine.
#include "dsk_internal.h"
#include "tile_device.h"

#ifdef MP_TL_PROD  // diagnostics build: producer waves 2 and 15 through phase A
#define MP_PST(w, i) do { if (tl && wave == (w) && lane == 0) tl[i] = wall_clock64(); } while (0)
#else
#define MP_PST(w, i) do { } while (0)
#endif
#if defined(MP_TL_SVC) || defined(MP_TL_PROD)
#define MP_TLN(i) do { } while (0)
#else
#define MP_TLN(i) tl[i] = wall_clock64()
#endif
// the workgroup barrier as the meeting point of a unit: LDS traffic ordered, requests in flight left alone
#define MP_UNIT_MEET() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MP_UNIT_PENDING() (wave == 0 && r_done < u_done)
#ifndef MP_SLEEP
#define MP_SLEEP 1
#endif
#ifndef MP_NAP0
#define MP_NAP0 16  // trips of the service loop (~70 ns each) between wave 0's polls of the first half's counter
#endif
#ifndef MP_NAP1
#define MP_NAP1 3   // ... of the second half's
#endif
#define MP_NP 14   // producer waves (2 .. 15); waves 0 and 1 serve
#define MP_NS1 6   // W2 steps of stage 1 / stage 2 a producer holds in registers (DeepSeek-V3: 80 / 14 and 64 / 14)
#define MP_NS2 5
#define MP_ROWS 32 // rows of a phase-A unit (two strips)
#define MP_HB 3    // hidden-vector blocks a producer quantises per stage (DeepSeek-V3: 40 / 14 and 32 / 14)

// LDS words the waves of a workgroup meet on (no s_barrier inside the pipeline)
enum { MPW_UNIT0 = 0, MPW_UNIT1 = 1, MPW_RDY0 = 2, MPW_RDY1 = 3, MPW_HID0 = 4, MPW_HID1 = 5, MPW_N = 8 };

DEV unsigned mp_sload_glc(const unsigned* p) {  // a scalar load that sees other XCDs' agent-scope stores of this launch
  unsigned r;
  asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
  return r;
}
// are the slot counters of slots [s_lo, s_lo + n), n <= 4, at their target?  Four scalar loads in flight, one wait (a counter sits
// MOE_CTR_STRIDE words = 256 bytes from the next; loads past the n-th read neighbouring counters and are ignored)
DEV bool mp_slots_ready(const unsigned* ctr0, int n, unsigned target) {
  static_assert(MOE_CTR_STRIDE == 64, "the immediates below are 256-byte strides");
  unsigned r0, r1, r2, r3;
  asm volatile(
      "s_load_dword %0, %4, 0x0 glc\n\t"
      "s_load_dword %1, %4, 0x100 glc\n\t"
      "s_load_dword %2, %4, 0x200 glc\n\t"
      "s_load_dword %3, %4, 0x300 glc\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3)
      : "s"(ctr0)
      : "memory");
  return r0 >= target && (n < 2 || r1 >= target) && (n < 3 || r2 >= target) && (n < 4 || r3 >= target);
}
typedef u32 u32x16_t __attribute__((ext_vector_type(16)));
DEV u32x16_t mp_sload16_glc(const void* p) {
  u32x16_t r;
  asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
  return r;
}
// scalar stores (glc: through to where other XCDs read), the write-back of the scalar data cache, scalar atomic add: the wave waits for
// each (the data SGPRs of an SMEM store must outlive it)
DEV void mp_sstore2(void* p, u32 a, u32 b) {
  const u32x2 v = {a, b};
  asm volatile("s_store_dwordx2 %0, %1, 0x0 glc" : : "s"(v), "s"(p) : "memory");
}
typedef u32 u32x8_t __attribute__((ext_vector_type(8)));
DEV u32x8_t mp_sload8(const void* p) {  // plain: data of an EARLIER launch
  u32x8_t r;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
  return r;
}
DEV u32x8_t mp_sload8_glc(const void* p) {
  u32x8_t r;
  asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
  return r;
}
DEV void mp_sstore4x4(void* p, u32x4 a, u32x4 b, u32x4 c, u32x4 d) {  // 64 contiguous bytes
  asm volatile("s_store_dwordx4 %0, %4, 0x0 glc\n\ts_store_dwordx4 %1, %4, 0x10 glc\n\ts_store_dwordx4 %2, %4, 0x20 glc\n\ts_store_dwordx4 %3, %4, 0x30 glc"
               : : "s"(a), "s"(b), "s"(c), "s"(d), "s"(p) : "memory");
}
DEV void mp_sstore4(void* p, u32x4 v) { asm volatile("s_store_dwordx4 %0, %1, 0x0 glc" : : "s"(v), "s"(p) : "memory"); }
DEV void mp_sstore1(void* p, u32 v) { asm volatile("s_store_dword %0, %1, 0x0 glc" : : "s"(v), "s"(p) : "memory"); }
DEV void mp_swb() { asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
DEV void mp_satomic_inc(unsigned* p) {
  const u32 one = 1u;
  asm volatile("s_atomic_add %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : : "s"(one), "s"(p) : "memory");
}
DEV unsigned mp_satomic_inc_ret(unsigned* p) {
  u32 v = 1u;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
  return v;
}
// LDS words the waves meet on.  NO release / acquire orderings here: hipcc implements them with s_waitcnt vmcnt(0), i.e. a producer
// that announces its partials would first wait for every tile it has already requested for the NEXT item.  What has to be ordered is
// LDS traffic only, and a wave's LDS operations execute in order: the s_waitcnt lgkmcnt(0) in front of the announcement is all it takes.
DEV void mp_arrive(unsigned* w) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV void mp_signal(unsigned* w, unsigned v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV unsigned mp_peek(unsigned* w) {
  const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
DEV void mp_wait_ge(unsigned* w, unsigned v) {
  while (mp_peek(w) < v) __builtin_amdgcn_s_sleep(MP_SLEEP);
}

// ------------------------------------------------------------------------------------------------------------------------
// moe_ffn_stage_kernel (option "moe_pipe" = 2): the one-phase launch (moe_ffn_tile_kernel<1>: same phase A, same Q8_K hand-over chain,
// same copies, same multiplies, same bits) with its SECOND HALF re-ordered around what the probes of this round say:
//   * the hand-over chain runs FIRST, before anybody on this CU requests W2 (2.0 us instead of 5.1 behind the requests, EXPERIMENTS 4.4),
//     while waves 0..7 quantise the shared expert's hidden vector;
//   * the W2 steps are requested in TWO stages - (shared expert + slots [0, K/2)), then slots [K/2, K) - with the hand-off wait and the
//     hidden vectors' copy requests BETWEEN them: the copies (16 KB) queue behind stage 1's tiles only, not behind all of W2;
//   * the wait polls through the scalar path (s_load_dword glc: not in the CU's in-order queue);
//   * stage 1 multiplies while stage 2's tiles stream.
// One workgroup barrier sits between a wave's requests and its multiplies (where the copies meet), none behind stage 2.
// ------------------------------------------------------------------------------------------------------------------------
#define MS_NS1 5  // W2 steps of stage 1 / stage 2 a wave holds in registers (DeepSeek-V3: 80 / 16 and 64 / 16)
#define MS_NS2 4
#ifndef MS_POLL_NAP
#define MS_POLL_NAP 12
#endif
__global__ __launch_bounds__(1024) void moe_ffn_stage_kernel(const MoeFfnArgs a) {
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
  const int K = a.K, KH = K >> 1, slots = K + (a.shared_n > 0 ? 1 : 0);
  unsigned long long* tl = a.timeline && bid < DSK_TL_WGS ? a.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = wall_clock64();
  const TLane TL = tlane_init(lane);
  {  // the router left Q8_K(rmsnorm(x)) behind (previous launch): copy it into block records
    ActSrc S;
    S.act_mode = ACT_Q8; S.n = a.dim; S.a_qs = a.a_qs; S.a_d = a.a_d; S.a_bsums = a.a_bsums;
    S.a_f32 = nullptr; S.norm_w = nullptr; S.eps = 0.f; S.pre_scale = 0.f;
    stage_q8<LAY_TILE, NW>(S, actA, tid, scratch);
  }
  // the routing into SGPRs once, through the scalar path (every a.route_e[k] below would be a vector load with a full wait behind
  // whatever this CU has requested by then)
  const u32x8_t RE = mp_sload8(a.route_e);
  auto route_of = [&](int k) {
    u32 r = RE[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) r = k == j ? RE[j] : r;
    return (int)r;
  };
  const int tiles_x = a.dim >> 4;
  const int t_lo = (int)((long long)tiles_x * bid / G), t_hi = (int)((long long)tiles_x * (bid + 1) / G);
  const int ntile = t_hi - t_lo, nrows = ntile * 16, r_lo = t_lo * 16;
  const int nbR = a.mi >> 8, nbS = a.shared_n >> 8;
  float xv = 0.f;
  if (tid < nrows) xv = a.x[r_lo + tid];
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();

  // ---- phase A: one w1/w3 GLU unit of 4 strips (64 rows), as in moe_ffn_tile_kernel ----
  {
    const int nb = a.dim >> 8, ips = tile_ips(nb);
    const int s = bid / a.UA, u = bid - s * a.UA;
    const int e = route_of(s);
    const uint8_t* const W1 = a.w1_qs + (size_t)e * a.e13_qs;
    const uint8_t* const W3 = a.w3_qs + (size_t)e * a.e13_qs;
    const int tb = u * 4, nt = 4;
    const int I = 2 * nt * ips;
    const int i0 = (int)((long long)I * wave / NW), i1 = (int)((long long)I * (wave + 1) / NW);
    auto strip_of = [&](int sidx, rsrc_t& W, int& soff0, const uint8_t*& act) {
      const bool m3 = sidx >= nt;
      W = make_rsrc(m3 ? W3 : W1);
      soff0 = (tb + (m3 ? sidx - nt : sidx)) * nb * TILE_B;
      act = actA;
    };
    tile_items<4>(i0, i1, ips, nb, red, TL, lane, strip_of, [](int, int) {});
    __syncthreads();
    if (wave < nt) {  // src/infer.cpp:859-872; write-through: the consumers sit on other CUs
      const float v1 = tile_strip_value(red + (size_t)wave * ips * 64, ips, lane);
      const float v3 = tile_strip_value(red + (size_t)(nt + wave) * ips * 64, ips, lane);
      const int rr = (tb + wave) * 16 + lane;
      if (lane < 16) __hip_atomic_store(a.hb + (size_t)s * a.hb_stride + rr, act_fn(v1, a.act) * v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival
    __syncthreads();
    if (tl && tid == 0) tl[2] = wall_clock64();
    if (wave == NW - 1) {
      // Q8_K hand-over (kernels_moe.hip), FIRST: nothing of this CU is in the memory queue in front of its round trips
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
        // the slot's counter counts blocks; whoever completes a slot counts it on ONE further word (the line of "slot K", which the
        // router launch re-arms with the others): 256 workgroups poll that word, not eight lines
        unsigned os = 0;
        if (lane == 0) os = __hip_atomic_fetch_add(a.slot_ctr + s * MOE_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0 && os == (unsigned)(a.mi >> 8) - 1u) __hip_atomic_fetch_add(a.slot_ctr + K * MOE_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (slots > K && wave < nbS) {
      // meanwhile: the shared expert's f32 hidden vector (ready since the router launch) -> slot K's block records
      const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.hb), (K * a.hb_stride + wave * 256 + lane * 4) * 4, 0, 16);
      const u32 w0 = hv.x, w1 = hv.y, w2 = hv.z, w3 = hv.w;
      const float v[4] = {u2f(w0), u2f(w1), u2f(w2), u2f(w3)};
      q8k_block_lds<LAY_TILE>(v, lane, actB + (size_t)K * a.lds_b + (size_t)wave * TREC);
    }
    __syncthreads();  // the chain is through (this CU's part of it): now the stream
  }
  if (tl && tid == 0) tl[7] = wall_clock64();

  // ---- phase B: this workgroup's tiles [t_lo, t_hi) of x, for all slots, in two stages ----
  const int R = K * ntile * nbR;  // routed partials: (slot k, tile t, block b) at red[(k ntile + t) nbR + b]; the shared expert's behind them
  const int nsh = slots > K ? ntile * nbS : 0;
  const int J1 = nsh + KH * ntile * nbR, J2 = (K - KH) * ntile * nbR;
  const int j0 = (int)((long long)J1 * wave / NW), j1 = (int)((long long)J1 * (wave + 1) / NW);
  const int k0 = (int)((long long)J2 * wave / NW), k1 = (int)((long long)J2 * (wave + 1) / NW);
  const rsrc_t WR = make_rsrc(a.w2_qs), WS = make_rsrc(slots > K ? a.sw2_qs : a.w2_qs);
  struct Cur { int k, t, b; bool sh; int ebase; };
  auto cur_routed = [&](int jj, int kbase) {
    Cur c;
    c.sh = false;
    c.k = kbase + jj / (ntile * nbR);
    const int rem = jj - (c.k - kbase) * ntile * nbR;
    c.t = rem / nbR; c.b = rem - c.t * nbR;
    c.ebase = (int)((size_t)route_of(c.k < K ? c.k : 0) * a.e2_qs);  // (the stack is < 2^31 bytes: moe_ffn_plan_tile)
    return c;
  };
  auto cur_next = [&](Cur& c) {
    ++c.b;
    if (c.sh) {
      if (c.b == nbS) {
        c.b = 0;
        if (++c.t == ntile) { c.sh = false; c.t = 0; c.k = 0; c.ebase = (int)((size_t)route_of(0) * a.e2_qs); }
      }
    } else if (c.b == nbR) {
      c.b = 0;
      if (++c.t == ntile) {
        c.t = 0;
        ++c.k;
        c.ebase = (int)((size_t)route_of(c.k < K ? c.k : 0) * a.e2_qs);
      }
    }
  };
  auto cur_soff = [&](const Cur& c) { return c.ebase + ((t_lo + c.t) * (c.sh ? nbS : nbR) + c.b) * TILE_B; };
  auto cur_rec = [&](const Cur& c) { return actB + (size_t)(c.sh ? K : c.k) * a.lds_b + (size_t)c.b * TREC; };
  auto cur_rix = [&](const Cur& c) { return c.sh ? R + c.t * nbS + c.b : (c.k * ntile + c.t) * nbR + c.b; };
  TStep S1[MS_NS1], S2[MS_NS2];
  const uint8_t *rec1[MS_NS1], *rec2[MS_NS2];
  int rix1[MS_NS1], rix2[MS_NS2];
  {  // stage 1: the shared expert's steps, then slots [0, KH)
    Cur c;
    if (j0 < nsh) { c.sh = true; c.k = K; c.t = j0 / nbS; c.b = j0 - c.t * nbS; c.ebase = 0; }
    else c = cur_routed(j0 < J1 ? j0 - nsh : 0, 0);
#pragma unroll
    for (int q = 0; q < MS_NS1; ++q)
      if (j0 + q < j1) {
        rec1[q] = cur_rec(c); rix1[q] = cur_rix(c);
        tstep_load(S1[q], c.sh ? WS : WR, TL, cur_soff(c));
        cur_next(c);
      }
  }
  // wait until every phase-A unit of every slot has been handed over: wave 0 polls the slot counters through the scalar path
  // (two batches of four loads), the others meet it at the barrier with their requests in flight
#ifdef MS_TL2  // diagnostics: wave 0: stage 1 requested -> [2]; poll passed -> [7]
  if (tl && tid == 0) tl[2] = wall_clock64();
#endif
  if (wave == 0) {
    unsigned spins = 0;
    if (mp_sload_glc(a.slot_ctr + MOE_GAVE_UP_WORD) != 0u) spins = (unsigned)a.spin_limit > 64u ? (unsigned)a.spin_limit - 64u : 0u;
    for (;;) {
      if (mp_sload_glc(a.slot_ctr + K * MOE_CTR_STRIDE) >= (unsigned)K) break;
      __builtin_amdgcn_s_sleep(MS_POLL_NAP);  // (sparse polls: a tight loop of 256 pollers on one line delays the very atomics it waits for)
      if (++spins > (unsigned)a.spin_limit) {
        if (lane == 0) { *a.err = 1u; __hip_atomic_store(a.slot_ctr + MOE_GAVE_UP_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        break;
      }
    }
    if (a.spin_limit < 0 && bid == 0 && lane == 0) *a.err = 1u;  // fault injection (option "moe_spin_limit" < 0)
#ifdef MS_TL2
    if (tl && lane == 0) tl[7] = wall_clock64();
#endif
  }
  lds_barrier();
  if (tl && tid == 0) tl[3] = wall_clock64();
  {  // the routed slots' hidden vectors: copies of the Q8_K blocks their producers left (sc1 loads: written during THIS launch)
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
  }
  {  // stage 2: slots [KH, K) - requested behind the copies
    Cur c = cur_routed(k0 < J2 ? k0 : 0, KH);
#pragma unroll
    for (int q = 0; q < MS_NS2; ++q)
      if (k0 + q < k1) {
        rec2[q] = cur_rec(c); rix2[q] = cur_rix(c);
        tstep_load(S2[q], WR, TL, cur_soff(c));
        cur_next(c);
      }
  }
  lds_barrier();  // the copies are in LDS (every wave's share); stage 2's tiles stay in flight
  if (tl && tid == 0) tl[4] = wall_clock64();
#ifndef DSK_NO_TAPS
  if (a.tap_qs && bid == 0)  // parity tap: what the slots staged
    for (int s = 0; s < slots; ++s)
      dump_staged_q8<LAY_TILE>(actB + (size_t)s * a.lds_b, s < K ? a.mi : a.shared_n, a.tap_qs + (size_t)s * a.tap_stride,
                               a.tap_d + (size_t)s * (a.tap_stride >> 8), tid, 1024);
#endif
#pragma unroll
  for (int q = 0; q < MS_NS1; ++q)
    if (j0 + q < j1) {
      float accd = 0.f, accm = 0.f;
      tstep_mac(S1[q], rec1[q], TL, accd, accm);
      red[(size_t)rix1[q] * 64 + lane] = titem_value(accd, accm, TL);
    }
#pragma unroll
  for (int q = 0; q < MS_NS2; ++q)
    if (k0 + q < k1) {
      float accd = 0.f, accm = 0.f;
      tstep_mac(S2[q], rec2[q], TL, accd, accm);
      red[(size_t)rix2[q] * 64 + lane] = titem_value(accd, accm, TL);
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

__global__ __launch_bounds__(1024) void moe_ffn_pipe_kernel(const MoeFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ unsigned sync_w[MPW_N];
  constexpr int NW = 16, NP = MP_NP;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int bid = blockIdx.x, G = a.grid;
  if (bid >= G) {  // tail prefetch workgroups (MoeFfnArgs::pf_wgs)
    tail_prefetch(a.pf_p, a.pf_n, tid, 1024);
    return;
  }
  const int K = a.K, KH = K >> 1, slots = K + (a.shared_n > 0 ? 1 : 0);
  uint8_t* actA = smem;
  uint8_t* actB = smem + a.lds_a;                                                        // slots x lds_b block records
  float* o_s = reinterpret_cast<float*>(smem + a.lds_a + (size_t)a.lds_b * (K + 1));   // [slot][rows_wg] slot outputs
  float* red = reinterpret_cast<float*>(smem + a.lds_a + (size_t)a.lds_b * (K + 1) + a.lds_o);
  unsigned long long* tl = a.timeline && bid < DSK_TL_WGS ? a.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = wall_clock64();
  if (tid < MPW_N) sync_w[tid] = 0u;
  const TLane TL = tlane_init(lane);

  // ---- prologue: the router left Q8_K(rmsnorm(x)) behind (previous launch): copy it into block records ----
  {
    ActSrc S;
    S.act_mode = ACT_Q8; S.n = a.dim; S.a_qs = a.a_qs; S.a_d = a.a_d; S.a_bsums = a.a_bsums;
    S.a_f32 = nullptr; S.norm_w = nullptr; S.eps = 0.f; S.pre_scale = 0.f;
    stage_q8<LAY_TILE, NW>(S, actA, tid, scratch);
  }
  // geometry of both phases
  const int nb = a.dim >> 8, ips = tile_ips(nb);  // rows of x: nb > 8, 4-block items (moe_pipe_applies)
  const int UH = a.mi / MP_ROWS;                  // phase-A units per slot
  const int sA = bid / UH, u0_ = bid - sA * UH;   // this workgroup's unit: rows [32 u, 32 u + 32) of slot sA, then of slot sA + KH
#define MP_SLOT(g) (sA + (g) * KH)
#define MP_UNIT(g) (u0_)
  const int I = 4 * ips;                          // items of a unit: (w1 | w3) x 2 strips x ips
  const int tiles_x = a.dim >> 4;
  const int t_lo = (int)((long long)tiles_x * bid / G), t_hi = (int)((long long)tiles_x * (bid + 1) / G);
  const int ntile = t_hi - t_lo, nrows = ntile * 16, r_lo = t_lo * 16;
  const int nbR = a.mi >> 8, nbS = a.shared_n >> 8;
  const int R = K * ntile * nbR;                  // routed partials of phase B: (slot k, tile t, block b) at red_b[(k ntile + t) nbR + b]
  float* red_b = red + (size_t)2 * I * 64;        // phase B's partials live behind the two units' (a service wave may still read those)
  // The routing (written by the router launch) into SGPRs, ONCE, through the scalar path: every a.route_e[k] in the code below would be
  // a vector load with a full wait - behind whatever tiles the CU has requested by then (it serves in order)
  const u32x8_t RE = mp_sload8(a.route_e);
  auto route_of = [&](int k) {
    u32 r = RE[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) r = k == j ? RE[j] : r;
    return (int)r;
  };
  float xv = 0.f;
  if (tid < nrows) xv = a.x[r_lo + tid];
  __syncthreads();
  if (tl && tid == 0) MP_TLN(1);

  if (wave >= 2) {
    // =========================== PRODUCERS ===========================
    const int p = wave - 2;
    {  // ---- phase A: two units, two 4-block items of each for this wave: a stream of four items, two of them requested at any time ----
      // (the deal is exact - moe_pipe_applies -: no guards, straight-line code, so that hipcc counts the outstanding loads exactly and
      // an item is multiplied as soon as ITS tiles are there while the next one streams.  Both items of the FIRST unit are requested
      // before anything else by every wave: the CU serves in order, so the first half's tiles come first for everybody.)
      TStep A[4], B[4];
      auto item_at = [&](int g, int it, rsrc_t& W, int& soff, int& rix, int& k) {
        const int s = MP_SLOT(g), u = MP_UNIT(g);
        const int e = route_of(s);
        const int i = 2 * p + it, st = i / ips;  // strip st: 0, 1 = w1's two strips, 2, 3 = w3's
        k = i - st * ips;
        W = make_rsrc((st >= 2 ? a.w3_qs : a.w1_qs) + (size_t)e * a.e13_qs);
        soff = ((u * 2 + (st & 1)) * nb + 4 * k) * TILE_B;
        rix = g * I + st * ips + k;
      };
      auto item_load = [&](TStep (&S)[4], rsrc_t W, int soff) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tstep_load(S[q], W, TL, soff + q * TILE_B);
      };
      auto item_mac = [&](const TStep (&S)[4], int k, int rix) {
        float accd = 0.f, accm = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) tstep_mac(S[q], actA + (size_t)(4 * k + q) * TREC, TL, accd, accm);
        red[(size_t)rix * 64 + lane] = titem_value(accd, accm, TL);
      };
      rsrc_t W0, W1_, W2, W3_;
      int so0, so1, so2, so3, rx0, rx1, rx2, rx3, k0_, k1_, k2_, k3_;
      item_at(0, 0, W0, so0, rx0, k0_);
      item_at(0, 1, W1_, so1, rx1, k1_);
      item_at(1, 0, W2, so2, rx2, k2_);
      item_at(1, 1, W3_, so3, rx3, k3_);
      item_load(A, W0, so0);
      item_load(B, W1_, so1);
      MP_PST(2, 1);
      item_mac(A, k0_, rx0);
      item_load(A, W2, so2);
      item_mac(B, k1_, rx1);
      MP_PST(2, 2);
      item_load(B, W3_, so3);
      MP_UNIT_MEET();  // the first unit's partials are in LDS (every producer's): the service waves go
      MP_PST(2, 3);
      item_mac(A, k2_, rx2);
      item_mac(B, k3_, rx3);
      MP_PST(2, 4);
      MP_UNIT_MEET();  // the second unit's
    }
    if (tl && wave == 2 && lane == 0) MP_TLN(2);
    {
    // ---- phase B: this workgroup's tiles [t_lo, t_hi) of x for all slots, two stages ----
    // stage 1: the shared expert's steps (tile, block), then slots [0, KH) as (slot, tile, block); stage 2: slots [KH, K)
    const int nsh = slots > K ? ntile * nbS : 0;
    const int J1 = nsh + KH * ntile * nbR, J2 = (K - KH) * ntile * nbR;
    const int j0 = J1 * p / NP, j1 = J1 * (p + 1) / NP;
    const int k0 = J2 * p / NP, k1 = J2 * (p + 1) / NP;
    const rsrc_t WR = make_rsrc(a.w2_qs), WS = make_rsrc(slots > K ? a.sw2_qs : a.w2_qs);
    struct Cur { int k, t, b; bool sh; int ebase; };
    auto cur_routed = [&](int jj, int kbase) {  // step jj of the (slot, tile, block) list that starts at slot kbase
      Cur c;
      c.sh = false;
      c.k = kbase + jj / (ntile * nbR);
      const int rem = jj - (c.k - kbase) * ntile * nbR;
      c.t = rem / nbR; c.b = rem - c.t * nbR;
      c.ebase = (int)((size_t)route_of(c.k < K ? c.k : 0) * a.e2_qs);  // (the stack is < 2^31 bytes: moe_ffn_plan_tile)
      return c;
    };
    auto cur_next = [&](Cur& c) {
      ++c.b;
      if (c.sh) {
        if (c.b == nbS) {
          c.b = 0;
          if (++c.t == ntile) { c.sh = false; c.t = 0; c.k = 0; c.ebase = (int)((size_t)route_of(0) * a.e2_qs); }
        }
      } else if (c.b == nbR) {
        c.b = 0;
        if (++c.t == ntile) {
          c.t = 0;
          ++c.k;
          c.ebase = (int)((size_t)route_of(c.k < K ? c.k : 0) * a.e2_qs);
        }
      }
    };
    auto cur_soff = [&](const Cur& c) { return c.ebase + ((t_lo + c.t) * (c.sh ? nbS : nbR) + c.b) * TILE_B; };
    auto cur_rec = [&](const Cur& c) { return actB + (size_t)(c.sh ? K : c.k) * a.lds_b + (size_t)c.b * TREC; };
    auto cur_rix = [&](const Cur& c) { return c.sh ? R + c.t * nbS + c.b : (c.k * ntile + c.t) * nbR + c.b; };
    // The hidden vectors become Q8_K block records HERE, spread over the producers (quantize_row_q8_K_ref, src/quant.cpp:616-653, per
    // 256-block): blocks p, p + 14, p + 28 of a stage's list - stage 1: the shared expert's blocks (f32 since the router launch), then
    // slots [0, KH); stage 2: slots [KH, K) - from the f32 values the service waves published (sc1 loads: written during THIS launch).
    // A wave's requests go out in the order it needs the data: hidden blocks of stage 1, W2 tiles of stage 1, hidden blocks of
    // stage 2, W2 tiles of stage 2 - the CU serves in order.
    const rsrc_t hr = make_rsrc(a.hb);
    auto hid_load = [&](int stage, u32x4 (&hv)[MP_HB]) {
      const int nblk_st = stage == 0 ? (slots > K ? nbS : 0) + KH * nbR : (K - KH) * nbR;
#pragma unroll
      for (int q = 0; q < MP_HB; ++q) {
        const int bb = p + NP * q;
        if (bb < nblk_st) {
          int s, b;
          if (stage == 0 && slots > K && bb < nbS) { s = K; b = bb; }
          else { const int r = bb - (stage == 0 && slots > K ? nbS : 0); s = (stage ? KH : 0) + r / nbR; b = r - (s - (stage ? KH : 0)) * nbR; }
          hv[q] = __builtin_amdgcn_raw_buffer_load_b128(hr, (s * a.hb_stride + b * 256 + lane * 4) * 4, 0, 16);
        }
      }
    };
    auto hid_quant = [&](int stage, const u32x4 (&hv)[MP_HB]) {
      const int nblk_st = stage == 0 ? (slots > K ? nbS : 0) + KH * nbR : (K - KH) * nbR;
#pragma unroll
      for (int q = 0; q < MP_HB; ++q) {
        const int bb = p + NP * q;
        if (bb < nblk_st) {
          int s, b;
          if (stage == 0 && slots > K && bb < nbS) { s = K; b = bb; }
          else { const int r = bb - (stage == 0 && slots > K ? nbS : 0); s = (stage ? KH : 0) + r / nbR; b = r - (s - (stage ? KH : 0)) * nbR; }
          const u32 w0 = hv[q].x, w1 = hv[q].y, w2 = hv[q].z, w3 = hv[q].w;
          const float v[4] = {u2f(w0), u2f(w1), u2f(w2), u2f(w3)};
          q8k_block_lds<LAY_TILE>(v, lane, actB + (size_t)s * a.lds_b + (size_t)b * TREC);
        }
      }
    };
    TStep S1[MP_NS1], S2[MP_NS2];
    const uint8_t *rec1[MP_NS1], *rec2[MP_NS2];
    int rix1[MP_NS1], rix2[MP_NS2];
    u32x4 hv1[MP_HB], hv2[MP_HB];
    mp_wait_ge(&sync_w[MPW_RDY0], 1u);  // every strip of the first half's slots is published (long ago, normally)
    hid_load(0, hv1);
    {
      Cur c;
      if (j0 < nsh) { c.sh = true; c.k = K; c.t = j0 / nbS; c.b = j0 - c.t * nbS; c.ebase = 0; }
      else c = cur_routed(j0 < J1 ? j0 - nsh : 0, 0);
#pragma unroll
      for (int q = 0; q < MP_NS1; ++q)
        if (j0 + q < j1) {
          rec1[q] = cur_rec(c); rix1[q] = cur_rix(c);
          tstep_load(S1[q], c.sh ? WS : WR, TL, cur_soff(c));
          cur_next(c);
        }
    }
    hid_quant(0, hv1);
    if (lane == 0) mp_arrive(&sync_w[MPW_HID0]);
    mp_wait_ge(&sync_w[MPW_RDY1], 1u);  // ... of the second half's (the service waves' scalar polls)
    if (tl && wave == 2 && lane == 0) MP_TLN(3);
    hid_load(1, hv2);
    {
      Cur c = cur_routed(k0 < J2 ? k0 : 0, KH);
#pragma unroll
      for (int q = 0; q < MP_NS2; ++q)
        if (k0 + q < k1) {
          rec2[q] = cur_rec(c); rix2[q] = cur_rix(c);
          tstep_load(S2[q], WR, TL, cur_soff(c));
          cur_next(c);
        }
    }
    // (the second half's blocks are quantised BEFORE stage 1 multiplies: their values arrive right behind stage 1's tiles, and the
    // registers that hold them are free again when the multiplies need theirs)
    hid_quant(1, hv2);
    if (lane == 0) mp_arrive(&sync_w[MPW_HID1]);
    mp_wait_ge(&sync_w[MPW_HID0], (unsigned)NP);  // stage 1's hidden vectors are block records (every producer's share)
#pragma unroll
    for (int q = 0; q < MP_NS1; ++q)
      if (j0 + q < j1) {
        float accd = 0.f, accm = 0.f;
        tstep_mac(S1[q], rec1[q], TL, accd, accm);
        red_b[(size_t)rix1[q] * 64 + lane] = titem_value(accd, accm, TL);
      }
    mp_wait_ge(&sync_w[MPW_HID1], (unsigned)NP);
    if (tl && wave == 2 && lane == 0) MP_TLN(4);
#pragma unroll
    for (int q = 0; q < MP_NS2; ++q)
      if (k0 + q < k1) {
        float accd = 0.f, accm = 0.f;
        tstep_mac(S2[q], rec2[q], TL, accd, accm);
        red_b[(size_t)rix2[q] * 64 + lane] = titem_value(accd, accm, TL);
      }
    }
  } else {
    // =========================== SERVICE WAVES ===========================
    const int nblk = a.mi >> 8;
    unsigned spins = 0;
    bool gave_up = false;
    MP_PST(0, 7);
    {
    // (an earlier launch of this token already gave up - a DEVICE word next to the counters says so: do not spin the limit out again)
    if (mp_sload_glc(a.slot_ctr + MOE_GAVE_UP_WORD) != 0u) spins = (unsigned)a.spin_limit > 64u ? (unsigned)a.spin_limit - 64u : 0u;
    // Arrivals are counted on two levels, so that no line is hammered: a strip arrives on its 256-block's counter (16 strips; a line of
    // its own per counter), the block's LAST strip arrives on the half's counter (the first two slot-counter lines, which the router
    // launch re-arms), and every workgroup's wave 0 polls that ONE word - sparsely: 256 pollers in a tight loop on one line keep its
    // memory channel busy, and every CU's in-order stream has requests waiting on that channel (measured: the first unit then takes
    // 16 us instead of 8)
    const unsigned half_target = (unsigned)(KH * nblk);
#ifdef MP_TL_SVC  // diagnostics build: stamps 1..7 = service wave 0's duties, step by step
#define MP_STAMP(i) do { if (tl && wave == 0 && lane == 0) tl[i] = wall_clock64(); } while (0)
#else
#define MP_STAMP(i) do { } while (0)
#endif
    // A unit's epilogue for this wave's strip (src/infer.cpp:859-872): add the partials, GLU, publish the 16 values through the scalar
    // path (4 x s_store_dwordx4 glc + write-back), then arrive on the strip's 256-block counter.  Wave 0 also watches the slot halves
    // and tells the producers (LDS flags) when a half is completely published.
    int u_done = 0, r_done = 0, nap = 0;
#pragma unroll 1
    while (u_done < 2 || (wave == 0 && r_done < 2)) {
      if (u_done < 2 && (r_done == u_done || !MP_UNIT_PENDING())) {
        // The units' partials are met at the workgroup barrier, not polled for: a wave that spins on an LDS word next to fourteen
        // streaming waves slows THEIR requests down (measured: the first unit's tiles accepted after 11 us instead of 2.7; a wave asleep
        // in s_barrier costs nothing).  The producers arrive without draining their requests; the second unit's barrier is entered
        // once the first half's flag is up (or at once when that has happened).
        MP_UNIT_MEET();
        const int g = u_done;
        MP_STAMP(1 + 4 * g);
        const float* redU = red + (size_t)g * I * 64;
        const int s = MP_SLOT(g), u = MP_UNIT(g);
        const float v1 = tile_strip_value(redU + (size_t)wave * ips * 64, ips, lane);
        const float v3 = tile_strip_value(redU + (size_t)(2 + wave) * ips * 64, ips, lane);
        const u32 h = __builtin_bit_cast(u32, act_fn(v1, a.act) * v3);  // row (lane & 15) of the strip, in all four lane groups
        const int strip = u * 2 + wave;
        float* dst = a.hb + (size_t)s * a.hb_stride + strip * 16;
        u32x4 hq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          hq[q] = u32x4{(u32)__builtin_amdgcn_readlane((int)h, 4 * q), (u32)__builtin_amdgcn_readlane((int)h, 4 * q + 1),
                        (u32)__builtin_amdgcn_readlane((int)h, 4 * q + 2), (u32)__builtin_amdgcn_readlane((int)h, 4 * q + 3)};
        mp_sstore4x4(dst, hq[0], hq[1], hq[2], hq[3]);
        mp_swb();
        if (mp_satomic_inc_ret(a.blk_ctr + (s * 8 + (strip >> 4)) * 16) == 15u) mp_satomic_inc(a.slot_ctr + g * MOE_CTR_STRIDE);
        MP_STAMP(2 + 4 * g);
        ++u_done;
        continue;
      }
      if (wave == 0 && r_done < u_done && --nap <= 0) {
        if (gave_up || mp_sload_glc(a.slot_ctr + r_done * MOE_CTR_STRIDE) >= half_target) {
          if (lane == 0) mp_signal(&sync_w[MPW_RDY0 + r_done], 1u);
          MP_STAMP(3 + 4 * r_done);
          ++r_done;
          nap = 0;
          continue;
        }
        // between polls: the first half has until the end of phase A (long naps), the second is waited for (short ones)
        nap = r_done == 0 && u_done < 2 ? MP_NAP0 : MP_NAP1;
        if (++spins > (unsigned)a.spin_limit) gave_up = true;
      }
      __builtin_amdgcn_s_sleep(MP_SLEEP);
    }
    if (tl && wave == 1 && lane == 0) MP_TLN(7);
    if (wave == 0 && gave_up && lane == 0) { *a.err = 1u; __hip_atomic_store(a.slot_ctr + MOE_GAVE_UP_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (a.spin_limit < 0 && bid == 0 && wave == 0 && lane == 0) *a.err = 1u;  // fault injection (option "moe_spin_limit" < 0)
    }
  }
  __syncthreads();
#ifndef DSK_NO_TAPS
  if (a.tap_qs && bid == 0)  // parity tap: what the slots staged
    for (int s = 0; s < slots; ++s)
      dump_staged_q8<LAY_TILE>(actB + (size_t)s * a.lds_b, s < K ? a.mi : a.shared_n, a.tap_qs + (size_t)s * a.tap_stride,
                               a.tap_d + (size_t)s * (a.tap_stride >> 8), tid, 1024);
#endif
  // one wave per (slot, tile): the rows' values (association of tile_device.h: one item per block)
  for (int q = wave; q < slots * ntile; q += NW) {
    const int s = q / ntile, tl_ = q - s * ntile;
    const int nbq = s < K ? nbR : nbS;
    const float* rs = red_b + (size_t)(s < K ? (s * ntile + tl_) * nbR : R + tl_ * nbS) * 64;
    const float v = tile_strip_value(rs, nbq, lane);
    if (lane < 16) {
      o_s[s * a.rows_wg + tl_ * 16 + lane] = v;
      a.eout[(size_t)s * a.dim + r_lo + tl_ * 16 + lane] = v;
    }
  }
  __syncthreads();
  if (tl && tid == 0) MP_TLN(5);
  if (tid < nrows) {  // x += w_k * o_k in k order (src/infer.cpp:874-877), then the shared expert (:900-903)
    for (int k = 0; k < K; ++k) xv = fmaf(o_s[k * a.rows_wg + tid], a.route_w[k], xv);
    if (slots > K) xv += o_s[K * a.rows_wg + tid];
    a.x[r_lo + tid] = xv;
  }
  // every slot counter was complete when this workgroup's service waves raised their last flag, so every strip is past its block
  // poll: the block counters go back to zero for the next launch (the slot counters are the router launch's to re-arm)
  if (tid == 0 && bid < K * 8) __hip_atomic_store(a.blk_ctr + bid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tl && tid == 0) MP_TLN(6);
}

// ---- host side ------------------------------------------------------------------------------------------
static size_t moe_pipe_lds(const MoeFfnArgs& a) {
  const int nbA = a.dim >> 8, nbR = a.mi >> 8, nbS = a.shared_n >> 8;
  const int I = 4 * tile_ips(nbA), ntile_max = a.rows_wg / 16;
  const int J = ntile_max * (a.K * nbR + nbS);
  return (size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o + (size_t)(2 * I + J) * 256;
}
// the pipelined form applies to a plan of moe_ffn_plan_tile when its exact deals exist (DeepSeek-V3: 8 slots, 2048-wide experts, 7168-wide x)
static bool moe_halves_applies(const MoeFfnArgs& a) {
  if (!a.tiled || a.quant != DSK_QUANT_Q2_K || a.hq_qs == nullptr) return false;
  if (a.K * 8 > a.grid || a.K * 8 * 16 > MOE_BLK_CTRS) return false;  // (8 block counters per slot, a 64-byte line each, re-armed by the first workgroups)
  const int nbA = a.dim >> 8, nbR = a.mi >> 8, nbS = a.shared_n >> 8, ips = tile_ips(nbA);
  if (nbA <= 8 || (nbA & 3) || 4 * ips != 2 * MP_NP) return false;    // whole 4-block items, exactly two per producer and unit
  if (a.K < 2 || a.K > 8 || (a.K & 1) || a.mi % 256) return false;    // (a half is polled with four scalar loads)
  if ((a.K / 2) * (a.mi / MP_ROWS) != a.grid) return false;          // one unit per workgroup and half
  if (nbR > 8 || nbS > 8) return false;
  const int ntile_max = a.rows_wg / 16;
  if (ntile_max * (nbS + (a.K / 2) * nbR) > MP_NP * MP_NS1 || ntile_max * (a.K / 2) * nbR > MP_NP * MP_NS2) return false;
  if (nbS + (a.K / 2) * nbR > MP_NP * MP_HB || (a.K / 2) * nbR > MP_NP * MP_HB) return false;  // hidden blocks per producer and stage
  return moe_pipe_lds(a) <= 150 * 1024;
}
// the staged form (option "moe_pipe" = 2) applies where the lean one-phase instantiation does and a wave's stage shares fit its registers
static bool moe_stage_applies(const MoeFfnArgs& a) {
  if (!a.tiled || a.quant != DSK_QUANT_Q2_K || a.hq_qs == nullptr) return false;
  const int nbA = a.dim >> 8, nbR = a.mi >> 8, nbS = a.shared_n >> 8;
  if (nbA <= 8 || a.K < 2 || a.K > 8 || nbR > 8 || nbS > 8 || nbS > 15 || a.mi % 64 || a.shared_n == 0) return false;  // (slot K's counter line is re-armed when there are K + 1 slots)
  if (a.K * a.UA != a.grid || (a.mi + 15) / 16 != a.UA * 4) return false;  // one whole 64-row unit per workgroup
  const int ntile_max = a.rows_wg / 16, KH = a.K / 2;
  if (ntile_max * (nbS + KH * nbR) > 16 * MS_NS1 || ntile_max * (a.K - KH) * nbR > 16 * MS_NS2) return false;
  const int itemsA = 8 * tile_ips(nbA), itemsB = ntile_max * (a.K * nbR + nbS);
  return (itemsA > itemsB ? itemsA : itemsB) * 256 <= a.lds_red;
}
bool moe_pipe_applies(const MoeFfnArgs& a) { return a.pipe == 2 ? moe_stage_applies(a) : moe_halves_applies(a); }
int launch_moe_ffn_pipe(hipStream_t st, const MoeFfnArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (a.pipe == 2) {
    const size_t lds = (size_t)a.lds_a + (size_t)a.lds_b * (a.K + 1) + a.lds_o + (size_t)a.lds_red;
    auto k = moe_ffn_stage_kernel;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), (uint32_t)lds, st, ev_start, ev_stop, 0u, a);
    else hipLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), lds, st, a);
    return DSK_OK;
  }
  const size_t lds = moe_pipe_lds(a);
  auto k = moe_ffn_pipe_kernel;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (ev_start && ev_stop) hipExtLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), (uint32_t)lds, st, ev_start, ev_stop, 0u, a);
  else hipLaunchKernelGGL(k, dim3(a.grid + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(1024), lds, st, a);
  return DSK_OK;
}
