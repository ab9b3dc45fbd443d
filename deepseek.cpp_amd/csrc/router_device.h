// router_device.h -- the MoE router + gate as device functions, shared by kernels_misc.hip (router_gate_kernel,
// gate_kernel) and kernels_gemv.hip (router_shared_kernel: the router launch that also carries the shared expert's
// w1/w3 GLU on the CUs the router leaves idle).
#pragma once
#include "attn_device.h"

namespace rd {
using ad::q8k_block;
using ad::wave_sum_dpp;
typedef ad::f32x4 f32x4;
typedef ad::u32x4 u32x4;
#define RDEV __device__ __forceinline__

// block-wide reductions through a small LDS scratch (xor-shuffle wave trees: the order the gate has always used)
RDEV float block_sum(float v, float* scratch, int tid, int nthreads) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int nw = nthreads >> 6;
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}
RDEV float block_max(float v, float* scratch, int tid, int nthreads) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  const int nw = nthreads >> 6;
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float t = scratch[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
  return t;
}

// sum over the four lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1]); every lane of the quad must execute it
RDEV int quad_sum(int x) {
  x += __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);
  x += __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);
  return x;
}

// ------------------------------------------------------------------------------------
// MoE router + gate in ONE launch.
// Router: F32 GEMV (src/infer.cpp:847, 121-157) over the FFN-normed x (the rmsnorm of
// src/infer.cpp:839 is recomputed in every workgroup's prologue: 28 KB from L2), split over K so
// that E = 256 rows still fill the chip; partial[c][e] are summed in c order (deterministic).
// Gate: the LAST workgroup to arrive (agent-scope release -> relaxed counter -> acquire) runs
// moe_gate, src/infer.cpp:493-599.  The reference's k rounds of "argmax over unmasked with strict >"
// select experts in descending score order, lowest index first among equals; that is the rank of
// e under the order (score desc, index asc), computed by all E threads in parallel:
//   rank(e) = #{ j : s[j] > s[e] or (s[j] == s[e] and j < e) }.
// Group-limited (:545-588): first keep the topk_group best of every group, then rank the
// survivors globally.  Weights: x[e_k] / wsum * scaling with wsum accumulated in k order.
// ------------------------------------------------------------------------------------
RDEV void gate_body(int e, float v, const float* __restrict__ bias, int E, int K, int norm_topk_prob, float scaling, int scoring,
                   int topk_method, int n_group, int topk_group, int* __restrict__ active_experts,
                   float* __restrict__ active_weights, float* __restrict__ scores_out, float* s, int* ci, int* sel, float* scratch,
                   int nthreads = 256, bool have_bv = false, float bv = 0.f) {
  // s[256]: scores; cs (aliases scratch area passed as `s + 256`) / ci[256]: compacted candidates.
  // Called by every thread of the workgroup (barriers inside); threads >= 256 only take part in those.
  float* cs = s + 256;
  if (e >= E) v = -INFINITY;
  if (scoring == DSK_SCORE_SOFTMAX) {  // softmax, src/infer.cpp:472-487
    const float mx = block_max(v, scratch, e, nthreads);
    const float ex = e < E ? expf(v - mx) : 0.f;
    const float sum = block_sum(ex, scratch, e, nthreads);
    v = ex / sum;
  } else {
    v = 1.0f / (1.0f + expf(-v));  // sigmoid, src/infer.cpp:489-491
  }
  if (bias && e < E) v += have_bv ? bv : bias[e];  // (have_bv: the caller loaded bias[e] ahead of time)
  if (e >= E) v = -INFINITY;
  if (e < 256) {
    s[e] = v;
    cs[e] = -INFINITY;
    ci[e] = 0x7fffffff;
  }
  if (e < E && scores_out) scores_out[e] = v;
  __syncthreads();
  // candidates, compacted: a wave64 VALU op takes 4 cycles, so the serial compare loops below are the
  // critical path of the whole launch -- they must run over the candidates only, not over all E
  int ncand = E;
  // With 1024 threads four of them share a candidate: each compares it with a quarter of its rivals (two 16-byte LDS reads in
  // flight instead of a chain of eight round trips) and the quad adds the counts up - ranks are integers: the same result.
  const bool quads = nthreads >= 1024;
  if (topk_method == DSK_TOPK_GROUP_LIMITED_GREEDY) {
    const int gs = E / n_group, tg = topk_group < gs ? topk_group : gs;
    ncand = n_group * tg;
    if (quads && (gs & 15) == 0) {
      const int e2 = e >> 2, p = e & 3, q = gs >> 2;
      const bool live = e2 < E;
      const float v2 = live ? s[e2] : 0.f;
      const int g = live ? e2 / gs : 0;
      int rank = 0;
      if (live) {
        const int j0 = g * gs + p * q;
        for (int j = j0; j < j0 + q; j += 4) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(s + j);
          rank += (sv.x > v2 || (sv.x == v2 && j + 0 < e2)) ? 1 : 0;
          rank += (sv.y > v2 || (sv.y == v2 && j + 1 < e2)) ? 1 : 0;
          rank += (sv.z > v2 || (sv.z == v2 && j + 2 < e2)) ? 1 : 0;
          rank += (sv.w > v2 || (sv.w == v2 && j + 3 < e2)) ? 1 : 0;
        }
      }
      rank = quad_sum(rank);
      if (live && p == 0 && rank < tg) {
        cs[g * tg + rank] = v2;
        ci[g * tg + rank] = e2;
      }
    } else if (e < E) {
      const int g = e / gs, g0 = g * gs;
      int rank = 0;
      if ((gs & 3) == 0) {  // group starts are multiples of 4: 16-byte LDS reads
        for (int j = g0; j < g0 + gs; j += 4) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(s + j);
          rank += (sv.x > v || (sv.x == v && j + 0 < e)) ? 1 : 0;
          rank += (sv.y > v || (sv.y == v && j + 1 < e)) ? 1 : 0;
          rank += (sv.z > v || (sv.z == v && j + 2 < e)) ? 1 : 0;
          rank += (sv.w > v || (sv.w == v && j + 3 < e)) ? 1 : 0;
        }
      } else {
        for (int j = g0; j < g0 + gs; ++j) {
          const float sj = s[j];
          rank += (sj > v || (sj == v && j < e)) ? 1 : 0;
        }
      }
      if (rank < tg) {  // survivor: (group, in-group rank) is a unique slot
        cs[g * tg + rank] = v;
        ci[g * tg + rank] = e;
      }
    }
  } else if (e < E) {
    cs[e] = v;
    ci[e] = e;
  }
  __syncthreads();
  if (quads && (ncand & 15) == 0) {
    const int c = e >> 2, p = e & 3, q = ncand >> 2;
    const bool live = c < ncand;
    const float v2 = live ? cs[c] : 0.f;
    const int i2 = live ? ci[c] : 0;
    int rank = 0;
    if (live) {
      for (int j0 = p * q; j0 < (p + 1) * q; j0 += 4) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(cs + j0);
        const u32x4 iv = *reinterpret_cast<const u32x4*>(ci + j0);
        rank += (sv.x > v2 || (sv.x == v2 && (int)iv.x < i2)) ? 1 : 0;
        rank += (sv.y > v2 || (sv.y == v2 && (int)iv.y < i2)) ? 1 : 0;
        rank += (sv.z > v2 || (sv.z == v2 && (int)iv.z < i2)) ? 1 : 0;
        rank += (sv.w > v2 || (sv.w == v2 && (int)iv.w < i2)) ? 1 : 0;
      }
    }
    rank = quad_sum(rank);
    if (live && p == 0 && rank < K && i2 < E) sel[rank] = i2;
  } else if (e < ncand) {
    const float v2 = cs[e];
    const int i2 = ci[e];
    int rank = 0;
    for (int j0 = 0; j0 < ncand; j0 += 4) {  // arrays are padded with (-inf, INT_MAX) up to 256
      const f32x4 sv = *reinterpret_cast<const f32x4*>(cs + j0);
      const u32x4 iv = *reinterpret_cast<const u32x4*>(ci + j0);
      rank += (sv.x > v2 || (sv.x == v2 && (int)iv.x < i2)) ? 1 : 0;
      rank += (sv.y > v2 || (sv.y == v2 && (int)iv.y < i2)) ? 1 : 0;
      rank += (sv.z > v2 || (sv.z == v2 && (int)iv.z < i2)) ? 1 : 0;
      rank += (sv.w > v2 || (sv.w == v2 && (int)iv.w < i2)) ? 1 : 0;
    }
    if (rank < K && i2 < E) sel[rank] = i2;
  }
  __syncthreads();
  if (K > 64) {  // (never the case for a DeepSeek config; kept for the op-level entry point)
    if (e == 0) {
      float wsum = 0.f;
      for (int k = 0; k < K; ++k) wsum += s[sel[k]];
      if (!norm_topk_prob) wsum = 1.0f;
      for (int k = 0; k < K; ++k) {
        active_experts[k] = sel[k];
        active_weights[k] = s[sel[k]] / wsum * scaling;
      }
    }
  } else if (e < 64) {  // weights: x[e_k] / wsum * scaling, wsum accumulated in k order (src/infer.cpp:590-598)
    const int ek = e < K ? sel[e] : 0;
    const float sk = e < K ? s[ek] : 0.f;
    float wsum = 0.f;
    for (int k = 0; k < K; ++k) wsum += __shfl(sk, k);
    if (!norm_topk_prob) wsum = 1.0f;
    if (e < K) {
      active_experts[e] = ek;
      active_weights[e] = sk / wsum * scaling;
    }
  }
}


// rmsnorm scale of the residual stream (src/infer.cpp:839, 601-611), computed by a whole 1024-thread workgroup; every
// caller gets the same bits (the shared expert's launch-mates quantise x with exactly the router's scale)
RDEV float router_norm_scale(const RouterArgs& a, int tid, float* scratch) {
  const int lane = tid & 63, wave = tid >> 6, dim = a.dim;
  float scale = 1.0f;
  {
    float ss = 0.f;
    for (int i0 = tid * 4; i0 < dim; i0 += 4 * 4096) {
      f32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + k * 4096 < dim) v[k] = *reinterpret_cast<const f32x4*>(a.x + i0 + k * 4096);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + k * 4096 < dim) {
          ss = fmaf(v[k].x, v[k].x, ss);
          ss = fmaf(v[k].y, v[k].y, ss);
          ss = fmaf(v[k].z, v[k].z, ss);
          ss = fmaf(v[k].w, v[k].w, ss);
        }
    }
    ss = wave_sum_dpp(ss);
    if (lane == 0) scratch[wave] = ss;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 4) total += (scratch[i] + scratch[i + 1]) + (scratch[i + 2] + scratch[i + 3]);
    scale = 1.0f / sqrtf(total / (float)dim + a.eps);
    __syncthreads();
  }
  return scale;
}

// 16-wave workgroups: RW rows x (16 / RW) column slices each, slices summed in LDS in slice order, so
// one launch is E / RW workgroups = E / RW arrivals on the counter (256 arrivals on one address cost 2 us).
// bid / nblocks: this workgroup's index among the router workgroups of the launch, and their number
template <int RW>
RDEV void router_body(const RouterArgs& a, int bid, int nblocks) {
  constexpr int SL = 16 / RW;
  __shared__ __attribute__((aligned(16))) float s[512];
  __shared__ __attribute__((aligned(16))) int surv[256];
  __shared__ float scratch[16];
  __shared__ float part[16];
  __shared__ int sel[256];
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = wave % RW, sl = wave / RW;
  const int row = bid * RW + r;
  const int dim = a.dim, E = a.n_routed;
  unsigned long long* tl = a.timeline && bid < DSK_TL_WGS ? a.timeline + (size_t)bid * 8 : nullptr;
  // (stamps 0 and 1 are kept in registers until the rows are done: a store in flight makes hipcc wait for EVERY outstanding
  // operation at the next use of a loaded register - here: for the weight rows before the norm could touch x)
  const unsigned long long t_entry = tl ? wall_clock64() : 0ull;
  unsigned long long t_scale = 0ull;
  // the gate's bias, requested now by every workgroup: the last arriver would otherwise wait for it (a cold line) behind the scores
  const float bias_v = a.bias && tid < E ? a.bias[tid] : 0.f;
  // (Round 5, measured and rejected: this wave's weight rows - and its slices of x and of the norm weights - requested before the
  // norm instead of behind it.  128 KB of requests per CU stall the waves in issue until the first of them are back: the norm scale
  // is known at 3.5 us instead of 1.25, rows done 4.1 -> 4.5, the launch 10.2 -> 10.5 us.)
  // (Round 6, measured and rejected as well: the order that works for the projections (kernels_gemv.hip gemv_ahead_kernel) - x for
  // the norm FIRST, a request barrier, then 2 or 4 of the wave's weight loads, the norm with LDS-only barriers: the norm scale is
  // known at 2.9 us instead of 1.4, rows done 4.6 instead of 4.3, the launch 12.8 -> 13.2 us in situ.  Whatever is queued behind
  // x delays x: 28 KB of x and 57 KB of rows per CU are more than the CU's window, unlike the projections' 19 - 43 KB.)
  const float scale = a.norm_w && !(a.dbg & 4) ? router_norm_scale(a, tid, scratch) : 1.0f;
  if (tl) t_scale = wall_clock64();
  float acc = 0.f;
  if (row < E && !(a.dbg & 2)) {
    const int chunk = ((dim / 4 + SL - 1) / SL + 63) / 64 * 64 * 4;  // floats per column slice, multiple of 256
    const int k0 = sl * chunk, k1 = min(dim, k0 + chunk);
    const float* wr = a.w + (size_t)row * dim;
    for (int i0 = k0 + lane * 4; i0 < k1; i0 += 8 * 256) {  // 8 weight loads in flight per lane
      f32x4 wv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k * 256 < k1) wv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wr + i0 + k * 256));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + k * 256;
        if (i < k1) {
          f32x4 y = *reinterpret_cast<const f32x4*>(a.x + i);
          if (a.norm_w) {
            const f32x4 nw = *reinterpret_cast<const f32x4*>(a.norm_w + i);
            y.x = y.x * scale * nw.x;
            y.y = y.y * scale * nw.y;
            y.z = y.z * scale * nw.z;
            y.w = y.w * scale * nw.w;
          }
          acc = fmaf(wv[k].x, y.x, acc);
          acc = fmaf(wv[k].y, y.y, acc);
          acc = fmaf(wv[k].z, y.z, acc);
          acc = fmaf(wv[k].w, y.w, acc);
        }
      }
    }
    acc = wave_sum_dpp(acc);
  }
  if (tl && tid == 0) { tl[0] = t_entry; tl[1] = t_scale; }
  if (a.q_qs && a.norm_w && wave == 0) {  // Q8_K of rmsnorm(x): block b by workgroup b (src/quant.cpp:616-653)
    for (int b = bid; b < dim / 256; b += nblocks) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + b * 256 + lane * 4);
      const f32x4 nw = *reinterpret_cast<const f32x4*>(a.norm_w + b * 256 + lane * 4);
      const float v[4] = {xv.x * scale * nw.x, xv.y * scale * nw.y, xv.z * scale * nw.z, xv.w * scale * nw.w};
      q8k_block(v, lane, a.q_qs + (size_t)b * 256, a.q_d + b, a.q_bsums + (size_t)b * 16);
    }
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (tid < RW && bid * RW + tid < E) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) v += part[k * RW + tid];  // slice order: deterministic
    // write-through (sc1) store: visible across XCDs once vmcnt drains, no L2 write-back fence needed
    __hip_atomic_store(a.partial + bid * RW + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tl && tid == 0) tl[2] = wall_clock64();
  // ---- publish the scores; the last workgroup to arrive runs the gate ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = old == (unsigned)nblocks - 1;
    if (is_last) FINISHER_ACQUIRE();
  }
  __syncthreads();
  if (tl && tid == 0) tl[3] = wall_clock64();
  if (!is_last) return;
  if (a.dbg & 1) { if (tid == 0) *a.counter = 0; return; }
  if (tid == 0) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
  if (a.zero_ctr && tid < a.zero_n) __hip_atomic_store(a.zero_ctr + tid * MOE_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float v = 0.f;
  if (tid < E) v = __hip_atomic_load(a.partial + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  gate_body(tid, v, a.bias, E, a.n_active, a.norm_topk_prob, a.scaling, a.scoring, a.topk_method, a.n_group, a.topk_group,
            a.active_experts, a.active_weights, a.scores_out, s, surv, sel, scratch, 1024, true, bias_v);
  if (tl && tid == 0) tl[4] = wall_clock64();
}
// MLA model path, the work of ONE small workgroup: rmsnorm of the latent (src/infer.cpp:1089), f16 cache entries of
// this position (:1092-1097), rotation of the sink keys (:1103-1110).  Runs either as mla_kv_write_kernel (256 threads)
// or as the last workgroup of the second-stage projection launch (gemv_kvwrite_kernel, 1024 threads): only the first
// 256 threads index the data and the extra waves add exact zeros to the block sum, so both give the same bits.
RDEV void mla_kv_write_body(const MlaKvArgs& a, const StepParams* __restrict__ sp, int tid, int nthreads) {
  __shared__ float scratch[16];
  const int rope = a.rope, lora = a.lora;
  const int kv_pos = sp->kv_pos, kv_sink = sp->kv_sink;
  const bool work = tid < 256;
  float ss = 0.f;
  if (work)
    for (int i = tid; i < lora; i += 256) ss = fmaf(a.kv_a[i], a.kv_a[i], ss);
  ss = block_sum(ss, scratch, tid, nthreads);
  const float scale = 1.0f / sqrtf(ss / (float)lora + a.eps);
  uint16_t* nc = a.nope_cache + (size_t)kv_pos * lora;
  uint16_t* rc = a.rope_cache + (size_t)kv_pos * rope;
  if (work)
    for (int i = tid; i < lora; i += 256) {
      // The reference rounds the product to f32 and THEN to f16 (src/infer.cpp:1092-1095, _cvtss_sh).  hipcc folds the last
      // product into the conversion (v_fma_mixlo_f16: ONE rounding) in the scalar form of this loop and not in its two-wide
      // form (v_pk_mul_f32 + v_cvt_pk_f16_f32), so instantiations of this body disagreed in one entry of ~500 by one f16
      // place (round 3: seen when the body was instantiated with write-through stores).  The empty asm pins the f32 rounding.
      float y = a.kv_a[i] * scale * a.norm_w[i];
      asm volatile("" : "+v"(y));
      nc[i] = ad::f2h(y);
    }
  if (tid < rope / 2) {
    const float* kr = a.kv_a + lora;
    const float v0 = kr[2 * tid], v1 = kr[2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    float re, im;
    ad::rope_rot(v0, v1, c, s, re, im);
    if (a.is_v3) {
      rc[2 * tid] = ad::f2h(re);
      rc[2 * tid + 1] = ad::f2h(im);
    } else {
      rc[tid] = ad::f2h(re);
      rc[tid + rope / 2] = ad::f2h(im);
    }
  }
  for (int r = 0; r < kv_sink; ++r) {
    uint16_t* kh = a.rope_cache + (size_t)r * rope;
    float re = 0.f, im = 0.f;
    if (tid < rope / 2) {
      const float v0 = ad::h2f(kh[2 * tid]), v1 = ad::h2f(kh[2 * tid + 1]);
      const float c = sp->rope_cs1[2 * tid], s = sp->rope_cs1[2 * tid + 1];
      ad::rope_rot(v0, v1, c, s, re, im);
    }
    __syncthreads();
    if (tid < rope / 2) {
      if (a.is_v3) {
        kh[2 * tid] = ad::f2h(re);
        kh[2 * tid + 1] = ad::f2h(im);
      } else {
        kh[tid] = ad::f2h(re);
        kh[tid + rope / 2] = ad::f2h(im);
      }
    }
    __syncthreads();
  }
}
}  // namespace rd
