// attn_device.h -- device functions of the per-head attention step shared by the fused
// "second-stage projections + attention" kernel (kernels_gemv.hip).  Same arithmetic as the stand-alone
// kernels of kernels_misc.hip (rope_kv_mha_kernel / attn_mha_kernel, which the op-level entry points use).
#pragma once
#include "dsk_internal.h"
#include <math.h>

namespace ad {
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define ADEV __device__ __forceinline__

ADEV float h2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
ADEV unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }  // RNE like _cvtss_sh(x,0), src/codec.h:26-27

// sum over the 16 lanes of a DPP row (every lane gets the total; VALU speed, no ds_bpermute)
ADEV float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// full-wave sum, fixed order: DPP inside the 16-lane rows, then the four rows
ADEV float wave_sum_dpp(float v) {
  v = row16_sum(v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

ADEV float wave_max_dpp(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// block-wide reductions through a small LDS scratch (deterministic order); all threads get the result
ADEV float block_sum(float v, float* scratch, int tid, int nthreads) {
  v = wave_sum_dpp(v);
  const int nw = nthreads >> 6;
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}
ADEV float block_max(float v, float* scratch, int tid, int nthreads) {
  v = wave_max_dpp(v);
  const int nw = nthreads >> 6;
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float t = scratch[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
  return t;
}

// Q8_K quantisation of one 256-block by one wave (quantize_row_q8_K_ref, src/quant.cpp:616-653)
ADEV void q8k_block(const float (&v)[4], int lane, int8_t* qs_blk, float* d_out, int16_t* bsums_blk) {
  float amax = 0.f, vmax = 0.f;
  int imax = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float ax = fabsf(v[i]);
    if (ax > amax) { amax = ax; vmax = v[i]; imax = lane * 4 + i; }
  }
  if (amax == 0.f) imax = lane * 4;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float oa = __shfl_xor(amax, off), ov = __shfl_xor(vmax, off);
    const int oi = __shfl_xor(imax, off);
    if (oa > amax || (oa == amax && oi < imax)) { amax = oa; vmax = ov; imax = oi; }
  }
  int q[4] = {0, 0, 0, 0};
  float d = 0.f;
  if (amax != 0.f) {
    const float iscale = __fdiv_rn(-127.f, vmax);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (int)rintf(__fmul_rn(iscale, v[i]));
      q[i] = r < 127 ? r : 127;
    }
    d = __fmul_rn(vmax, 1.0f / -127.f);
  }
  const u32 packed = (u32)(q[0] & 0xff) | ((u32)(q[1] & 0xff) << 8) | ((u32)(q[2] & 0xff) << 16) | ((u32)(q[3] & 0xff) << 24);
  reinterpret_cast<u32*>(qs_blk)[lane] = packed;
  int s = q[0] + q[1] + q[2] + q[3];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  if ((lane & 3) == 0) bsums_blk[lane >> 2] = (int16_t)s;
  if (lane == 0) *d_out = d;
}


// the same block quantisation with WRITE-THROUGH stores (sc1): for a block whose readers sit on other CUs of the SAME launch
// (kernels_moe.hip: the hidden vectors of the fused expert launch); the caller drains (s_waitcnt vmcnt(0)) before it arrives
ADEV void q8k_block_wt(const float (&v)[4], int lane, int8_t* qs_blk, float* d_out, int16_t* bsums_blk) {
  float amax = 0.f, vmax = 0.f;
  int imax = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float ax = fabsf(v[i]);
    if (ax > amax) { amax = ax; vmax = v[i]; imax = lane * 4 + i; }
  }
  if (amax == 0.f) imax = lane * 4;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float oa = __shfl_xor(amax, off), ov = __shfl_xor(vmax, off);
    const int oi = __shfl_xor(imax, off);
    if (oa > amax || (oa == amax && oi < imax)) { amax = oa; vmax = ov; imax = oi; }
  }
  int q[4] = {0, 0, 0, 0};
  float d = 0.f;
  if (amax != 0.f) {
    const float iscale = __fdiv_rn(-127.f, vmax);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (int)rintf(__fmul_rn(iscale, v[i]));
      q[i] = r < 127 ? r : 127;
    }
    d = __fmul_rn(vmax, 1.0f / -127.f);
  }
  const u32 packed = (u32)(q[0] & 0xff) | ((u32)(q[1] & 0xff) << 8) | ((u32)(q[2] & 0xff) << 16) | ((u32)(q[3] & 0xff) << 24);
  __hip_atomic_store(reinterpret_cast<u32*>(qs_blk) + lane, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int s = q[0] + q[1] + q[2] + q[3];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  const int s2 = __shfl_down(s, 4);  // the next sub-block's sum: two int16 go out as one dword
  if ((lane & 7) == 0) __hip_atomic_store(reinterpret_cast<u32*>(bsums_blk) + (lane >> 3), (u32)(s & 0xffff) | ((u32)s2 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) __hip_atomic_store(d_out, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One rotated pair (src/infer.cpp:659-666) with every product and sum rounded on its own: hipcc contracts `v0 * c - v1 * s` into
// an fma whose choice of operand depends on the surrounding code, so two instantiations of the same source line can differ in
// the last bit (round 3: seen between two instantiations of mla_head_kernel).  This is the oracle's arithmetic (-ffp-contract=off).
// (HIP's __fmul_rn / __fsub_rn are plain operators and contract like them: the pragma is what pins the rounding.)
ADEV void rope_rot(float v0, float v1, float c, float s, float& re, float& im) {
#pragma clang fp contract(off)
  const float a = v0 * c, b = v1 * s, d = v0 * s, e = v1 * c;
  re = a - b;
  im = d + e;
}
// RoPE of the head's query (in LDS, in place), key / value assembly from the LDS copy of this head's
// kv_b rows, f16 cache write at kv_pos, rotation of the attention-sink keys (src/infer.cpp:956-1020).
template <int NT>
ADEV void rope_kv_from_lds(const AttnMhaArgs& a, const StepParams* __restrict__ sp, int h, int tid, float* q_s, const float* kvb_s,
                           bool rotate_sinks = true) {
  const int hd = a.head_dim, nope = a.nope, rope = a.rope, vd = a.v_dim;
  // (split contexts: every split writes the same row at kv_pos, only split 0 -- whose range holds the sink rows -- rotates them)
  const int kv_pos = sp->kv_pos, kv_sink = rotate_sinks ? sp->kv_sink : 0;
  float qre = 0.f, qim = 0.f;
  if (tid < rope / 2) {  // rope (V2: de-interleaving) src/infer.cpp:648-668; rope_v3 :670-685
    const float v0 = q_s[nope + 2 * tid], v1 = q_s[nope + 2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    rope_rot(v0, v1, c, s, qre, qim);
  }
  __syncthreads();
  if (tid < rope / 2) {
    if (a.is_v3) {
      q_s[nope + 2 * tid] = qre;
      q_s[nope + 2 * tid + 1] = qim;
    } else {
      q_s[nope + tid] = qre;
      q_s[nope + tid + rope / 2] = qim;
    }
  }
  // key = [k_nope | rope(k_rope)], value  -> f16 caches at kv_pos
  uint16_t* kc = a.key_cache + ((size_t)kv_pos * a.n_heads + h) * hd;
  uint16_t* vc = a.value_cache + ((size_t)kv_pos * a.n_heads + h) * vd;
  for (int i = tid; i < nope; i += NT) kc[i] = f2h(kvb_s[i]);
  for (int i = tid; i < vd; i += NT) vc[i] = f2h(kvb_s[nope + i]);
  if (tid < rope / 2) {
    const float* kr = a.kv_a + a.lora;
    const float v0 = kr[2 * tid], v1 = kr[2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    float re, im;
    rope_rot(v0, v1, c, s, re, im);
    if (a.is_v3) {
      kc[nope + 2 * tid] = f2h(re);
      kc[nope + 2 * tid + 1] = f2h(im);
    } else {
      kc[nope + tid] = f2h(re);
      kc[nope + tid + rope / 2] = f2h(im);
    }
  }
  // sinks: rotate the rope part of cached keys 0..kv_sink-1 by one position, in f16
  // (src/infer.cpp:1008-1020, rope f16 variants :687-724)
  for (int r = 0; r < kv_sink; ++r) {
    uint16_t* kh = a.key_cache + ((size_t)r * a.n_heads + h) * hd + nope;
    float re = 0.f, im = 0.f;
    if (tid < rope / 2) {
      const float v0 = h2f(kh[2 * tid]), v1 = h2f(kh[2 * tid + 1]);
      const float c = sp->rope_cs1[2 * tid], s = sp->rope_cs1[2 * tid + 1];
      rope_rot(v0, v1, c, s, re, im);
    }
    __syncthreads();
    if (tid < rope / 2) {
      if (a.is_v3) {
        kh[2 * tid] = f2h(re);
        kh[2 * tid + 1] = f2h(im);
      } else {
        kh[tid] = f2h(re);
        kh[tid + rope / 2] = f2h(im);
      }
    }
    __syncthreads();
  }
}

// q: the head's query (global, or the LDS copy the fused kernel rotated in place).  NT threads; part: NT / (v_dim / 4) * v_dim floats.
template <int NT>
// Positions [t_lo, t_hi) of the cache.  ml == nullptr: the whole context, normalised softmax (the reference's
// two-pass arithmetic).  ml != nullptr (split contexts): the weights stay exp(s - m) with m the maximum over the
// range; ml[0] = m, ml[1] = their sum, and the returned mix is un-normalised -- the merge divides once.
ADEV float attn_mha_body(const AttnMhaArgs& a, const float* q, int t_lo, int t_hi, int h, int tid, float* att, float* scratch, float* part,
                         float* ml = nullptr) {
  const int kv_len = t_hi - t_lo;  // att[] and every loop below are relative to t_lo
  const uint16_t* const kc = a.key_cache + (size_t)t_lo * a.n_heads * a.head_dim;
  const uint16_t* const vc = a.value_cache + (size_t)t_lo * a.n_heads * a.v_dim;
  const int wave = tid >> 6, lane = tid & 63, grp = lane >> 4, sl = lane & 15;
  const int hd = a.head_dim, vd = a.v_dim, H = a.n_heads;
  float qv[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int d0 = 64 * j + sl * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) qv[j][i] = d0 < hd ? q[d0 + i] : 0.f;
  }
  const float inv = sqrtf((float)hd);
  // 4 positions per 16-lane group and step (16 per wave, 64 per workgroup): up to 16 eight-byte loads in
  // flight per lane, so a long context streams the cache instead of paying one latency per position
  for (int t0 = wave * 16; t0 < kv_len; t0 += NT / 4) {
    f16x4 k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 4 + grp;
      if (t < kv_len) {
        const uint16_t* kr = kc + ((size_t)t * H + h) * hd;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (64 * j + sl * 4 < hd) k[u][j] = *reinterpret_cast<const f16x4*>(kr + 64 * j + sl * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 4 + grp;
      float p = 0.f;
      if (t < kv_len) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (64 * j + sl * 4 < hd) {
            p = fmaf(qv[j][0], (float)k[u][j].x, p);
            p = fmaf(qv[j][1], (float)k[u][j].y, p);
            p = fmaf(qv[j][2], (float)k[u][j].z, p);
            p = fmaf(qv[j][3], (float)k[u][j].w, p);
          }
      }
      p = row16_sum(p);
      if (sl == 0 && t < kv_len) att[t] = p / inv;
    }
  }
  // the first two value rows of this thread are requested now: their latency hides behind the softmax
  const int tpp = vd >> 2;        // threads per position
  const int TG = NT / tpp;        // positions in flight
  const int g = tid / tpp, i4 = tid - g * tpp;
  f16x4 vpre[2] = {};
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (g < TG && g + k * TG < kv_len)
      vpre[k] = *reinterpret_cast<const f16x4*>(vc + ((size_t)(g + k * TG) * H + h) * vd + i4 * 4);
  __syncthreads();
  if (!ml && kv_len <= 64 && TG >= 32) {  // (TG >= 32: the two prefetched value rows per thread are all of a 64-position context)
    // Short contexts (every position lives in wave 0's lanes): the block-wide max / sum below reduce to wave 0's own DPP trees -
    // the other waves contribute -inf and exact zeros - so every wave can take them from the scores itself, redundantly: the
    // SAME bits as the general path with five workgroup barriers and two scratch walks less, and the partial sums of the value mix
    // are added over the position groups that exist (the others hold exact zeros).  attn tail 2.9 -> ~1.9 us at kv_len <= 24.
    const float a0 = lane < kv_len ? att[lane] : -INFINITY;
    const float mx0 = wave_max_dpp(a0);
    const float e0 = lane < kv_len ? expf(a0 - mx0) : 0.f;
    const float sum0 = wave_sum_dpp(e0);
    const float w0 = e0 / sum0;  // softmax weight of position `lane`
    float acc0[4] = {0.f, 0.f, 0.f, 0.f};
    float wk[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) wk[k] = __shfl(w0, g + k * TG < 64 ? g + k * TG : 0);  // (every lane takes part in the exchange)
    if (g < TG) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = g + k * TG;
        const float w = wk[k];
        if (t < kv_len) {
          acc0[0] = fmaf(w, (float)vpre[k].x, acc0[0]);
          acc0[1] = fmaf(w, (float)vpre[k].y, acc0[1]);
          acc0[2] = fmaf(w, (float)vpre[k].z, acc0[2]);
          acc0[3] = fmaf(w, (float)vpre[k].w, acc0[3]);
        }
      }
      if (g < kv_len) *reinterpret_cast<f32x4*>(part + (size_t)g * vd + i4 * 4) = f32x4{acc0[0], acc0[1], acc0[2], acc0[3]};
    }
    __syncthreads();
    float o0 = 0.f;
    const int ng = TG < kv_len ? TG : kv_len;
    if (tid < vd)
      for (int gg = 0; gg < ng; ++gg) o0 += part[gg * vd + tid];
    return o0;
  }
  // softmax, src/infer.cpp:472-487
  float mx = -INFINITY;
  for (int t = tid; t < kv_len; t += NT) mx = fmaxf(mx, att[t]);
  mx = block_max(mx, scratch, tid, NT);
  float sum = 0.f;
  for (int t = tid; t < kv_len; t += NT) {
    const float e = expf(att[t] - mx);
    att[t] = e;
    sum += e;
  }
  sum = block_sum(sum, scratch, tid, NT);
  if (!ml)
    for (int t = tid; t < kv_len; t += NT) att[t] = att[t] / sum;
  if (ml && tid == 0) { ml[0] = mx; ml[1] = sum; }
  __syncthreads();
  // mix values: thread (g, i4) sums positions g, g+TG, ... for outputs 4*i4..4*i4+3; groups are added in order
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (g < TG) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = g + k * TG;
      if (t < kv_len) {
        const float w = att[t];
        acc[0] = fmaf(w, (float)vpre[k].x, acc[0]);
        acc[1] = fmaf(w, (float)vpre[k].y, acc[1]);
        acc[2] = fmaf(w, (float)vpre[k].z, acc[2]);
        acc[3] = fmaf(w, (float)vpre[k].w, acc[3]);
      }
    }
    for (int t0 = g + 2 * TG; t0 < kv_len; t0 += 4 * TG) {  // 4 rows in flight per thread
      f16x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t0 + k * TG < kv_len) v[k] = *reinterpret_cast<const f16x4*>(vc + ((size_t)(t0 + k * TG) * H + h) * vd + i4 * 4);
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
    *reinterpret_cast<f32x4*>(part + (size_t)g * vd + i4 * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  }
  __syncthreads();
  float o = 0.f;
  if (tid < vd)
    for (int gg = 0; gg < TG; ++gg) o += part[gg * vd + tid];
  return o;
}

// Q8_K copy of the concatenated head outputs for the wo GEMV: the LAST head of a 256-block to arrive
// (write-through stores, one counter per block) quantises the block.  o: this thread's output (tid < v_dim).
ADEV void attn_out_q8(const AttnMhaArgs& a, int h, int tid, float o, int* last_flag) {
  const int vd = a.v_dim;
  if (!a.q_qs) {
    if (tid < vd) a.out[(size_t)h * vd + tid] = o;
    return;
  }
  if (tid < vd) __hip_atomic_store(a.out + (size_t)h * vd + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int b_first = (h * vd) >> 8, b_last = ((h + 1) * vd - 1) >> 8;
  for (int b = b_first; b <= b_last; ++b) {
    if (tid == 0) {
      const int h0 = (b * 256) / vd, h1 = min((b * 256 + 255) / vd, a.n_heads - 1);
      const unsigned old = __hip_atomic_fetch_add(a.q_counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *last_flag = old == (unsigned)(h1 - h0);
      if (*last_flag) {
        FINISHER_ACQUIRE();
        __hip_atomic_store(a.q_counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (*last_flag && tid < 64) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __hip_atomic_load(a.out + (size_t)b * 256 + tid * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q8k_block(v, tid, a.q_qs + (size_t)b * 256, a.q_d + b, a.q_bsums + (size_t)b * 16);
    }
    __syncthreads();
  }
}

// MLA long contexts: merge of mla_flash_kernel's chunk partials for head h - out = sum_c e^(m_c - M) O_c / sum_c e^(m_c - M) l_c -
// by a 1024-thread workgroup; `part` = 64 + 1024 floats of LDS, o_s = the head's lora (<= 512) merged columns, valid after the
// call (it ends with a barrier).  ONE body for the decode launch (kernels_gemv.hip mla_head_kernel) and the batched prompt path
// (kernels_hydrate.hip hyd_mla_merge_kernel): the same statements in the same order, the same bits.
ADEV void mla_merge_partials(const float* __restrict__ part_ml, const float* __restrict__ part_o, int H, int h, int lora, int nc, int tid,
                             float* part, float* o_s) {
  float* wc = part;  // per-chunk weights e^(m_c - M) / L, computed once
  if (tid < 64) {
    const float mc = tid < nc ? part_ml[((size_t)tid * H + h) * 2] : -INFINITY;
    const float lc = tid < nc ? part_ml[((size_t)tid * H + h) * 2 + 1] : 0.f;
    const float M = wave_max_dpp(mc);
    const float e = tid < nc ? expf(mc - M) : 0.f;
    const float Lsum = wave_sum_dpp(e * lc);
    wc[tid] = e / Lsum;
  }
  __syncthreads();
  // column tid & 511, even chunks in the lower half of the workgroup, odd ones in the upper; 8 partials requested per
  // round trip (one dependent load per chunk made this merge ~0.4 us x the number of chunks)
  {
    const int col = tid & 511, half = tid >> 9;
    float o = 0.f;
    for (int c0 = half; c0 < nc; c0 += 16) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + 2 * j;
        v[j] = (c < nc && col < lora) ? part_o[((size_t)c * H + h) * lora + col] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + 2 * j;
        if (c < nc) o = fmaf(wc[c], v[j], o);
      }
    }
    part[64 + tid] = o;
  }
  __syncthreads();
  if (tid < lora && tid < 512) o_s[tid] = part[64 + tid] + part[64 + 512 + tid];
  __syncthreads();
}
}  // namespace ad
