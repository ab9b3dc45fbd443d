// gemv_device.h -- device-side building blocks of the weight-streaming GEMV kernels (gfx950, wave64), shared by
// kernels_gemv.hip (the generic multi-task GEMV, the per-head attention kernels) and kernels_moe.hip (the fused routed-expert
// FFN).  Item arithmetic of the K-quants (src/quant.cpp:434-783), Q8_K staging in LDS (src/quant.cpp:616-653 fused with
// rmsnorm, src/infer.cpp:601-611), chunked plane loads through buffer descriptors, fixed-order reductions.
#pragma once
#include "dsk_internal.h"
#include "attn_device.h"
#include "router_device.h"
#include <cstdlib>
#include <type_traits>
#include <hip/hip_ext.h>

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__

// hipcc would select the two-operand v_dot4c for this builtin and zero the accumulator of every chain with a v_mov: 4 of
// ~75 instructions per Q2_K item in VALU-bound kernels.  The Makefile therefore builds the kernels without the dot6-insts
// target feature (NODOT4C): the compiler then picks the three-operand v_dot4_i32_i8 with the literal 0 by itself, hazards
// handled.  (The same instruction through inline asm is NOT an option: the hazard recogniser cannot see a DOT inside an
// asm statement and omits the wait states gfx950 needs between a DOT write and a different opcode's read of the same
// register - measured: garbage, then a memory fault.)
DEV int sdot4(u32 a, u32 b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
DEV float h2f(u32 bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); }
// NB: __builtin_bit_cast applied directly to an ext_vector component (w.y) is miscompiled by hipcc 7.2
// (every component reads element 0); always go through a by-value scalar.
DEV float u2f(u32 v) { return __builtin_bit_cast(float, v); }
template <typename T>
DEV T ldg_nt(const T* p) { return __builtin_nontemporal_load(p); }

DEV float act_fn(float x, int act) {
  if (act == DSK_ACT_SILU) return x / (1.0f + expf(-x));                       // src/infer.cpp:640-642
  return 0.5f * x * (1.0f + tanhf(0.797885f * (x + 0.044715f * x * x * x)));  // src/infer.cpp:636-638
}

// ------------------------------------------------------------------------------------
// Activation staging layout in LDS (K-quants): one 80-byte record per item i = 4*block + quarter,
// quarter q = 2*h + lh:
//   [ 0..63]  the 64 int8 activations the item multiplies: 4 runs of 16 (s = 0..3), run s = elements
//             128*h + 32*s + 16*lh .. +15 of the block (sub-block j = 8*h + 2*s + lh)
//   [64..79]  meta.  Q2_K: int8 bsum_hi[4] | uint8 bsum_lo[4] | f32 d/16 | f32 d
//                    Q3_K: int16 bsum[4] | f32 d | pad
// A lane reads its record with 5 ds_read_b128 at immediate offsets; the 80-byte stride spreads 16
// consecutive lanes over all 64 banks.  Consecutive lanes take consecutive items, and the lanes-per-row
// count is a multiple of 4, so a lane's quarter never changes and item -> address is one add.
// ------------------------------------------------------------------------------------
#define ITEM_LDS 80
DEV size_t kq_lds_bytes(int n) { return (size_t)(n >> 6) * ITEM_LDS; }
// staging layouts (template parameter LAY of the staging routines): item records with Q3_K / Q2_K meta, or the block records
// of the tiled Q2_K path (tile_device.h: codes[256] linear | zeros[16] | (bsum hi[4], lo[4]) x 4 | d); a block is 320 bytes
// in all three
#define LAY_Q3 0
#define LAY_Q2 1
#define LAY_TILE 2

// ------------------------------------------------------------------------------------
// Q2_K: one item = (block b, quarter q): bytes qs[32*h + 16*lh .. +15].  Word k of the item, shifted by
// 2*s, holds elements 128*h + 32*s + 16*lh + 4*k .. +3 (layout: dequantize_row_q2_K,
// src/quant.cpp:217-247).  Scalar spec being computed: src/quant.cpp:746-780.
// VALU budget: at 5.4 TB/s the chip affords ~130 wave instructions per item, so every op counts:
// fields are masked in place (x1, x4, x16, x16), the scale products are 24-bit mads, the min term is
// two dot4 against the split bsums, and the 1/16 is folded into the staged activation scale (exact).
// ------------------------------------------------------------------------------------
DEV float q2k_item(u32x4 w, u32 scw, u32 dm, const u32x4 (&a)[4], u32x4 meta, float acc) {
  int x0 = sdot4(w.x & 0x03030303u, a[0].x, 0);
  x0 = sdot4(w.y & 0x03030303u, a[0].y, x0);
  x0 = sdot4(w.z & 0x03030303u, a[0].z, x0);
  x0 = sdot4(w.w & 0x03030303u, a[0].w, x0);
  int x1 = sdot4(w.x & 0x0C0C0C0Cu, a[1].x, 0);
  x1 = sdot4(w.y & 0x0C0C0C0Cu, a[1].y, x1);
  x1 = sdot4(w.z & 0x0C0C0C0Cu, a[1].z, x1);
  x1 = sdot4(w.w & 0x0C0C0C0Cu, a[1].w, x1);
  int x2 = sdot4(w.x & 0x30303030u, a[2].x, 0);
  x2 = sdot4(w.y & 0x30303030u, a[2].y, x2);
  x2 = sdot4(w.z & 0x30303030u, a[2].z, x2);
  x2 = sdot4(w.w & 0x30303030u, a[2].w, x2);
  int x3 = sdot4((w.x >> 2) & 0x30303030u, a[3].x, 0);
  x3 = sdot4((w.y >> 2) & 0x30303030u, a[3].y, x3);
  x3 = sdot4((w.z >> 2) & 0x30303030u, a[3].z, x3);
  x3 = sdot4((w.w >> 2) & 0x30303030u, a[3].w, x3);
  const int d0 = scw & 0xF, d1 = (scw >> 8) & 0xF, d2 = (scw >> 16) & 0xF, d3 = (scw >> 24) & 0xF;
  // 16 * (sum_s d_s * true x_s): x1 carries a factor 4, x2 / x3 a factor 16; all products < 2^23
  const int t23 = __mul24(d3, x3) + __mul24(d2, x2);
  const int t1 = (__mul24(d1, x1) << 2) + t23;
  const int isum16 = (__mul24(d0, x0) << 4) + t1;
  const u32 m4 = (scw >> 4) & 0x0F0F0F0Fu;
  const int summs = (sdot4(m4, meta.x, 0) << 8) + (int)__builtin_amdgcn_udot4(m4, meta.y, 0u, false);
  const float dall16 = u2f(meta.z) * h2f(dm & 0xffff);
  const float dmin = u2f(meta.w) * h2f(dm >> 16);
  acc = fmaf(dall16, (float)isum16, acc);
  acc = fmaf(-dmin, (float)summs, acc);
  return acc;
}

// ------------------------------------------------------------------------------------
// Q3_K: value = (qs >> 2s) & 3 | hbit << 2, minus 4; hbit of element 128h+32s+l is bit 4h+s
// of hmask[l]; 6-bit scales minus 32 (src/quant.cpp:384-432, scalar spec :558-610).
// ------------------------------------------------------------------------------------
DEV int q3k_scale(u32 a0, u32 a1, u32 a2, int j) {  // j = 0..15 (src/quant.cpp:592-597)
  const int jj = j & 7;
  const u32 word = (jj < 4) ? a0 : a1;
  u32 byte = (word >> (8 * (jj & 3))) & 0xFF;
  const u32 low4 = (j < 8) ? (byte & 0xF) : (byte >> 4);
  const u32 hi2 = (a2 >> (8 * (j & 3) + 2 * (j >> 2))) & 3;
  return (int)(low4 | (hi2 << 4)) - 32;
}

DEV float q3k_item(u32x4 w, u32x4 hm, u32 s0, u32 s1, u32 s2, u32 d16, int h, int lh, const u32x4 (&a)[4],
                   u32x4 meta, float acc) {
  const int bsv[4] = {(int)(short)(meta.x & 0xffff), (int)meta.x >> 16, (int)(short)(meta.y & 0xffff), (int)meta.y >> 16};
  int total = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int bit = 4 * h + s;
    u32 v0 = ((w.x >> (2 * s)) & 0x03030303u) | (((hm.x >> bit) & 0x01010101u) << 2);
    u32 v1 = ((w.y >> (2 * s)) & 0x03030303u) | (((hm.y >> bit) & 0x01010101u) << 2);
    u32 v2 = ((w.z >> (2 * s)) & 0x03030303u) | (((hm.z >> bit) & 0x01010101u) << 2);
    u32 v3 = ((w.w >> (2 * s)) & 0x03030303u) | (((hm.w >> bit) & 0x01010101u) << 2);
    int x = sdot4(v0, a[s].x, 0);
    x = sdot4(v1, a[s].y, x);
    x = sdot4(v2, a[s].z, x);
    x = sdot4(v3, a[s].w, x);
    x -= 4 * bsv[s];  // the "- 4" of every element of the sub-block
    total += __mul24(q3k_scale(s0, s1, s2, 8 * h + 2 * s + lh), x);
  }
  return fmaf(u2f(meta.z) * h2f(d16), (float)total, acc);
}

// ------------------------------------------------------------------------------------
// Q8_K quantisation of one 256-block by one wave (quantize_row_q8_K_ref, src/quant.cpp:616-653;
// same arithmetic as kernels_misc.hip q8k_block) writing the item-record layout described above.
// ------------------------------------------------------------------------------------
// DPP lane exchanges (VALU speed; __shfl_xor lowers to ds_bpermute, ~100 cycles each)
template <int CTRL>
DEV u32 dpp_u32(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
DEV float dpp_f32(float v) { return u2f(dpp_u32<CTRL>(__builtin_bit_cast(u32, v))); }
#define DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
// max over the wave of a non-negative float's bit pattern, as a wave-uniform value
DEV u32 wave_max_bits(u32 v) {
  v = max(v, dpp_u32<DPP_XOR1>(v));
  v = max(v, dpp_u32<DPP_XOR2>(v));
  v = max(v, dpp_u32<DPP_HALF_MIRROR>(v));
  v = max(v, dpp_u32<DPP_MIRROR>(v));  // every lane of a 16-lane row holds the row maximum
  const u32 a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const u32 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

// write one lane's share of a staged block: 4 consecutive int8 (elements 4*lane..+3), the sub-block
// sum (lanes 4j) and the block scale (lanes 0..3, one per quarter record)
template <int LAY>
DEV void q8k_store_lds(u32 packed, int quadsum, float d, int lane, uint8_t* blk) {
  constexpr bool Q2META = LAY == LAY_Q2;
  if (LAY == LAY_TILE) {
    *reinterpret_cast<u32*>(blk + lane * 4) = packed;
    if ((lane & 3) == 0) {
      const int j = lane >> 2;
      blk[272 + 8 * (j >> 2) + (j & 3)] = (uint8_t)(quadsum >> 8);
      blk[276 + 8 * (j >> 2) + (j & 3)] = (uint8_t)(quadsum & 0xff);
    }
    if (lane < 4) *reinterpret_cast<u32*>(blk + 256 + lane * 4) = 0u;
    if (lane == 0) *reinterpret_cast<float*>(blk + 304) = d;
    return;
  }
  const int h = lane >> 5, sidx = (lane >> 3) & 3, lh = (lane >> 2) & 1;
  uint8_t* rec = blk + (2 * h + lh) * ITEM_LDS;
  *reinterpret_cast<u32*>(rec + sidx * 16 + (lane & 3) * 4) = packed;
  if ((lane & 3) == 0) {
    if (Q2META) {
      rec[64 + sidx] = (uint8_t)(quadsum >> 8);
      rec[68 + sidx] = (uint8_t)(quadsum & 0xff);
    } else {
      reinterpret_cast<short*>(rec + 64)[sidx] = (short)quadsum;
    }
  }
  if (lane < 4) {
    float* m = reinterpret_cast<float*>(blk + lane * ITEM_LDS + 72);
    if (Q2META) { m[0] = d * 0.0625f; m[1] = d; }
    else m[0] = d;
  }
}

// rounding half of quantize_row_q8_K_ref given the block's signed max (src/quant.cpp:630-650)
template <int LAY>
DEV void q8k_round_lds(const float (&v)[4], float vmax, int lane, uint8_t* blk) {
  int q[4] = {0, 0, 0, 0};
  float d = 0.f;
  if (vmax != 0.f) {
    const float iscale = __fdiv_rn(-127.f, vmax);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (int)rintf(__fmul_rn(iscale, v[i]));
      q[i] = r < 127 ? r : 127;
    }
    d = __fmul_rn(vmax, 1.0f / -127.f);
  }
  const u32 packed = (u32)(q[0] & 0xff) | ((u32)(q[1] & 0xff) << 8) | ((u32)(q[2] & 0xff) << 16) | ((u32)(q[3] & 0xff) << 24);
  int sum = q[0] + q[1] + q[2] + q[3];
  sum += (int)dpp_u32<DPP_XOR1>((u32)sum);
  sum += (int)dpp_u32<DPP_XOR2>((u32)sum);
  q8k_store_lds<LAY>(packed, sum, d, lane, blk);
}

template <int LAY>
DEV void q8k_block_lds(const float (&v)[4], int lane, uint8_t* blk) {
  // max = signed value of the FIRST element with the largest |x| (src/quant.cpp:622-629)
  float amax_l = 0.f, vmax_l = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float ax = fabsf(v[i]);
    if (ax > amax_l) { amax_l = ax; vmax_l = v[i]; }
  }
  const u32 amax_bits = wave_max_bits(__builtin_bit_cast(u32, amax_l));
  // the lowest lane holding the maximum owns the first occurrence (lanes hold consecutive elements)
  const unsigned long long owners = __ballot(__builtin_bit_cast(u32, amax_l) == amax_bits);
  const int owner = __ffsll((long long)owners) - 1;
  const float vmax = u2f(__builtin_amdgcn_readlane(__builtin_bit_cast(u32, vmax_l), owner));
  q8k_round_lds<LAY>(v, vmax, lane, blk);
}

// workgroup barrier for LDS traffic only: unlike __syncthreads() (a fence: s_waitcnt vmcnt(0) first) it leaves this wave's
// global loads in flight -- the weight chunk requested ahead of the staging prologue keeps streaming across it
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// orders the workgroup's REQUESTS only (no wait for any data): what was requested before it by any wave is queued in the
// CU's L1 ahead of what any wave requests after it
DEV void issue_barrier() { asm volatile("s_barrier" ::: "memory"); }

DEV float wave_sum(float v) {  // fixed order: quads, rows of 16, then the four rows
  v += dpp_f32<DPP_XOR1>(v);
  v += dpp_f32<DPP_XOR2>(v);
  v += dpp_f32<DPP_HALF_MIRROR>(v);
  v += dpp_f32<DPP_MIRROR>(v);
  const float a = u2f(__builtin_amdgcn_readlane(__builtin_bit_cast(u32, v), 0));
  const float b = u2f(__builtin_amdgcn_readlane(__builtin_bit_cast(u32, v), 16));
  const float c = u2f(__builtin_amdgcn_readlane(__builtin_bit_cast(u32, v), 32));
  const float d = u2f(__builtin_amdgcn_readlane(__builtin_bit_cast(u32, v), 48));
  return (a + b) + (c + d);
}

// sum over the 2^lpr_log2 lanes that share a row (every lane of the group gets the total);
// fixed order, DPP inside a 16-lane row, ds_bpermute only across rows
DEV float lanes_sum(float v, int lpr_log2) {
  if (lpr_log2 >= 1) v += dpp_f32<DPP_XOR1>(v);
  if (lpr_log2 >= 2) v += dpp_f32<DPP_XOR2>(v);
  if (lpr_log2 >= 3) v += dpp_f32<DPP_HALF_MIRROR>(v);
  if (lpr_log2 >= 4) v += dpp_f32<DPP_MIRROR>(v);
  if (lpr_log2 >= 5) v += __shfl_xor(v, 16);
  if (lpr_log2 >= 6) v += __shfl_xor(v, 32);
  return v;
}

// fixed-order sum of the per-wave partials
template <int NW>
DEV float scratch_total(const float* scratch) {
  if (NW == 4) return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; i += 4) t += (scratch[i] + scratch[i + 1]) + (scratch[i + 2] + scratch[i + 3]);
  return t;
}

// sum of squares of x[0..n) over the whole workgroup (NW waves), deterministic order
template <int NW>
DEV float wg_sumsq(const float* __restrict__ x, int n, int tid, float* scratch) {
  float ss = 0.f;
  for (int i0 = tid * 4; i0 < n; i0 += 8 * NW * 256) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * NW * 256 < n) v[k] = *reinterpret_cast<const f32x4*>(x + i0 + k * NW * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * NW * 256 < n) {
        ss = fmaf(v[k].x, v[k].x, ss);
        ss = fmaf(v[k].y, v[k].y, ss);
        ss = fmaf(v[k].z, v[k].z, ss);
        ss = fmaf(v[k].w, v[k].w, ss);
      }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) scratch[tid >> 6] = ss;
  __syncthreads();
  const float t = scratch_total<NW>(scratch);
  __syncthreads();
  return t;
}

// Stage one activation vector of a K-quant task in LDS (item records, see ITEM_LDS).  The VALU work of
// the quantisation (~90 wave instructions per 256-block) is the serial part of every launch, which is
// why big launches run 16-wave workgroups: 4x fewer blocks per wave, 4x fewer redundant prologues.
// The activation source of a launch whose workgroups all stage the SAME vector, passed as plain kernel
// arguments (gfx950 preloads the leading kernel arguments into SGPRs, -mllvm -amdgpu-kernarg-preload-count):
// the staging loads leave before the (cold) descriptor has been read instead of after it.
struct ActSrc {
  int act_mode, n;
  const int8_t* a_qs;
  const float* a_d;
  const int16_t* a_bsums;
  const float* a_f32;
  const float* norm_w;
  float eps;
  float pre_scale;  // > 0: the rmsnorm scale is already known (router_shared_kernel passes the router's own bits)
};
DEV float pre_scale_of(const ActSrc& s) { return s.pre_scale; }
DEV float pre_scale_of(const GemvTask&) { return 0.f; }

template <int LAY, int NW, typename SRC>
DEV void stage_q8(const SRC& T, uint8_t* lds, int tid, float* scratch, unsigned long long* tl = nullptr) {
  constexpr bool Q2META = LAY == LAY_Q2;
  const int n = T.n, nb = n >> 8;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (LAY == LAY_TILE && T.act_mode == ACT_Q8) {  // ready Q8_K vector into block records
    for (int i = tid; i < (n >> 4); i += NW * 64) {
      const int b = i >> 4, j = i & 15;
      uint8_t* rec = lds + (size_t)b * 320;
      *reinterpret_cast<u32x4*>(rec + j * 16) = reinterpret_cast<const u32x4*>(T.a_qs)[i];
      const int bs = T.a_bsums[i];
      rec[272 + 8 * (j >> 2) + (j & 3)] = (uint8_t)(bs >> 8);
      rec[276 + 8 * (j >> 2) + (j & 3)] = (uint8_t)(bs & 0xff);
    }
    for (int b = tid; b < nb; b += NW * 64) {
      *reinterpret_cast<u32x4*>(lds + (size_t)b * 320 + 256) = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<float*>(lds + (size_t)b * 320 + 304) = T.a_d[b];
    }
    return;
  }
  if (T.act_mode == ACT_Q8) {  // ready Q8_K vector: 16-byte runs (one sub-block each) go straight to their record
    for (int i = tid; i < (n >> 4); i += NW * 64) {
      const int b = i >> 4, j = i & 15, h = j >> 3, sidx = (j >> 1) & 3, lh = j & 1;
      uint8_t* rec = lds + (size_t)(b * 4 + 2 * h + lh) * ITEM_LDS;
      *reinterpret_cast<u32x4*>(rec + sidx * 16) = reinterpret_cast<const u32x4*>(T.a_qs)[i];
      const int bs = T.a_bsums[i];
      if (Q2META) {
        rec[64 + sidx] = (uint8_t)(bs >> 8);
        rec[68 + sidx] = (uint8_t)(bs & 0xff);
      } else {
        reinterpret_cast<short*>(rec + 64)[sidx] = (short)bs;
      }
    }
    for (int i = tid; i < nb * 4; i += NW * 64) {
      const float d = T.a_d[i >> 2];
      float* m = reinterpret_cast<float*>(lds + (size_t)i * ITEM_LDS + 72);
      if (Q2META) { m[0] = d * 0.0625f; m[1] = d; }
      else m[0] = d;
    }
    return;
  }
  constexpr int KB1 = 32 / NW;  // blocks per wave of the single-pass path
  if (T.act_mode == ACT_F32_NORM && nb <= 32) {
    // rmsnorm (src/infer.cpp:601-611) + Q8_K in ONE memory round trip: wave w owns blocks w, w+NW, ...;
    // x and the norm weight are loaded once and stay in registers across the sum-of-squares reduction.
    f32x4 t[KB1], wv[KB1];
#pragma unroll
    for (int k = 0; k < KB1; ++k) {
      const int b = wave + NW * k;
      if (b < nb) {
        t[k] = *reinterpret_cast<const f32x4*>(T.a_f32 + b * 256 + lane * 4);
        wv[k] = *reinterpret_cast<const f32x4*>(T.norm_w + b * 256 + lane * 4);
      }
    }
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
    if (tl && tid == 0) tl[4] = wall_clock64();
    if (lane == 0) scratch[wave] = ss;
    __syncthreads();
    const float total = scratch_total<NW>(scratch);
    const float pre = pre_scale_of(T);
    const float scale = pre > 0.f ? pre : 1.0f / sqrtf(total / (float)n + T.eps);
    if (tl && tid == 0) tl[5] = wall_clock64();
#pragma unroll
    for (int k = 0; k < KB1; ++k) {
      const int b = wave + NW * k;
      if (b < nb) {
        float v[4] = {t[k].x * scale * wv[k].x, t[k].y * scale * wv[k].y, t[k].z * scale * wv[k].z, t[k].w * scale * wv[k].w};
        q8k_block_lds<LAY>(v, lane, lds + (size_t)b * 4 * ITEM_LDS);
      }
    }
    return;
  }
  float scale = 1.0f;
  if (T.act_mode == ACT_F32_NORM) {  // long vectors: two passes
    const float total = wg_sumsq<NW>(T.a_f32, n, tid, scratch);
    scale = 1.0f / sqrtf(total / (float)n + T.eps);
  }
  // wave w quantises blocks w, w+NW, ...; the loads of KB blocks are issued together
  constexpr int KB = NW == 4 ? 8 : 5;
  for (int b0 = wave; b0 < nb; b0 += KB * NW) {
    f32x4 t[KB], wv[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int b = b0 + NW * k;
      if (b < nb) {
        t[k] = *reinterpret_cast<const f32x4*>(T.a_f32 + b * 256 + lane * 4);
        if (T.act_mode == ACT_F32_NORM) wv[k] = *reinterpret_cast<const f32x4*>(T.norm_w + b * 256 + lane * 4);
      }
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int b = b0 + NW * k;
      if (b < nb) {
        float v[4] = {t[k].x, t[k].y, t[k].z, t[k].w};
        if (T.act_mode == ACT_F32_NORM) {
          v[0] = v[0] * scale * wv[k].x;
          v[1] = v[1] * scale * wv[k].y;
          v[2] = v[2] * scale * wv[k].z;
          v[3] = v[3] * scale * wv[k].w;
          }
        q8k_block_lds<LAY>(v, lane, lds + (size_t)b * 4 * ITEM_LDS);
      }
    }
  }
}

// Stage an f32 activation vector (F8 / F16 / F32 weights)
template <int NW, typename SRC>
DEV void stage_f32(const SRC& T, float* l_x, int tid, float* scratch) {
  const int n = T.n;
  if (T.act_mode == ACT_F32_NORM) {
    const float total = wg_sumsq<NW>(T.a_f32, n, tid, scratch);
    const float scale = 1.0f / sqrtf(total / (float)n + T.eps);
    for (int i = tid; i < n; i += NW * 64) {
      const float y = T.a_f32[i] * scale * T.norm_w[i];
      l_x[i] = y;
    }
  } else {
    const f32x4* src = reinterpret_cast<const f32x4*>(T.a_f32);
    f32x4* dst = reinterpret_cast<f32x4*>(l_x);
    for (int i = tid; i < (n >> 2); i += NW * 64) dst[i] = src[i];
    for (int i = (n & ~3) + tid; i < n; i += NW * 64) l_x[i] = T.a_f32[i];
  }
}

// parity tap: the staged item records of an n-vector back to the linear Q8_K form (int8 codes, block scales)
template <int LAY>
__device__ __attribute__((noinline)) void dump_staged_q8(const uint8_t* lds, int n, int8_t* qs, float* d, int tid, int nthreads) {
  constexpr bool Q2META = LAY == LAY_Q2;
  if (LAY == LAY_TILE) {
    for (int i = tid; i < (n >> 4); i += nthreads)
      reinterpret_cast<u32x4*>(qs)[i] = *reinterpret_cast<const u32x4*>(lds + (size_t)(i >> 4) * 320 + (i & 15) * 16);
    for (int b = tid; b < (n >> 8); b += nthreads) d[b] = *reinterpret_cast<const float*>(lds + (size_t)b * 320 + 304);
    return;
  }
  for (int i = tid; i < (n >> 4); i += nthreads) {  // 16-byte runs = sub-blocks
    const int b = i >> 4, j = i & 15, h = j >> 3, sidx = (j >> 1) & 3, lh = j & 1;
    const uint8_t* rec = lds + (size_t)(b * 4 + 2 * h + lh) * ITEM_LDS;
    reinterpret_cast<u32x4*>(qs)[i] = *reinterpret_cast<const u32x4*>(rec + sidx * 16);
  }
  for (int b = tid; b < (n >> 8); b += nthreads) {
    const float* m = reinterpret_cast<const float*>(lds + (size_t)b * 4 * ITEM_LDS + 72);
    d[b] = Q2META ? m[1] : m[0];
  }
}

template <int QT>
DEV float fitem(u32x4 w, const float* xa, float partial) {
  if (QT == DSK_QUANT_F32) {
    partial = fmaf(u2f(w.x), xa[0], partial);
    partial = fmaf(u2f(w.y), xa[1], partial);
    partial = fmaf(u2f(w.z), xa[2], partial);
    partial = fmaf(u2f(w.w), xa[3], partial);
  } else if (QT == DSK_QUANT_F16) {
    const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f16x2 p = __builtin_bit_cast(f16x2, ww[k]);
      partial = fmaf((float)p.x, xa[2 * k], partial);
      partial = fmaf((float)p.y, xa[2 * k + 1], partial);
    }
  } else {  // fp8 byte -> f16 is the byte shifted into the high half (src/codec.h:40-48)
    const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, ww[k], 0x010c000cu));
      f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, ww[k], 0x030c020cu));
      partial = fmaf((float)lo.x, xa[4 * k], partial);
      partial = fmaf((float)lo.y, xa[4 * k + 1], partial);
      partial = fmaf((float)hi.x, xa[4 * k + 2], partial);
      partial = fmaf((float)hi.y, xa[4 * k + 3], partial);
    }
  }
  return partial;
}

// resolved per-task weight pointers
struct WPtr {
  const uint8_t *qs, *sc, *hm, *dm, *qs2, *sc2, *hm2, *dm2;
  const float *scale, *scale2;
  bool present;
};
DEV WPtr resolve(const GemvTask& T) {
  WPtr p;
  int le = 0;
  p.present = true;
  if (T.e_qs != 0) {
    // slot -> expert on the device (the reference reads active_experts on the host, src/infer.cpp:854)
    const int e = T.expert_ids ? T.expert_ids[T.slot] : T.slot;
    le = e - T.expert_base;
    p.present = le >= 0 && le < T.local_experts;  // otherwise the expert lives on another GPU
    if (!p.present) le = 0;
  }
  p.qs = T.qs + (size_t)le * T.e_qs;
  p.sc = T.sc ? T.sc + (size_t)le * T.e_sc : nullptr;
  p.hm = T.hm ? T.hm + (size_t)le * T.e_hm : nullptr;
  p.dm = T.dm ? T.dm + (size_t)le * T.e_dm : nullptr;
  p.scale = T.scale ? T.scale + (size_t)le * T.e_scale : nullptr;
  p.qs2 = T.qs2 ? T.qs2 + (size_t)le * T.e_qs : nullptr;
  p.sc2 = T.sc2 ? T.sc2 + (size_t)le * T.e_sc : nullptr;
  p.hm2 = T.hm2 ? T.hm2 + (size_t)le * T.e_hm : nullptr;
  p.dm2 = T.dm2 ? T.dm2 + (size_t)le * T.e_dm : nullptr;
  p.scale2 = T.scale2 ? T.scale2 + (size_t)le * T.e_scale : nullptr;
  return p;
}


// ------------------------------------------------------------------------------------
// One "chunk" = U column steps x R rows of weight data held in registers.  Loading and computing
// are separate so that the first chunk can be requested from HBM BEFORE the workgroup stages its
// activation vector (the prologue then overlaps the memory latency instead of preceding it).
// ------------------------------------------------------------------------------------
// K-quant planes are read through buffer descriptors: address = plane base (SGPRs) + per-lane row
// offset (VGPR, fixed for a row group) + column-step offset (SGPR), so a load costs no VALU at all.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
DEV rsrc_t make_rsrc(const void* p) {
  // the base is wave-uniform by construction (task pointers come from the launch descriptor); saying so
  // keeps the compiler from wrapping every load in a waterfall loop
  const unsigned long long v = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
}
// ... or, for a wave that has nothing to read there, a descriptor of zero records: every load returns zeros without touching memory
DEV rsrc_t make_rsrc_n(const void* p, bool live) {
  const unsigned long long v = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  const u32 nrec = __builtin_amdgcn_readfirstlane(live ? 0xffffffffu : 0u);  // (wave-uniform by construction, like the base)
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)nrec, 0x00020000);
}
#define BUF_NT 2  // streaming data: non-temporal
struct KQRsrc {
  rsrc_t qs, sc, hm, dm, qs2, sc2, hm2, dm2;
};
template <int QT, bool GLU>
DEV KQRsrc kq_rsrc(const WPtr& P) {
  KQRsrc r;
  r.qs = make_rsrc(P.qs); r.sc = make_rsrc(P.sc); r.dm = make_rsrc(P.dm);
  r.hm = make_rsrc(QT == DSK_QUANT_Q3_K ? P.hm : P.qs);
  r.qs2 = make_rsrc(GLU ? P.qs2 : P.qs); r.sc2 = make_rsrc(GLU ? P.sc2 : P.sc); r.dm2 = make_rsrc(GLU ? P.dm2 : P.dm);
  r.hm2 = make_rsrc(GLU && QT == DSK_QUANT_Q3_K ? P.hm2 : P.qs);
  return r;
}

template <int QT, int R, int U, bool GLU>
struct ChunkKQ {
  u32x4 w[U][R], w2[U][R], hmv[U][R], hmv2[U][R];
  u32 scw[U][R], scw2[U][R], dmw[U][R], dmw2[U][R], s1w[U][R], s2w[U][R], s1w2[U][R], s2w2[U][R];
};

// rowblk[r] = row * nb + (sub >> 2): index of the lane's first super-block; q = sub & 3 its quarter
template <int QT, int R, int U, bool GLU>
DEV void load_chunk_kq(ChunkKQ<QT, R, U, GLU>& c, const KQRsrc& B, int its, int items, int sub, int lpr_log2, int q, const int (&rowblk)[R], int it0) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int it = it0 + u;
    if (it >= its) break;  // wave-uniform: the trailing steps of the last chunk do no work at all
    const int sblk = (it << lpr_log2) >> 2;  // super-blocks advanced by this column step (scalar)
    // a ragged LAST step (the lane count does not divide the row's items): lanes past the end re-read the row's last
    // super-block and get a zero scale -- uniform branch, no cost for exact fits
    const bool ragged = (it << lpr_log2) + (1 << lpr_log2) > items;
    const bool dead = ragged && sub + (it << lpr_log2) >= items;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int vb = dead ? rowblk[r] - (sub >> 2) + (items >> 2) - 1 - sblk : rowblk[r];
      c.w[u][r] = __builtin_amdgcn_raw_buffer_load_b128(B.qs, vb * 64 + q * 16, sblk * 64, BUF_NT);
      if (QT == DSK_QUANT_Q2_K) {
        c.scw[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc, vb * 16 + q * 4, sblk * 16, BUF_NT);
        c.dmw[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.dm, vb * 4, sblk * 4, BUF_NT);
      } else {
        c.hmv[u][r] = __builtin_amdgcn_raw_buffer_load_b128(B.hm, vb * 32 + (q & 1) * 16, sblk * 32, BUF_NT);
        c.scw[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc, vb * 12, sblk * 12, BUF_NT);
        c.s1w[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc, vb * 12 + 4, sblk * 12, BUF_NT);
        c.s2w[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc, vb * 12 + 8, sblk * 12, BUF_NT);
        c.dmw[u][r] = (u32)__builtin_amdgcn_raw_buffer_load_b16(B.dm, vb * 2, sblk * 2, BUF_NT);
      }
      if (GLU) {
        c.w2[u][r] = __builtin_amdgcn_raw_buffer_load_b128(B.qs2, vb * 64 + q * 16, sblk * 64, BUF_NT);
        if (QT == DSK_QUANT_Q2_K) {
          c.scw2[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc2, vb * 16 + q * 4, sblk * 16, BUF_NT);
          c.dmw2[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.dm2, vb * 4, sblk * 4, BUF_NT);
        } else {
          c.hmv2[u][r] = __builtin_amdgcn_raw_buffer_load_b128(B.hm2, vb * 32 + (q & 1) * 16, sblk * 32, BUF_NT);
          c.scw2[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc2, vb * 12, sblk * 12, BUF_NT);
          c.s1w2[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc2, vb * 12 + 4, sblk * 12, BUF_NT);
          c.s2w2[u][r] = __builtin_amdgcn_raw_buffer_load_b32(B.sc2, vb * 12 + 8, sblk * 12, BUF_NT);
          c.dmw2[u][r] = (u32)__builtin_amdgcn_raw_buffer_load_b16(B.dm2, vb * 2, sblk * 2, BUF_NT);
        }
      }
      if (ragged && dead) {  // d = dmin = 0: a finite product with 0
        c.dmw[u][r] = 0;
        if (GLU) c.dmw2[u][r] = 0;
      }
    }
  }
}

// lds_lane = staged vector + sub * ITEM_LDS (the lane's record of column step 0)
template <int QT, int R, int U, bool GLU>
DEV void compute_chunk_kq(const ChunkKQ<QT, R, U, GLU>& c, int its, int items, int sub, int lpr_log2, int q, int it0, const uint8_t* lds_lane,
                          float (&acc)[R], float (&acc2)[R]) {
  const int h = q >> 1, lh = q & 1;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (it0 + u >= its) break;
    const int item0 = (it0 + u) << lpr_log2;
    const uint8_t* rec = lds_lane + (size_t)item0 * ITEM_LDS;
    if (item0 + (1 << lpr_log2) > items && sub + item0 >= items)  // ragged last step: a staged (finite) record
      rec = lds_lane + (size_t)(items - 1 - sub) * ITEM_LDS;
    u32x4 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const u32x4*>(rec + s * 16);
    const u32x4 meta = *reinterpret_cast<const u32x4*>(rec + 64);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (QT == DSK_QUANT_Q2_K) {
        acc[r] = q2k_item(c.w[u][r], c.scw[u][r], c.dmw[u][r], a, meta, acc[r]);
        if (GLU) acc2[r] = q2k_item(c.w2[u][r], c.scw2[u][r], c.dmw2[u][r], a, meta, acc2[r]);
      } else {
        acc[r] = q3k_item(c.w[u][r], c.hmv[u][r], c.scw[u][r], c.s1w[u][r], c.s2w[u][r], c.dmw[u][r], h, lh, a, meta, acc[r]);
        if (GLU) acc2[r] = q3k_item(c.w2[u][r], c.hmv2[u][r], c.scw2[u][r], c.s1w2[u][r], c.s2w2[u][r], c.dmw2[u][r], h, lh, a, meta, acc2[r]);
      }
    }
  }
}

// F8E5M2 / F16 / F32 weights; products are f32 FMAs like the reference (src/infer.cpp:289-297);
// the block scale is applied once per 16-byte item (the reference scales every weight before the
// FMA: same value up to one f32 rounding per item).
template <int QT, int R, int U, bool GLU>
struct ChunkF {
  u32x4 w[U][R], w2[U][R];
  float sv[U][R], sv2[U][R];
  int itemv[U];
};
template <int QT>
struct FTraits {
  static constexpr int EPI = QT == DSK_QUANT_F32 ? 4 : (QT == DSK_QUANT_F16 ? 8 : 16);  // elements per 16-byte item
  static constexpr int ESZ = 16 / EPI;
};

template <int QT, int R, int U, bool GLU>
DEV void load_chunk_f(ChunkF<QT, R, U, GLU>& c, const WPtr& P, int n, int b0, int b1, int lpr_log2, int lane, const int (&row)[R], int it0) {
  constexpr int EPI = FTraits<QT>::EPI, ESZ = FTraits<QT>::ESZ;
  const int LPR = 1 << lpr_log2;
  const int sub = lane & (LPR - 1);
  const int items = n / EPI;
  const int its = (items + LPR - 1) >> lpr_log2;
  const size_t row_bytes = (size_t)n * ESZ;
  const int sc_cols = (n + b1 - 1) / b1;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int it = it0 + u;
    if (it >= its) break;
    int item = sub + (it << lpr_log2);
    const bool live = item < items;
    if (!live) item = items - 1;
    c.itemv[u] = item;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      c.w[u][r] = ldg_nt(reinterpret_cast<const u32x4*>(P.qs + (size_t)row[r] * row_bytes + (size_t)item * 16));
      c.sv[u][r] = P.scale ? P.scale[(size_t)(row[r] / b0) * sc_cols + (item * EPI) / b1] : 1.0f;
      c.sv2[u][r] = 0.f;
      if (GLU) {
        c.w2[u][r] = ldg_nt(reinterpret_cast<const u32x4*>(P.qs2 + (size_t)row[r] * row_bytes + (size_t)item * 16));
        c.sv2[u][r] = P.scale2 ? P.scale2[(size_t)(row[r] / b0) * sc_cols + (item * EPI) / b1] : 1.0f;
      }
      if (!live) c.sv[u][r] = c.sv2[u][r] = 0.f;
    }
  }
}

template <int QT, int R, int U, bool GLU>
DEV void compute_chunk_f(const ChunkF<QT, R, U, GLU>& c, int n, int lpr_log2, int it0, const float* l_x, float (&acc)[R], float (&acc2)[R]) {
  constexpr int EPI = FTraits<QT>::EPI;
  const int its = (n / EPI + (1 << lpr_log2) - 1) >> lpr_log2;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (it0 + u >= its) break;
    const int item = c.itemv[u];
    float xa[EPI];
#pragma unroll
    for (int k = 0; k < EPI / 4; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(l_x + item * EPI + 4 * k);
      xa[4 * k] = v.x;
      xa[4 * k + 1] = v.y;
      xa[4 * k + 2] = v.z;
      xa[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] = fmaf(fitem<QT>(c.w[u][r], xa, 0.f), c.sv[u][r], acc[r]);
      if (GLU) acc2[r] = fmaf(fitem<QT>(c.w2[u][r], xa, 0.f), c.sv2[u][r], acc2[r]);
    }
  }
}

#ifndef KQ_PIPELINE
// 0: one chunk: load U steps, multiply them, repeat (the default).  1: two half-chunks alternate -- same registers, half
// the loads in flight per burst: measured SLOWER everywhere (experts phase 17.0 -> 20.0 us, wo 12.0 -> 13.9): at 16
// waves x 128 VGPRs a lane cannot hold two full chunks, and 64 B per lane in flight is below what one CU needs to keep
// its ~24 GB/s share of HBM busy.  2: two full chunks (spills at 16 waves; for 8-wave / 256-VGPR experiments).
#define KQ_PIPELINE 0
#endif
// dot products of R rows (x 64/LPR rows per wave) with a staged activation vector.
// `pre`: the first chunk was already requested by the caller (prefetch across the prologue).
template <int QT, int R, int U, bool GLU>
DEV void rows_dot_kq(const KQRsrc& B, int items, int sub, int lpr_log2, int q, const int (&rowblk)[R], const uint8_t* lds_lane,
                     float (&acc)[R], float (&acc2)[R]) {
  const int its = (items + (1 << lpr_log2) - 1) >> lpr_log2;
  // (requesting the first weight chunk before the staging prologue was measured and is slower: loads
  // return in order, so the prologue's L2 reads queue behind the HBM reads, and the chunk costs registers)
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = acc2[r] = 0.f;
  if constexpr (KQ_PIPELINE && (U >= 4 || (U >= 2 && QT == DSK_QUANT_Q2_K))) {
    // Two half-chunks in flight alternately: while one is multiplied the other's loads are outstanding, so a wave's
    // VALU work (~half of the time budget of a streamed item) overlaps its own memory traffic instead of alternating
    // with it (the waves of a workgroup start in lock step: without this every wave loads, then every wave computes).
    // Column steps are consumed in order, exactly as below: same sums, same bits.
    constexpr int H = KQ_PIPELINE == 2 ? U : U / 2;  // 2: two FULL chunks alternate (twice the loads in flight)
    ChunkKQ<QT, R, H, GLU> ca, cb;
    load_chunk_kq<QT, R, H, GLU>(ca, B, its, items, sub, lpr_log2, q, rowblk, 0);
    if (H < its) load_chunk_kq<QT, R, H, GLU>(cb, B, its, items, sub, lpr_log2, q, rowblk, H);
    for (int it0 = 0; it0 < its; it0 += 2 * H) {
      compute_chunk_kq<QT, R, H, GLU>(ca, its, items, sub, lpr_log2, q, it0, lds_lane, acc, acc2);
      if (it0 + 2 * H < its) load_chunk_kq<QT, R, H, GLU>(ca, B, its, items, sub, lpr_log2, q, rowblk, it0 + 2 * H);
      if (it0 + H < its) {
        compute_chunk_kq<QT, R, H, GLU>(cb, its, items, sub, lpr_log2, q, it0 + H, lds_lane, acc, acc2);
        if (it0 + 3 * H < its) load_chunk_kq<QT, R, H, GLU>(cb, B, its, items, sub, lpr_log2, q, rowblk, it0 + 3 * H);
      }
    }
  } else {
    ChunkKQ<QT, R, U, GLU> c;
    // (a short first chunk for every other wave quartet, to break the chip-wide load / multiply lock step, was measured:
    // experts phase 38.2 -> 39.1 us, dense w1/w3 24.7 -> 23.2 us, the extra code path cost every small launch ~0.3 us)
    for (int it0 = 0; it0 < its; it0 += U) {
      load_chunk_kq<QT, R, U, GLU>(c, B, its, items, sub, lpr_log2, q, rowblk, it0);
      compute_chunk_kq<QT, R, U, GLU>(c, its, items, sub, lpr_log2, q, it0, lds_lane, acc, acc2);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    acc[r] = lanes_sum(acc[r], lpr_log2);
    if (GLU) acc2[r] = lanes_sum(acc2[r], lpr_log2);
  }
}

// The same dot products for a row length known at COMPILE time (ITS column steps of 2^LL lanes, exact fit): two
// 2-step buffers alternate and every bound folds away, so the code is straight-line and hipcc's waitcnt pass counts the
// loads exactly (vmcnt(N) ladders instead of vmcnt(0)): one buffer is multiplied while the other's loads are in flight.
// Column steps are consumed in order: same sums, same bits as rows_dot_kq.
// (A rolling window at single-step granularity - request DEPTH = 4 steps, multiply step u, request step u + 4 in its
// place - was measured too: slower everywhere (dense w1/w3 21.4 -> 25.3 us, the router launch 10.7 -> 13.1, experts 35.8
// -> 36.6).  Two-step buffers it is.)
template <int QT, int R, bool GLU, int ITS, int LL>
DEV void rows_dot_kq_exact(const KQRsrc& B, int sub, int q, const int (&rowblk)[R], const uint8_t* lds_lane, float (&acc)[R], float (&acc2)[R]) {
  constexpr int ITEMS = ITS << LL;
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = acc2[r] = 0.f;
  ChunkKQ<QT, R, 2, GLU> ca, cb;
  load_chunk_kq<QT, R, 2, GLU>(ca, B, ITS, ITEMS, sub, LL, q, rowblk, 0);
  if (ITS > 2) load_chunk_kq<QT, R, 2, GLU>(cb, B, ITS, ITEMS, sub, LL, q, rowblk, 2);
#pragma unroll
  for (int it0 = 0; it0 < ITS; it0 += 4) {
    compute_chunk_kq<QT, R, 2, GLU>(ca, ITS, ITEMS, sub, LL, q, it0, lds_lane, acc, acc2);
    if (it0 + 4 < ITS) load_chunk_kq<QT, R, 2, GLU>(ca, B, ITS, ITEMS, sub, LL, q, rowblk, it0 + 4);
    if (it0 + 2 < ITS) {
      compute_chunk_kq<QT, R, 2, GLU>(cb, ITS, ITEMS, sub, LL, q, it0 + 2, lds_lane, acc, acc2);
      if (it0 + 6 < ITS) load_chunk_kq<QT, R, 2, GLU>(cb, B, ITS, ITEMS, sub, LL, q, rowblk, it0 + 6);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    acc[r] = lanes_sum(acc[r], LL);
    if (GLU) acc2[r] = lanes_sum(acc2[r], LL);
  }
}

// tail prefetch workgroup (MoeFfnArgs::pf_wgs): plain cacheable 16-byte loads over up to six ranges, results discarded
DEV void tail_prefetch(const void* const (&p)[6], const int (&n)[6], int tid, int nthreads) {
  u32 acc = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const uint8_t* b = static_cast<const uint8_t*>(p[r]);
    if (!b) continue;
    for (int off = tid * 16; off + 16 <= n[r]; off += nthreads * 16) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(b + off);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  asm volatile("" ::"v"(acc));
}

template <int QT, int R, int U, bool GLU>
DEV void rows_dot_f(const WPtr& P, int n, int b0, int b1, int lpr_log2, int lane, const int (&row)[R], const uint8_t* lds,
                    float (&acc)[R], float (&acc2)[R]) {
  ChunkF<QT, R, U, GLU> c;
  const int items = n / FTraits<QT>::EPI;
  const int its = (items + (1 << lpr_log2) - 1) >> lpr_log2;
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = acc2[r] = 0.f;
  for (int it0 = 0; it0 < its; it0 += U) {
    load_chunk_f<QT, R, U, GLU>(c, P, n, b0, b1, lpr_log2, lane, row, it0);
    compute_chunk_f<QT, R, U, GLU>(c, n, lpr_log2, it0, reinterpret_cast<const float*>(lds), acc, acc2);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    acc[r] = lanes_sum(acc[r], lpr_log2);
    if (GLU) acc2[r] = lanes_sum(acc2[r], lpr_log2);
  }
}

