// kernels_gemv.hip -- batch-1 weight-streaming GEMV kernels for gfx950 (wave64).
//
// Replaces the reference's five _matmul overloads and matmul_expert
// (src/infer.cpp:121-379, 423-469) and the vec_dot kernels (src/quant.cpp:434-614, 666-783).
//
// K-quants are W2A8 / W3A8 exactly like the reference: activations arrive as Q8_K
// (int8 + per-256 scale + per-16 sums, src/quant.cpp:616-653), the sub-block dot products
// are integer (v_dot4_i32_i8) and only the per-super-block scaling is float.
//
// Work decomposition (HBM-bound, no MFMA -- one token, so there is no N dimension):
//   * the unit of work ("item") is 16 contiguous bytes of the qs plane = 64 weights
//     (a quarter of a super-block): one global_load_dwordx4 per lane, adjacent lanes read
//     adjacent 16 B, so every load instruction covers whole 128-B lines;
//   * LPR lanes cooperate on one row (LPR = largest power of two dividing the row's item
//     count), a wave works on 64/LPR rows at a time and on R such row groups back to back
//     with the Q8 activations of the current column item held in registers;
//   * the Q8 activation vector is staged once per workgroup in LDS (n + 36*n/256 bytes);
//   * partial sums are combined with wave shuffles; no atomics, fixed order => deterministic.
#include "dsk_internal.h"

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__

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
// Q2_K: one item = (block b, quarter q = 2*h + lh): bytes qs[32*h + 16*lh .. +15].
// Word k of the item, shifted by 2*s, holds elements 128*h + 32*s + 16*lh + 4*k .. +3, i.e.
// sub-block j = 8*h + 2*s + lh (layout: dequantize_row_q2_K, src/quant.cpp:217-247).
// Scalar spec being computed: src/quant.cpp:746-780.
// ------------------------------------------------------------------------------------
DEV float q2k_item(u32x4 w, u32 scw, u32 dm, const u32x4 (&a)[4], u32x2 bsp, float dx, float acc) {
  // masks keep the 2-bit fields in place (x1, x4, x16, x16): values stay < 128 for the signed dot
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
  const int m0 = (scw >> 4) & 0xF, m1 = (scw >> 12) & 0xF, m2 = (scw >> 20) & 0xF, m3 = scw >> 28;
  // exact: x1 is a multiple of 4, x2/x3 of 16
  const int isum = d0 * x0 + ((d1 * x1) >> 2) + ((d2 * x2 + d3 * x3) >> 4);
  const int b0 = (int)(short)(bsp.x & 0xffff), b1 = (int)bsp.x >> 16;
  const int b2 = (int)(short)(bsp.y & 0xffff), b3 = (int)bsp.y >> 16;
  const int summs = m0 * b0 + m1 * b1 + m2 * b2 + m3 * b3;
  const float dall = dx * h2f(dm & 0xffff);
  const float dmin = dx * h2f(dm >> 16);
  acc = fmaf(dall, (float)isum, acc);
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
                   u32x2 bsp, float dx, float acc) {
  const int bsv[4] = {(int)(short)(bsp.x & 0xffff), (int)bsp.x >> 16, (int)(short)(bsp.y & 0xffff), (int)bsp.y >> 16};
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
    total += q3k_scale(s0, s1, s2, 8 * h + 2 * s + lh) * x;
  }
  return fmaf(dx * h2f(d16), (float)total, acc);
}

// ------------------------------------------------------------------------------------
// K-quant GEMV kernel
// ------------------------------------------------------------------------------------
template <int QT, int R, bool GLU>
__global__ __launch_bounds__(256) void gemv_kq_kernel(GemvSeg sg, int lpr_log2) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int LPR = 1 << lpr_log2, RPW = 64 >> lpr_log2;
  const int tid = threadIdx.x;
  const int slot = blockIdx.y;
  const int n = sg.n, nb = n >> 8;

  // slot -> expert (on-device routing: the reference reads active_experts on the host,
  // src/infer.cpp:854; here the ids never leave HBM)
  int le = 0;
  if (sg.e_qs != 0) {
    const int e = sg.expert_ids ? sg.expert_ids[slot] : slot;
    le = e - sg.expert_base;
    if (le < 0 || le >= sg.local_experts) return;  // expert lives on another GPU
  }
  const uint8_t* QS = sg.qs + (size_t)le * sg.e_qs;
  const uint8_t* SC = sg.sc + (size_t)le * sg.e_sc;
  const uint8_t* DM = sg.dm + (size_t)le * sg.e_dm;
  const uint8_t* HM = QT == DSK_QUANT_Q3_K ? sg.hm + (size_t)le * sg.e_hm : nullptr;
  const uint8_t *QS2 = nullptr, *SC2 = nullptr, *DM2 = nullptr, *HM2 = nullptr;
  if (GLU) {
    QS2 = sg.qs2 + (size_t)le * sg.e_qs;
    SC2 = sg.sc2 + (size_t)le * sg.e_sc;
    DM2 = sg.dm2 + (size_t)le * sg.e_dm;
    if (QT == DSK_QUANT_Q3_K) HM2 = sg.hm2 + (size_t)le * sg.e_hm;
  }

  // ---- stage the Q8 activation vector in LDS: qs | bsums (quarter order) | d ----
  uint8_t* l_qs = smem;
  short* l_bs = reinterpret_cast<short*>(smem + n);
  float* l_d = reinterpret_cast<float*>(smem + n + nb * 32);
  {
    const size_t aoff = (size_t)slot * sg.a_slot_stride;
    const u32x4* src = reinterpret_cast<const u32x4*>(sg.a_qs + aoff);
    u32x4* dst = reinterpret_cast<u32x4*>(l_qs);
    for (int i = tid; i < (n >> 4); i += 256) dst[i] = src[i];
    const short* bsrc = sg.a_bsums + (aoff >> 4);
    for (int i = tid; i < nb * 16; i += 256) {
      const int b = i >> 4, j = i & 15;
      const int h = j >> 3, s = (j >> 1) & 3, lh = j & 1;
      l_bs[b * 16 + (2 * h + lh) * 4 + s] = bsrc[i];
    }
    const float* dsrc = sg.a_d + (aoff >> 8);
    for (int i = tid; i < nb; i += 256) l_d[i] = dsrc[i];
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int sub = lane & (LPR - 1), rloc = lane >> lpr_log2;
  const int row0 = (blockIdx.x * 4 + wave) * (RPW * R);
  if (row0 >= sg.rows) return;
  const int its = (nb * 4) >> lpr_log2;

  size_t roff[R];
  bool valid[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int row = row0 + r * RPW + rloc;
    valid[r] = row < sg.rows;
    roff[r] = (size_t)(valid[r] ? row : sg.rows - 1) * nb;
  }
  float acc[R], acc2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = acc2[r] = 0.f;

  for (int it = 0; it < its; ++it) {
    const int item = sub + it * LPR;
    const int b = item >> 2, q = item & 3, h = q >> 1, lh = q & 1;
    // weights first: independent of LDS, keeps HBM requests in flight as early as possible
    u32x4 w[R], w2[R], hmv[R], hmv2[R];
    u32 scw[R], scw2[R], dmw[R], dmw2[R], s1w[R], s2w[R], s1w2[R], s2w2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      w[r] = ldg_nt(reinterpret_cast<const u32x4*>(QS + roff[r] * 64 + (size_t)item * 16));
      if (QT == DSK_QUANT_Q2_K) {
        scw[r] = ldg_nt(reinterpret_cast<const u32*>(SC + roff[r] * 16 + (size_t)item * 4));
        dmw[r] = ldg_nt(reinterpret_cast<const u32*>(DM + (roff[r] + b) * 4));
      } else {
        hmv[r] = ldg_nt(reinterpret_cast<const u32x4*>(HM + (roff[r] + b) * 32 + lh * 16));
        const u32* sp = reinterpret_cast<const u32*>(SC + (roff[r] + b) * 12);
        scw[r] = ldg_nt(sp);
        s1w[r] = ldg_nt(sp + 1);
        s2w[r] = ldg_nt(sp + 2);
        dmw[r] = ldg_nt(reinterpret_cast<const unsigned short*>(DM + (roff[r] + b) * 2));
      }
      if (GLU) {
        w2[r] = ldg_nt(reinterpret_cast<const u32x4*>(QS2 + roff[r] * 64 + (size_t)item * 16));
        if (QT == DSK_QUANT_Q2_K) {
          scw2[r] = ldg_nt(reinterpret_cast<const u32*>(SC2 + roff[r] * 16 + (size_t)item * 4));
          dmw2[r] = ldg_nt(reinterpret_cast<const u32*>(DM2 + (roff[r] + b) * 4));
        } else {
          hmv2[r] = ldg_nt(reinterpret_cast<const u32x4*>(HM2 + (roff[r] + b) * 32 + lh * 16));
          const u32* sp = reinterpret_cast<const u32*>(SC2 + (roff[r] + b) * 12);
          scw2[r] = ldg_nt(sp);
          s1w2[r] = ldg_nt(sp + 1);
          s2w2[r] = ldg_nt(sp + 2);
          dmw2[r] = ldg_nt(reinterpret_cast<const unsigned short*>(DM2 + (roff[r] + b) * 2));
        }
      }
    }
    // activations of this column item (shared by all R rows and both GLU matrices)
    u32x4 a[4];
    const uint8_t* ap = l_qs + b * 256 + h * 128 + lh * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const u32x4*>(ap + s * 32);
    const u32x2 bsp = *reinterpret_cast<const u32x2*>(l_bs + b * 16 + q * 4);
    const float dx = l_d[b];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (QT == DSK_QUANT_Q2_K) {
        acc[r] = q2k_item(w[r], scw[r], dmw[r], a, bsp, dx, acc[r]);
        if (GLU) acc2[r] = q2k_item(w2[r], scw2[r], dmw2[r], a, bsp, dx, acc2[r]);
      } else {
        acc[r] = q3k_item(w[r], hmv[r], scw[r], s1w[r], s2w[r], dmw[r], h, lh, a, bsp, dx, acc[r]);
        if (GLU) acc2[r] = q3k_item(w2[r], hmv2[r], scw2[r], s1w2[r], s2w2[r], dmw2[r], h, lh, a, bsp, dx, acc2[r]);
      }
    }
  }

  // ---- combine the LPR lanes of each row (fixed butterfly order) ----
#pragma unroll
  for (int r = 0; r < R; ++r) {
    for (int off = LPR >> 1; off >= 1; off >>= 1) {
      acc[r] += __shfl_xor(acc[r], off);
      if (GLU) acc2[r] += __shfl_xor(acc2[r], off);
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!valid[r]) continue;
      const int row = row0 + r * RPW + rloc;
      float* o = sg.out + (size_t)slot * sg.out_slot_stride + row;
      if (GLU) *o = act_fn(acc[r], sg.act) * acc2[r];   // src/infer.cpp:859-872
      else if (sg.epilogue == EPI_ADD) *o += acc[r];    // residual add, src/infer.cpp:832-834,928-930
      else *o = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------
// F8E5M2 / F16 / F32 weights, f32 activations staged in LDS.
// item = 16 bytes of one row = 16 / 8 / 4 weights.  fp8 byte -> f16 is the byte shifted into
// the high half (src/codec.h:40-48), f16 -> f32 is exact; products are f32 FMAs like the
// reference (src/infer.cpp:289-297); the block scale is applied once per item (the reference
// scales every weight before the FMA: same value up to one f32 rounding per item).
// ------------------------------------------------------------------------------------
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
  } else {
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

template <int QT, int R, bool GLU>
__global__ __launch_bounds__(256) void gemv_f_kernel(GemvSeg sg, int lpr_log2) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int LPR = 1 << lpr_log2, RPW = 64 >> lpr_log2;
  constexpr int EPI = QT == DSK_QUANT_F32 ? 4 : (QT == DSK_QUANT_F16 ? 8 : 16);  // elements per item
  constexpr int ESZ = 16 / EPI;
  const int tid = threadIdx.x, slot = blockIdx.y, n = sg.n;
  int le = 0;
  if (sg.e_qs != 0) {
    const int e = sg.expert_ids ? sg.expert_ids[slot] : slot;
    le = e - sg.expert_base;
    if (le < 0 || le >= sg.local_experts) return;
  }
  const uint8_t* W = sg.qs + (size_t)le * sg.e_qs;
  const uint8_t* W2 = GLU ? sg.qs2 + (size_t)le * sg.e_qs : nullptr;
  const float* S = sg.scale ? sg.scale + (size_t)le * sg.e_scale : nullptr;
  const float* S2 = (GLU && sg.scale2) ? sg.scale2 + (size_t)le * sg.e_scale : nullptr;

  float* l_x = reinterpret_cast<float*>(smem);
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(sg.a_f32 + (size_t)slot * sg.a_slot_stride);
    u32x4* dst = reinterpret_cast<u32x4*>(l_x);
    for (int i = tid; i < (n >> 2); i += 256) dst[i] = src[i];
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int sub = lane & (LPR - 1), rloc = lane >> lpr_log2;
  const int row0 = (blockIdx.x * 4 + wave) * (RPW * R);
  if (row0 >= sg.rows) return;
  const int items = n / EPI;
  const int its = items >> lpr_log2;
  const size_t row_bytes = (size_t)n * ESZ;

  int rowi[R];
  bool valid[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int row = row0 + r * RPW + rloc;
    valid[r] = row < sg.rows;
    rowi[r] = valid[r] ? row : sg.rows - 1;
  }
  float acc[R], acc2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = acc2[r] = 0.f;

  for (int it = 0; it < its; ++it) {
    const int item = sub + it * LPR;
    u32x4 w[R], w2[R];
    float sv[R], sv2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      w[r] = ldg_nt(reinterpret_cast<const u32x4*>(W + (size_t)rowi[r] * row_bytes + (size_t)item * 16));
      sv[r] = S ? S[(size_t)(rowi[r] / sg.b0) * sg.sc_cols + (item * EPI) / sg.b1] : 1.0f;
      if (GLU) {
        w2[r] = ldg_nt(reinterpret_cast<const u32x4*>(W2 + (size_t)rowi[r] * row_bytes + (size_t)item * 16));
        sv2[r] = S2 ? S2[(size_t)(rowi[r] / sg.b0) * sg.sc_cols + (item * EPI) / sg.b1] : 1.0f;
      }
    }
    float xa[EPI];
#pragma unroll
    for (int k = 0; k < EPI / 4; ++k) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(l_x + item * EPI + 4 * k);
      xa[4 * k] = u2f(v.x);
      xa[4 * k + 1] = u2f(v.y);
      xa[4 * k + 2] = u2f(v.z);
      xa[4 * k + 3] = u2f(v.w);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] = fmaf(fitem<QT>(w[r], xa, 0.f), sv[r], acc[r]);
      if (GLU) acc2[r] = fmaf(fitem<QT>(w2[r], xa, 0.f), sv2[r], acc2[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    for (int off = LPR >> 1; off >= 1; off >>= 1) {
      acc[r] += __shfl_xor(acc[r], off);
      if (GLU) acc2[r] += __shfl_xor(acc2[r], off);
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!valid[r]) continue;
      const int row = row0 + r * RPW + rloc;
      float* o = sg.out + (size_t)slot * sg.out_slot_stride + row;
      if (GLU) *o = act_fn(acc[r], sg.act) * acc2[r];
      else if (sg.epilogue == EPI_ADD) *o += acc[r];
      else *o = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------
static int pick_lpr(int items) {
  int lpr = 64;
  while (lpr > 1 && (items % lpr) != 0) lpr >>= 1;
  return lpr;
}
static int pick_r(long rows, int lpr, int n_slots) {
  const long groups = (rows + (64 / lpr) - 1) / (64 / lpr);  // waves at R = 1, per slot
  const long waves1 = groups * (n_slots > 0 ? n_slots : 1);
  if (waves1 / 4 >= 2048) return 4;
  if (waves1 / 2 >= 2048) return 2;
  return 1;
}

template <int QT, int R>
static int launch_kq(hipStream_t st, const GemvSeg& sg, int lpr_log2) {
  const int rpw = 64 >> lpr_log2;
  const int groups = (sg.rows + rpw * R - 1) / (rpw * R);
  dim3 grid((groups + 3) / 4, sg.n_slots > 0 ? sg.n_slots : 1);
  const size_t lds = (size_t)sg.n + (size_t)(sg.n / 256) * 36;
  if (sg.epilogue == EPI_GLU)
    hipLaunchKernelGGL((gemv_kq_kernel<QT, R, true>), grid, dim3(256), lds, st, sg, lpr_log2);
  else
    hipLaunchKernelGGL((gemv_kq_kernel<QT, R, false>), grid, dim3(256), lds, st, sg, lpr_log2);
  return DSK_OK;
}
template <int QT>
static int launch_kq_r(hipStream_t st, const GemvSeg& sg, int lpr_log2, int r) {
  switch (r) {
    case 4: return launch_kq<QT, 4>(st, sg, lpr_log2);
    case 2: return launch_kq<QT, 2>(st, sg, lpr_log2);
    default: return launch_kq<QT, 1>(st, sg, lpr_log2);
  }
}

template <int QT, int R, bool GLU>
static int launch_f2(hipStream_t st, const GemvSeg& sg, int lpr_log2) {
  const int rpw = 64 >> lpr_log2;
  const int groups = (sg.rows + rpw * R - 1) / (rpw * R);
  dim3 grid((groups + 3) / 4, sg.n_slots > 0 ? sg.n_slots : 1);
  const size_t lds = (size_t)sg.n * 4;
  auto k = gemv_f_kernel<QT, R, GLU>;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, sg, lpr_log2);
  return DSK_OK;
}
template <int QT>
static int launch_f_r(hipStream_t st, const GemvSeg& sg, int lpr_log2, int r) {
  const bool glu = sg.epilogue == EPI_GLU;
  switch (r) {
    case 4: return glu ? launch_f2<QT, 4, true>(st, sg, lpr_log2) : launch_f2<QT, 4, false>(st, sg, lpr_log2);
    case 2: return glu ? launch_f2<QT, 2, true>(st, sg, lpr_log2) : launch_f2<QT, 2, false>(st, sg, lpr_log2);
    default: return glu ? launch_f2<QT, 1, true>(st, sg, lpr_log2) : launch_f2<QT, 1, false>(st, sg, lpr_log2);
  }
}
static int ilog2(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}

int launch_gemv(hipStream_t st, int quant, const GemvSeg& sg) {
  if (sg.rows <= 0 || sg.n <= 0) DSK_FAIL(DSK_ERR_INVALID, "gemv: empty shape %d x %d", sg.rows, sg.n);
  if (quant == DSK_QUANT_Q2_K || quant == DSK_QUANT_Q3_K) {
    if (sg.n % QK_K) DSK_FAIL(DSK_ERR_INVALID, "k-quant gemv: n=%d is not a multiple of 256 (src/quantizer.cpp:8)", sg.n);
    const int items = sg.n / 64;
    const int lpr = pick_lpr(items);  // items is a multiple of 4
    const int r = pick_r(sg.rows, lpr, sg.n_slots);
    if (quant == DSK_QUANT_Q2_K) return launch_kq_r<DSK_QUANT_Q2_K>(st, sg, ilog2(lpr), r);
    return launch_kq_r<DSK_QUANT_Q3_K>(st, sg, ilog2(lpr), r);
  }
  const int epi = quant == DSK_QUANT_F32 ? 4 : (quant == DSK_QUANT_F16 ? 8 : 16);
  if (sg.n % epi) DSK_FAIL(DSK_ERR_INVALID, "gemv: n=%d is not a multiple of %d (src/infer.cpp:169,246)", sg.n, epi);
  if ((size_t)sg.n * 4 > 160 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "gemv: n=%d does not fit LDS", sg.n);
  const int lpr = pick_lpr(sg.n / epi);
  const int r = pick_r(sg.rows, lpr, sg.n_slots);
  switch (quant) {
    case DSK_QUANT_F32: return launch_f_r<DSK_QUANT_F32>(st, sg, ilog2(lpr), r);
    case DSK_QUANT_F16: return launch_f_r<DSK_QUANT_F16>(st, sg, ilog2(lpr), r);
    case DSK_QUANT_F8E5M2: return launch_f_r<DSK_QUANT_F8E5M2>(st, sg, ilog2(lpr), r);
  }
  DSK_FAIL(DSK_ERR_INVALID, "gemv: bad quant %d", quant);
}
