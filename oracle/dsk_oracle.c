/*
 * dsk_oracle.c -- CPU oracle (plain C).  TEST INFRASTRUCTURE ONLY: see dsk_oracle.h.
 *
 * Restates, from the behaviour of andrewkchan/deepseek.cpp (reference paths below are
 * relative to /root/reference), the arithmetic of one decoded token.  Where the reference
 * binary's float association is fixed by explicit AVX2 intrinsics (K-quant, F16 and F8
 * GEMV) this file reproduces that association with scalar "virtual lanes" so that it is
 * bit-identical to the reference build; where the reference leaves the order to the
 * compiler (-O3 -ffast-math: rmsnorm, softmax, attention) it uses the source order and
 * parity with the reference is by tolerance.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (see oracle/Makefile) -- fused
 * multiply-adds appear only where written as fmaf().
 */
#include "dsk_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define QK_K 256
#define KV_SINKS 2 /* src/model.h:14 */

static char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int fail(const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return -1;
}

/* ------------------------------------------------------------------------- */
/* scalar codecs: src/codec.h:22-57 (F16C _cvtsh_ss / _cvtss_sh(x,0) semantics) */
/* ------------------------------------------------------------------------- */
float orc_half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t orc_float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) { /* inf / nan */
    uint32_t man = absx & 0x7fffffu;
    return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0));
  }
  if (absx >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (absx < 0x38800000u) { /* result is subnormal or zero (|f| < 2^-14) */
    if (absx < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (ties-to-even at exactly 2^-25 -> 0) */
    int e = (int)(absx >> 23);                    /* biased exponent, 102..112 */
    uint32_t man = (absx & 0x7fffffu) | 0x800000u;
    int shift = 126 - e; /* bits to drop so that result unit = 2^-24 */
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half & 1))) half++;
    return (uint16_t)(sign | half);
  }
  /* normal */
  uint32_t e = (absx >> 23) - 127 + 15;
  uint32_t man = absx & 0x7fffffu;
  uint32_t half = (e << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++;
  return (uint16_t)(sign | half);
}

float orc_f8e5m2_to_float(uint8_t b) { /* src/codec.h:40-48: byte is the upper byte of an f16 */
  return orc_half_to_float((uint16_t)((uint16_t)b << 8));
}
uint8_t orc_float_to_f8e5m2(float f) { /* src/codec.h:49-57: truncate */
  return (uint8_t)(orc_float_to_half(f) >> 8);
}

/* ------------------------------------------------------------------------- */
/* Q8_K activation quantisation: src/quant.cpp:616-653, nearest_int src/quant.cpp:34-39 */
/* ------------------------------------------------------------------------- */
static inline int nearest_int(float fval) {
  float val = fval + 12582912.f;
  int i;
  memcpy(&i, &val, sizeof(int));
  return (i & 0x007fffff) - 0x00400000;
}

void orc_q8k_quantize(const float* x, int n, int8_t* qs, float* d, int16_t* bsums) {
  int nb = n / QK_K;
  for (int i = 0; i < nb; i++) {
    float max = 0, amax = 0;
    for (int j = 0; j < QK_K; ++j) {
      float ax = fabsf(x[j]);
      if (ax > amax) {
        amax = ax;
        max = x[j];
      }
    }
    if (!amax) { /* reference leaves bsums untouched here; its buffer is zero-initialised */
      d[i] = 0;
      memset(qs, 0, QK_K);
      memset(bsums, 0, sizeof(int16_t) * (QK_K / 16));
    } else {
      const float iscale = -127.f / max;
      for (int j = 0; j < QK_K; ++j) {
        int v = nearest_int(iscale * x[j]);
        qs[j] = (int8_t)(v < 127 ? v : 127);
      }
      for (int j = 0; j < QK_K / 16; ++j) {
        int sum = 0;
        for (int ii = 0; ii < 16; ++ii) sum += qs[j * 16 + ii];
        bsums[j] = (int16_t)sum;
      }
      /* Source says `1/iscale` (src/quant.cpp:650); under the reference's -ffast-math build
       * (Makefile:31) gcc folds 1/(-127/max) to max * (1/-127), which is what the reference
       * binary stores (checked bit-for-bit against oracle/_ref in tests/test_oracle_pin.py). */
      d[i] = max * (1.0f / -127.f);
    }
    x += QK_K;
    qs += QK_K;
    bsums += QK_K / 16;
  }
}

/* ------------------------------------------------------------------------- */
/* K-quant blocks: src/quant.h:41-52 (q2_K, 84 B), src/quant.h:70-76 (q3_K, 110 B) */
/* ------------------------------------------------------------------------- */
#define Q2K_BYTES 84
#define Q3K_BYTES 110

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* dequantize_row_q2_K: src/quant.cpp:217-247 */
static void dequant_row_q2k(const uint8_t* row, int n, float* y) {
  int nb = n / QK_K;
  for (int i = 0; i < nb; i++) {
    const uint8_t* blk = row + (size_t)i * Q2K_BYTES;
    const uint8_t* scales = blk;
    const uint8_t* q = blk + 16;
    const float d = orc_half_to_float(rd16(blk + 80));
    const float min = orc_half_to_float(rd16(blk + 82));
    int is = 0;
    for (int nn = 0; nn < QK_K; nn += 128) {
      int shift = 0;
      for (int j = 0; j < 4; ++j) {
        uint8_t sc = scales[is++];
        float dl = d * (sc & 0xF), ml = min * (sc >> 4);
        for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l] >> shift) & 3)) - ml;
        sc = scales[is++];
        dl = d * (sc & 0xF);
        ml = min * (sc >> 4);
        for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l + 16] >> shift) & 3)) - ml;
        shift += 2;
      }
      q += 32;
    }
  }
}

/* 6-bit scale unpack: src/quant.cpp:402-407 / 592-597 */
static void q3k_scales(const uint8_t* packed12, int8_t out[16]) {
  const uint32_t kmask1 = 0x03030303, kmask2 = 0x0f0f0f0f;
  uint32_t aux[4];
  memcpy(aux, packed12, 12);
  uint32_t tmp = aux[2];
  aux[2] = ((aux[0] >> 4) & kmask2) | (((tmp >> 4) & kmask1) << 4);
  aux[3] = ((aux[1] >> 4) & kmask2) | (((tmp >> 6) & kmask1) << 4);
  aux[0] = (aux[0] & kmask2) | (((tmp >> 0) & kmask1) << 4);
  aux[1] = (aux[1] & kmask2) | (((tmp >> 2) & kmask1) << 4);
  memcpy(out, aux, 16);
}

/* dequantize_row_q3_K: src/quant.cpp:384-432 */
static void dequant_row_q3k(const uint8_t* row, int n, float* y) {
  int nb = n / QK_K;
  for (int i = 0; i < nb; i++) {
    const uint8_t* blk = row + (size_t)i * Q3K_BYTES;
    const uint8_t* hm = blk;
    const uint8_t* q = blk + 32;
    int8_t scales[16];
    q3k_scales(blk + 96, scales);
    const float d_all = orc_half_to_float(rd16(blk + 108));
    uint8_t m = 1;
    int is = 0;
    for (int nn = 0; nn < QK_K; nn += 128) {
      int shift = 0;
      for (int j = 0; j < 4; ++j) {
        float dl = d_all * (scales[is++] - 32);
        for (int l = 0; l < 16; ++l)
          *y++ = dl * ((int8_t)((q[l + 0] >> shift) & 3) - ((hm[l + 0] & m) ? 0 : 4));
        dl = d_all * (scales[is++] - 32);
        for (int l = 0; l < 16; ++l)
          *y++ = dl * ((int8_t)((q[l + 16] >> shift) & 3) - ((hm[l + 16] & m) ? 0 : 4));
        shift += 2;
        m <<= 1;
      }
      q += 32;
    }
  }
}

void orc_dequant_row(int quant, const void* row, int n, float* y) {
  if (quant == DSK_QUANT_Q2_K) dequant_row_q2k((const uint8_t*)row, n, y);
  else if (quant == DSK_QUANT_Q3_K) dequant_row_q3k((const uint8_t*)row, n, y);
}

/* hsum_float_8: src/quant.cpp:46-52 */
static inline float hsum8(const float a[8]) {
  float r0 = a[4] + a[0], r1 = a[5] + a[1], r2 = a[6] + a[2], r3 = a[7] + a[3];
  return (r0 + r2) + (r1 + r3);
}

/*
 * ggml_vec_dot_q2_K_q8_K, AVX2 branch: src/quant.cpp:678-742 (the scalar spec is :743-781).
 * Integer part: lane m (0..7) of `sumi` collects, over both 128-halves and the four 2-bit
 * shifts, scale * sum of the 4 byte products at byte positions 4m..4m+3 of the 32-byte
 * group (madd_epi16 after maddubs_epi16); `prod` lane l = mins[2l]*bsums[2l] +
 * mins[2l+1]*bsums[2l+1] (:697).  Float part per block: acc = fma(dmin, prod, acc) (:699)
 * then acc = fma(d, sumi, acc) (:738); finally hsum_float_8 (:742).
 */
static float vec_dot_q2k(int n, const uint8_t* wrow, const int8_t* q8, const float* yd,
                         const int16_t* bsums) {
  int nb = n / QK_K;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nb; ++i) {
    const uint8_t* blk = wrow + (size_t)i * Q2K_BYTES;
    const uint8_t* sc = blk;
    const uint8_t* q2 = blk + 16;
    const int8_t* a = q8 + (size_t)i * QK_K;
    const int16_t* bs = bsums + (size_t)i * 16;
    const float d = yd[i] * orc_half_to_float(rd16(blk + 80));
    const float dmin = -yd[i] * orc_half_to_float(rd16(blk + 82));
    for (int l = 0; l < 8; ++l) {
      int prod = (sc[2 * l] >> 4) * bs[2 * l] + (sc[2 * l + 1] >> 4) * bs[2 * l + 1];
      acc[l] = fmaf(dmin, (float)prod, acc[l]);
    }
    int sumi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int jh = 0; jh < 2; ++jh) {
      for (int s = 0; s < 4; ++s) {
        for (int m = 0; m < 8; ++m) {
          int scale = sc[8 * jh + 2 * s + (m >> 2)] & 0xF;
          int p = 0;
          for (int t = 0; t < 4; ++t) {
            int l = 4 * m + t;
            p += ((q2[32 * jh + l] >> (2 * s)) & 3) * a[128 * jh + 32 * s + l];
          }
          sumi[m] += scale * p;
        }
      }
    }
    for (int m = 0; m < 8; ++m) acc[m] = fmaf(d, (float)sumi[m], acc[m]);
  }
  return hsum8(acc);
}

/* ggml_vec_dot_q3_K_q8_K, AVX2 branch: src/quant.cpp:445-547 (scalar spec :558-610). */
static float vec_dot_q3k(int n, const uint8_t* wrow, const int8_t* q8, const float* yd) {
  int nb = n / QK_K;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nb; ++i) {
    const uint8_t* blk = wrow + (size_t)i * Q3K_BYTES;
    const uint8_t* hm = blk;
    const uint8_t* q3 = blk + 32;
    int8_t scales[16];
    q3k_scales(blk + 96, scales);
    const int8_t* a = q8 + (size_t)i * QK_K;
    const float d = yd[i] * orc_half_to_float(rd16(blk + 108));
    int sumi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int jh = 0; jh < 2; ++jh) {
      for (int s = 0; s < 4; ++s) {
        int bit = 4 * jh + s;
        for (int m = 0; m < 8; ++m) {
          int scale = scales[8 * jh + 2 * s + (m >> 2)] - 32;
          int p = 0;
          for (int t = 0; t < 4; ++t) {
            int l = 4 * m + t;
            int v = ((q3[32 * jh + l] >> (2 * s)) & 3) - (((hm[l] >> bit) & 1) ? 0 : 4);
            p += v * a[128 * jh + 32 * s + l];
          }
          sumi[m] += scale * p;
        }
      }
    }
    for (int m = 0; m < 8; ++m) acc[m] = fmaf(d, (float)sumi[m], acc[m]);
  }
  return hsum8(acc);
}

/* ------------------------------------------------------------------------- */
/* GEMV: the five _matmul overloads, src/infer.cpp:121-379, dispatch :381-417 */
/* ------------------------------------------------------------------------- */
static int cdiv(int a, int b) { return (a + b - 1) / b; }

/* F32: src/infer.cpp:121-157.  The reference binary evaluates this loop as a strictly
 * sequential scalar sum of rounded products. */
static void gemv_f32(float* out, const float* x, const float* w, int n, int d,
                     const int32_t* block_size, const float* scale) {
  float one = 1.0f;
  int32_t dummy[2] = {d, n};
  if (!scale) {
    scale = &one;
    block_size = dummy;
  }
  int scale_num_cols = cdiv(n, block_size[1]);
  for (int i = 0; i < d; i++) {
    int scale_i = i / block_size[0];
    float val = 0.0f;
    for (int sj = 0; sj < cdiv(n, block_size[1]); sj++) {
      float sv = scale[scale_i * scale_num_cols + sj];
      for (int jj = 0; jj < block_size[1]; jj++) {
        int j = sj * block_size[1] + jj;
        if (j >= n) break;
        val += (w[(size_t)i * n + j] * x[j]) * sv;
      }
    }
    out[i] = val;
  }
}

/* 16 virtual lanes = the two 8-lane AVX2 accumulators of src/infer.cpp:187-227 / 264-307:
 * w' = cvt(w) * scale (vmulps), acc = fma(w', x, acc); reduce lo+hi, halves, dpps. */
static inline float reduce16(const float acc[16]) {
  float s8[8], s4[4];
  for (int l = 0; l < 8; l++) s8[l] = acc[l] + acc[l + 8];
  for (int l = 0; l < 4; l++) s4[l] = s8[l] + s8[l + 4];
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

static void gemv_f16(float* out, const float* x, const uint16_t* w, int n, int d,
                     const int32_t* block_size, const float* scale) { /* src/infer.cpp:161-233 */
  float one = 1.0f;
  int32_t dummy[2] = {d, n};
  if (!scale) {
    scale = &one;
    block_size = dummy;
  }
  int scale_num_cols = cdiv(n, block_size[1]);
  for (int i = 0; i < d; i++) {
    int scale_i = i / block_size[0];
    float acc[16] = {0};
    for (int sj = 0; sj < cdiv(n, block_size[1]); sj++) {
      float sv = scale[scale_i * scale_num_cols + sj];
      for (int jj = 0; jj < block_size[1]; jj += 16) {
        int j = sj * block_size[1] + jj;
        if (j >= n) break;
        for (int l = 0; l < 16; l++) {
          float wv = orc_half_to_float(w[(size_t)i * n + j + l]) * sv;
          acc[l] = fmaf(wv, x[j + l], acc[l]);
        }
      }
    }
    out[i] = reduce16(acc);
  }
}

static void gemv_f8(float* out, const float* x, const uint8_t* w, int n, int d,
                    const int32_t* block_size, const float* scale) { /* src/infer.cpp:238-313 */
  float one = 1.0f;
  int32_t dummy[2] = {d, n};
  if (!scale) {
    scale = &one;
    block_size = dummy;
  }
  int scale_num_cols = cdiv(n, block_size[1]);
  for (int i = 0; i < d; i++) {
    int scale_i = i / block_size[0];
    float acc[16] = {0};
    for (int sj = 0; sj < cdiv(n, block_size[1]); sj++) {
      float sv = scale[scale_i * scale_num_cols + sj];
      for (int jj = 0; jj < block_size[1]; jj += 16) {
        int j = sj * block_size[1] + jj;
        if (j >= n) break;
        for (int l = 0; l < 16; l++) {
          float wv = orc_f8e5m2_to_float(w[(size_t)i * n + j + l]) * sv;
          acc[l] = fmaf(wv, x[j + l], acc[l]);
        }
      }
    }
    out[i] = reduce16(acc);
  }
}

/* K-quants: quantise x to Q8_K once, then one vec_dot per row: src/infer.cpp:315-379.
 * (The reference re-quantises in 512-element chunks; blocks are independent, so the result
 * is the same as quantising the whole vector.) */
static int gemv_kquant(int quant, float* out, const float* x, const uint8_t* w, int n, int d) {
  if (n % QK_K) return fail("k-quant gemv: n % 256 != 0");
  int nb = n / QK_K;
  int8_t* qs = (int8_t*)malloc((size_t)n);
  float* yd = (float*)malloc(sizeof(float) * nb);
  int16_t* bs = (int16_t*)malloc(sizeof(int16_t) * nb * 16);
  orc_q8k_quantize(x, n, qs, yd, bs);
  size_t row_bytes = (size_t)nb * (quant == DSK_QUANT_Q2_K ? Q2K_BYTES : Q3K_BYTES);
  for (int i = 0; i < d; i++) {
    const uint8_t* row = w + (size_t)i * row_bytes;
    out[i] = quant == DSK_QUANT_Q2_K ? vec_dot_q2k(n, row, qs, yd, bs) : vec_dot_q3k(n, row, qs, yd);
  }
  free(qs);
  free(yd);
  free(bs);
  return 0;
}

/* The same K-quant GEMV with the activation ALREADY quantised (teacher forcing at a Q8_K staging point: the parity
 * harness feeds the device's own int8 codes and block scales, so the integer dots see identical operands and only the
 * float association differs).  bsums are recomputed from the codes (exact, src/quant.cpp:643-649). */
int orc_gemv_q8(int quant, const void* w, int d, int n, const int8_t* qs, const float* yd, float* out) {
  if (quant != DSK_QUANT_Q2_K && quant != DSK_QUANT_Q3_K) return fail("gemv_q8: k-quants only");
  if (n % QK_K) return fail("k-quant gemv: n % 256 != 0");
  int nb = n / QK_K;
  int16_t* bs = (int16_t*)malloc(sizeof(int16_t) * nb * 16);
  for (int j = 0; j < nb * 16; ++j) {
    int sum = 0;
    for (int ii = 0; ii < 16; ++ii) sum += qs[j * 16 + ii];
    bs[j] = (int16_t)sum;
  }
  size_t row_bytes = (size_t)nb * (quant == DSK_QUANT_Q2_K ? Q2K_BYTES : Q3K_BYTES);
  for (int i = 0; i < d; i++) {
    const uint8_t* row = (const uint8_t*)w + (size_t)i * row_bytes;
    out[i] = quant == DSK_QUANT_Q2_K ? vec_dot_q2k(n, row, qs, yd, bs) : vec_dot_q3k(n, row, qs, yd);
  }
  free(bs);
  return 0;
}

/* The Q2_K x Q8_K GEMV with the f32 ASSOCIATION of the device's tiled kernels (csrc/tile_device.h), on a given Q8_K vector.
 * Same integers as ggml_vec_dot_q2_K_q8_K (src/quant.cpp:666-783: the sub-block sums times their 4-bit scales, the 4-bit
 * mins times the sub-block sums of the codes); what differs from vec_dot_q2k above - the AVX2 lane order - is only how the
 * block terms are added in f32.  The device splits the 16 sub-blocks of a block into four groups g (sub-blocks 4g .. 4g+3),
 * keeps per group
 *     accd = fma(dx*d, float(F_g * isum_g), accd),  accm = fma(dx*dmin, float(summs_g), accm)      [F_g = 4 (g even), 16 (g odd)]
 * over the blocks of an ITEM (4 consecutive blocks when the row has more than 8, one block otherwise), closes an item with
 *     P = fma(accd, 1 / F_g, -accm),
 * adds the items' P per group in order (from 0) and the four groups as (S0 + S1) + (S2 + S3).
 * Test infrastructure: lets the parity tests demand BIT equality from the tiled kernels (tests/test_tiles_gpu.py) while the
 * distance between the two associations is bounded on the CPU (tests/test_oracle_pin.py). */
int orc_gemv_q2k_tiles(const void* w, int d, int n, const int8_t* qs, const float* yd, float* out) {
  if (n % QK_K) return fail("k-quant gemv: n % 256 != 0");
  const int nb = n / QK_K, seg = nb > 8 ? 4 : 1;
  for (int r = 0; r < d; ++r) {
    const uint8_t* row = (const uint8_t*)w + (size_t)r * nb * Q2K_BYTES;
    float S[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < nb; b0 += seg) {
      float accd[4] = {0.f, 0.f, 0.f, 0.f}, accm[4] = {0.f, 0.f, 0.f, 0.f};
      for (int b = b0; b < b0 + seg && b < nb; ++b) {
        const uint8_t* blk = row + (size_t)b * Q2K_BYTES;
        const uint8_t* sc = blk;
        const uint8_t* q2 = blk + 16;
        const int8_t* a = qs + (size_t)b * QK_K;
        const float dd = yd[b] * orc_half_to_float(rd16(blk + 80));
        const float dmn = yd[b] * orc_half_to_float(rd16(blk + 82));
        for (int g = 0; g < 4; ++g) {
          int isum = 0, summs = 0;
          for (int i = 0; i < 4; ++i) {
            const int j = 4 * g + i, h = j >> 3, s = (j & 7) >> 1, lh = j & 1;  /* layout of dequantize_row_q2_K, src/quant.cpp:217-247 */
            int dot = 0, bsum = 0;
            for (int t = 0; t < 16; ++t) {
              dot += ((q2[32 * h + 16 * lh + t] >> (2 * s)) & 3) * a[16 * j + t];
              bsum += a[16 * j + t];
            }
            isum += (sc[j] & 0xF) * dot;
            summs += (sc[j] >> 4) * bsum;
          }
          const int F = (g & 1) ? 16 : 4;
          accd[g] = fmaf(dd, (float)(F * isum), accd[g]);
          accm[g] = fmaf(dmn, (float)summs, accm[g]);
        }
      }
      for (int g = 0; g < 4; ++g) S[g] += fmaf(accd[g], (g & 1) ? 0.0625f : 0.25f, -accm[g]);
    }
    out[r] = (S[0] + S[1]) + (S[2] + S[3]);
  }
  return 0;
}

int orc_gemv(int quant, const void* w, const float* scale, const int32_t* block_size, int d,
             int n, const float* x, float* out) {
  switch (quant) {
    case DSK_QUANT_F32: gemv_f32(out, x, (const float*)w, n, d, block_size, scale); return 0;
    case DSK_QUANT_F16:
      if (n % 16) return fail("f16 gemv: n % 16 != 0");
      gemv_f16(out, x, (const uint16_t*)w, n, d, block_size, scale);
      return 0;
    case DSK_QUANT_F8E5M2:
      if (n % 16) return fail("f8 gemv: n % 16 != 0");
      gemv_f8(out, x, (const uint8_t*)w, n, d, block_size, scale);
      return 0;
    case DSK_QUANT_Q2_K:
    case DSK_QUANT_Q3_K: return gemv_kquant(quant, out, x, (const uint8_t*)w, n, d);
  }
  return fail("bad quant");
}

/* matmul_expert: src/infer.cpp:423-469 */
int orc_gemv_expert(int quant, const void* w, const float* scale, const int32_t* block_size,
                    int expert, int d, int n, const float* x, float* out) {
  size_t expert_size = (size_t)n * d;
  const float* sdata = NULL;
  if (scale) {
    int ess = cdiv(d, block_size[0]) * cdiv(n, block_size[1]);
    sdata = scale + (size_t)expert * ess;
  }
  size_t off = (size_t)expert * expert_size;
  const uint8_t* base = (const uint8_t*)w;
  switch (quant) {
    case DSK_QUANT_F32: base += off * 4; break;
    case DSK_QUANT_F16: base += off * 2; break;
    case DSK_QUANT_F8E5M2: base += off; break;
    case DSK_QUANT_Q2_K: base += off / QK_K * Q2K_BYTES; break;
    case DSK_QUANT_Q3_K: base += off / QK_K * Q3K_BYTES; break;
    default: return fail("bad quant");
  }
  return orc_gemv(quant, base, sdata, block_size, d, n, x, out);
}

/* Model::_copy_embedding: src/infer.cpp:1217-1263 */
int orc_embed_row(int quant, const void* w, const float* scale, const int32_t* block_size,
                  int dim, int token, float* out) {
  switch (quant) {
    case DSK_QUANT_F32:
      memcpy(out, (const float*)w + (size_t)token * dim, sizeof(float) * dim);
      return 0;
    case DSK_QUANT_F16:
      for (int i = 0; i < dim; i++) out[i] = orc_half_to_float(((const uint16_t*)w)[(size_t)token * dim + i]);
      return 0;
    case DSK_QUANT_F8E5M2: {
      int ncols = cdiv(dim, block_size[1]);
      for (int i = 0; i < dim; i++) {
        float s = scale[(token / block_size[0]) * ncols + i / block_size[1]];
        out[i] = orc_f8e5m2_to_float(((const uint8_t*)w)[(size_t)token * dim + i]) * s;
      }
      return 0;
    }
    case DSK_QUANT_Q2_K:
      dequant_row_q2k((const uint8_t*)w + (size_t)token * (dim / QK_K) * Q2K_BYTES, dim, out);
      return 0;
    case DSK_QUANT_Q3_K:
      dequant_row_q3k((const uint8_t*)w + (size_t)token * (dim / QK_K) * Q3K_BYTES, dim, out);
      return 0;
  }
  return fail("bad quant");
}

/* ------------------------------------------------------------------------- */
/* small ops */
/* ------------------------------------------------------------------------- */
void orc_rmsnorm(float* o, const float* x, const float* weight, int size, float eps) { /* src/infer.cpp:601-611 */
  float rms = 0.0f;
  for (int i = 0; i < size; ++i) rms += x[i] * x[i];
  rms = sqrtf(rms / size + eps);
  float scale = 1.0f / rms;
  for (int i = 0; i < size; ++i) o[i] = x[i] * scale * weight[i];
}

void orc_softmax(float* o, const float* x, int size) { /* src/infer.cpp:472-487 */
  float score_max = -FLT_MAX;
  for (int i = 0; i < size; ++i)
    if (x[i] > score_max) score_max = x[i];
  float score_sum = 0.0f;
  for (int i = 0; i < size; ++i) {
    o[i] = expf(x[i] - score_max);
    score_sum += o[i];
  }
  for (int i = 0; i < size; ++i) o[i] /= score_sum;
}

/*
 * Sampler::sample (src/sampler.cpp:41-75) and sample_argmax (:28-39).  temperature == 0: the first maximum (strict >).
 * Otherwise: max, sum of expf((l - max) / T) left to right, r = coin * top_p with coin = rand() / (float)RAND_MAX,
 * then the first i IN VOCABULARY ORDER with cumsum >= r -- for top_p < 1 the reference sorts an index array (:58-63)
 * but its loop (:67-72) still walks logits[i], so the sort has no effect on the result; vocab_size - 1 if none.
 */
int orc_sample(const float* logits, int vocab_size, float temperature, float top_p, float coin) {
  float max_val = -FLT_MAX;
  int argmax = 0;
  for (int i = 0; i < vocab_size; ++i)
    if (logits[i] > max_val) { max_val = logits[i]; argmax = i; }
  if (temperature == 0.0f) return argmax;
  float sum = 0.0f;
  for (int i = 0; i < vocab_size; ++i) sum += expf((logits[i] - max_val) / temperature);
  const float r = coin * top_p;
  float cumsum = 0.0f;
  for (int i = 0; i < vocab_size; ++i) {
    cumsum += expf((logits[i] - max_val) / temperature) / sum;
    if (cumsum >= r) return i;
  }
  return vocab_size - 1;
}

/* Sampler::sample_prob (src/sampler.cpp:12-26): softmax probability of logits[index], sums left to right in f32 */
float orc_sample_prob(const float* logits, int vocab_size, int index) {
  float max_val = -FLT_MAX;
  for (int i = 0; i < vocab_size; ++i)
    if (logits[i] > max_val) max_val = logits[i];
  float sum = 0.0f;
  for (int i = 0; i < vocab_size; ++i) sum += expf(logits[i] - max_val);
  return expf(logits[index] - max_val) / sum;
}

static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); } /* src/infer.cpp:489-491 */
static inline float siluf(float x) { return x / (1.0f + expf(-x)); }       /* src/infer.cpp:640-642 */
static inline float geluf(float x) {                                       /* src/infer.cpp:636-638 */
  return 0.5f * x * (1.0f + tanhf(0.797885f * (x + 0.044715f * x * x * x)));
}

/*
 * moe_gate: src/infer.cpp:493-599.  Tie-break: strict '>' so the lowest index wins.
 * Group-limited first pass (:551-566) starts from best = -1 and compares against x[-1]; the
 * intended "first unmasked candidate, then strict greater" semantics are implemented (they
 * coincide with the reference binary whenever a group holds a score > 0).
 */
void orc_moe_gate(const float* scores_in, const float* bias, int n_routed, int n_active,
                  int norm_topk_prob, float routed_scaling_factor, int scoring_func,
                  int topk_method, int n_group, int topk_group, int32_t* active_experts,
                  float* active_weights, float* scores_out) {
  float xbuf[256];
  float* x = scores_out ? scores_out : xbuf;
  if (scoring_func == DSK_SCORE_SOFTMAX) {
    orc_softmax(x, scores_in, n_routed);
  } else {
    for (int i = 0; i < n_routed; i++) x[i] = sigmoidf(scores_in[i]);
  }
  if (bias)
    for (int i = 0; i < n_routed; ++i) x[i] += bias[i];

  uint8_t mask[32];
  memset(mask, 0, sizeof mask);
  float wsum = 0.0f;
  if (topk_method == DSK_TOPK_GROUP_LIMITED_GREEDY) {
    int group_size = n_routed / n_group;
    for (int g = 0; g < n_group; g++) {
      for (int k = 0; k < topk_group; k++) {
        int best = -1;
        for (int j = g * group_size; j < (g + 1) * group_size; j++) {
          if ((mask[j / 8] & (1u << (j % 8))) == 0 && (best == -1 || x[j] > x[best])) best = j;
        }
        mask[best / 8] |= 1u << (best % 8);
      }
    }
    for (int i = 0; i < 32; i++) mask[i] = (uint8_t)~mask[i];
  }
  for (int k = 0; k < n_active; ++k) {
    int best = -1;
    for (int j = 0; j < n_routed; ++j) {
      if ((mask[j / 8] & (1u << (j % 8))) == 0 && (best == -1 || x[j] > x[best])) best = j;
    }
    active_experts[k] = best;
    wsum += x[best];
    mask[best / 8] |= 1u << (best % 8);
  }
  if (!norm_topk_prob) wsum = 1.0f;
  for (int k = 0; k < n_active; ++k)
    active_weights[k] = x[active_experts[k]] / wsum * routed_scaling_factor;
}

/* Source: freq = 1.0f / powf(theta, (float)j_head / (float)head_dim) (src/infer.cpp:655,675).
 * The reference's -O3 -ffast-math build (Makefile:31) evaluates it as
 * powf(theta, -(j_head * (1.0f / head_dim))) -- seen in its assembly and checked against
 * oracle/_ref bit-for-bit in tests/test_oracle_pin.py; at pos ~ 1e3 the two forms differ by 1e-5
 * in cos/sin, so the oracle (and the engine's host-side table) follow the binary. */
static inline float ref_rope_freq(float theta, int j_head, int head_dim) {
  const float inv = 1.0f / (float)head_dim;
  return powf(theta, -((float)j_head * inv));
}

/* rope (V2, de-interleaving output) src/infer.cpp:648-668; rope_v3 (in place) :670-685 */
void orc_rope(float* vec, int d, int head_dim, int pos, float theta, int is_v3) {
  float buf[512];
  for (int i = 0; i < d; i += 2) {
    int j_head = i % head_dim;
    float freq = ref_rope_freq(theta, j_head, head_dim);
    float val = pos * freq;
    float fcr = cosf(val), fci = sinf(val);
    float v0 = vec[i], v1 = vec[i + 1];
    if (is_v3) {
      vec[i] = v0 * fcr - v1 * fci;
      vec[i + 1] = v0 * fci + v1 * fcr;
    } else {
      buf[i / 2] = v0 * fcr - v1 * fci;
      buf[i / 2 + d / 2] = v0 * fci + v1 * fcr;
    }
  }
  if (!is_v3)
    for (int i = 0; i < d; i++) vec[i] = buf[i];
}

/* f16 variants used for the attention-sink rotation: src/infer.cpp:687-724 */
void orc_rope_f16(uint16_t* vec, int d, int head_dim, int pos, float theta, int is_v3) {
  float buf[512];
  for (int i = 0; i < d; i += 2) {
    int j_head = i % head_dim;
    float freq = ref_rope_freq(theta, j_head, head_dim);
    float val = pos * freq;
    float fcr = cosf(val), fci = sinf(val);
    float v0 = orc_half_to_float(vec[i]), v1 = orc_half_to_float(vec[i + 1]);
    if (is_v3) {
      vec[i] = orc_float_to_half(v0 * fcr - v1 * fci);
      vec[i + 1] = orc_float_to_half(v0 * fci + v1 * fcr);
    } else {
      buf[i / 2] = v0 * fcr - v1 * fci;
      buf[i / 2 + d / 2] = v0 * fci + v1 * fcr;
    }
  }
  if (!is_v3)
    for (int i = 0; i < d; i++) vec[i] = orc_float_to_half(buf[i]);
}

/* attn: src/infer.cpp:728-762 */
void orc_attn(float* xout, float* atth, const float* qh, const uint16_t* kh, const uint16_t* vh,
              int head_dim, int v_head_dim, int n_heads, int kv_len) {
  int k_stride = n_heads * head_dim;
  for (int t = 0; t < kv_len; ++t) {
    float score = 0.0f;
    for (int i = 0; i < head_dim; ++i) score += qh[i] * orc_half_to_float(kh[(size_t)t * k_stride + i]);
    score /= sqrtf((float)head_dim);
    atth[t] = score;
  }
  orc_softmax(atth, atth, kv_len);
  int v_stride = n_heads * v_head_dim;
  for (int i = 0; i < v_head_dim; ++i) {
    float vi = 0.0f;
    for (int t = 0; t < kv_len; ++t) vi += atth[t] * orc_half_to_float(vh[(size_t)t * v_stride + i]);
    xout[i] = vi;
  }
}

/* attn_mla: src/infer.cpp:766-804 */
void orc_attn_mla(float* xout, float* atth, const float* qh_c, const float* qh_rope,
                  const uint16_t* ckv, const uint16_t* krope, int head_dim, int kv_lora_rank,
                  int rope_dim, int kv_len) {
  for (int t = 0; t < kv_len; ++t) {
    float score = 0.0f;
    for (int i = 0; i < kv_lora_rank; ++i) score += qh_c[i] * orc_half_to_float(ckv[(size_t)t * kv_lora_rank + i]);
    for (int i = 0; i < rope_dim; ++i) score += qh_rope[i] * orc_half_to_float(krope[(size_t)t * rope_dim + i]);
    score /= sqrtf((float)head_dim);
    atth[t] = score;
  }
  orc_softmax(atth, atth, kv_len);
  for (int i = 0; i < kv_lora_rank; ++i) {
    float vi = 0.0f;
    for (int t = 0; t < kv_len; ++t) vi += atth[t] * orc_half_to_float(ckv[(size_t)t * kv_lora_rank + i]);
    xout[i] = vi;
  }
}

/* ------------------------------------------------------------------------- */
/* model: tensor table + KV caches + scratch (src/model.cpp:149-285, 354-461, 518-620, 677-726) */
/* ------------------------------------------------------------------------- */
typedef struct {
  const void* w;
  const float* s; /* F8 block scales */
  int quant;
  int32_t shape[4];
  size_t bytes;
} orc_tensor;

#define ORC_NROLES 32

typedef struct {
  orc_tensor t[ORC_NROLES];
  uint16_t *key_cache, *value_cache;   /* MHA: (seq, H*head_dim), (seq, H*v) */
  uint16_t *kv_nope_cache, *kv_rope_cache; /* MLA: (seq, kv_lora), (seq, rope) */
} orc_layer;

struct orc_model {
  dsk_config c;
  int head_dim;
  orc_tensor g[4]; /* EMBED, FINAL_NORM, OUTPUT */
  orc_layer* L;
  int finalized;
  /* scratch */
  float *x, *xb, *xb2, *hb, *hb2, *q_a, *q, *kv_a, *kv_b, *k, *v, *att, *q_c, *q_rope;
  float *moe_weights, *active_w;
  int32_t* active_e;
  /* taps */
  float* trace_x;       /* n_layers * dim */
  int32_t* route_e;     /* n_layers * n_active */
  float* route_w;
  float* router_logits; /* n_layers * n_routed */
};

int orc_model_create(const dsk_config* cfg, orc_model** out) {
  orc_model* m = (orc_model*)calloc(1, sizeof *m);
  if (!m) return fail("oom");
  m->c = *cfg;
  m->head_dim = cfg->qk_nope_head_dim + cfg->qk_rope_head_dim;
  m->L = (orc_layer*)calloc((size_t)cfg->n_layers, sizeof(orc_layer));
  *out = m;
  return 0;
}

int orc_model_bind(orc_model* m, int role, int layer, int quant, const int32_t shape[4],
                   const void* host_ptr, size_t bytes) {
  int is_scale = role >= DSK_ROLE_SCALE;
  int r = is_scale ? role - DSK_ROLE_SCALE : role;
  orc_tensor* t;
  if (r < 10) {
    if (r > 2) return fail("bad model-level role");
    t = &m->g[r];
  } else {
    if (layer < 0 || layer >= m->c.n_layers || r >= ORC_NROLES) return fail("bad layer/role");
    t = &m->L[layer].t[r];
  }
  if (is_scale) {
    t->s = (const float*)host_ptr;
  } else {
    t->w = host_ptr;
    t->quant = quant;
    memcpy(t->shape, shape, sizeof t->shape);
    t->bytes = bytes;
  }
  return 0;
}

static int imax(int a, int b) { return a > b ? a : b; }

int orc_model_finalize(orc_model* m) {
  const dsk_config* c = &m->c;
  int H = c->n_heads, hd = m->head_dim;
  if (!m->g[DSK_ROLE_EMBED].w || !m->g[DSK_ROLE_FINAL_NORM].w) return fail("missing embed/final norm");
  if (!m->g[DSK_ROLE_OUTPUT].w) m->g[DSK_ROLE_OUTPUT] = m->g[DSK_ROLE_EMBED]; /* tied: src/model.cpp:852-856 */
  size_t seq = (size_t)c->max_seq_len;
  for (int l = 0; l < c->n_layers; l++) {
    orc_layer* L = &m->L[l];
    if (c->use_mla) {
      L->kv_nope_cache = (uint16_t*)calloc(seq * c->kv_lora_rank, 2);
      L->kv_rope_cache = (uint16_t*)calloc(seq * c->qk_rope_head_dim, 2);
    } else {
      L->key_cache = (uint16_t*)calloc(seq * H * hd, 2);
      L->value_cache = (uint16_t*)calloc(seq * H * c->v_head_dim, 2);
    }
  }
  int xb2n = imax(c->dim, imax(H * c->v_head_dim, H * c->kv_lora_rank));
  int hbn = imax(imax(c->dim, c->hidden_dim), imax(1, c->n_shared_experts) * c->moe_intermediate_size);
  m->x = (float*)calloc((size_t)c->dim, 4);
  m->xb = (float*)calloc((size_t)c->dim, 4);
  m->xb2 = (float*)calloc((size_t)xb2n, 4);
  m->hb = (float*)calloc((size_t)hbn, 4);
  m->hb2 = (float*)calloc((size_t)hbn, 4);
  m->q_a = (float*)calloc((size_t)imax(1, c->q_lora_rank), 4);
  m->q = (float*)calloc((size_t)H * hd, 4);
  m->kv_a = (float*)calloc((size_t)c->kv_lora_rank + c->qk_rope_head_dim, 4);
  m->kv_b = (float*)calloc((size_t)H * (c->qk_nope_head_dim + c->v_head_dim), 4);
  m->k = (float*)calloc((size_t)H * hd, 4);
  m->v = (float*)calloc((size_t)H * c->v_head_dim, 4);
  m->att = (float*)calloc((size_t)H * seq, 4);
  m->q_c = (float*)calloc((size_t)H * imax(1, c->kv_lora_rank), 4);
  m->q_rope = (float*)calloc((size_t)H * imax(1, c->qk_rope_head_dim), 4);
  m->moe_weights = (float*)calloc((size_t)imax(1, c->n_routed_experts), 4);
  m->active_w = (float*)calloc((size_t)imax(1, c->n_active_routed), 4);
  m->active_e = (int32_t*)calloc((size_t)imax(1, c->n_active_routed), 4);
  m->trace_x = (float*)calloc((size_t)c->n_layers * c->dim, 4);
  m->route_e = (int32_t*)calloc((size_t)c->n_layers * imax(1, c->n_active_routed), 4);
  m->route_w = (float*)calloc((size_t)c->n_layers * imax(1, c->n_active_routed), 4);
  m->router_logits = (float*)calloc((size_t)c->n_layers * imax(1, c->n_routed_experts), 4);
  m->finalized = 1;
  return 0;
}

int orc_model_destroy(orc_model* m) {
  if (!m) return 0;
  for (int l = 0; l < m->c.n_layers; l++) {
    free(m->L[l].key_cache);
    free(m->L[l].value_cache);
    free(m->L[l].kv_nope_cache);
    free(m->L[l].kv_rope_cache);
  }
  free(m->L);
  float* bufs[] = {m->x, m->xb, m->xb2, m->hb, m->hb2, m->q_a, m->q, m->kv_a, m->kv_b, m->k, m->v,
                   m->att, m->q_c, m->q_rope, m->moe_weights, m->active_w, m->trace_x, m->route_w,
                   m->router_logits};
  for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; i++) free(bufs[i]);
  free(m->active_e);
  free(m->route_e);
  free(m);
  return 0;
}

static int mm(const orc_model* m, const orc_tensor* t, const float* x, float* out) {
  if (!t->w) return fail("forward: unbound tensor");
  return orc_gemv(t->quant, t->w, t->s, m->c.block_size, t->shape[0], t->shape[1], x, out);
}
static int mm_e(const orc_model* m, const orc_tensor* t, int e, const float* x, float* out) {
  if (!t->w) return fail("forward: unbound tensor");
  return orc_gemv_expert(t->quant, t->w, t->s, m->c.block_size, e, t->shape[1], t->shape[2], x, out);
}

static void rope_any(float* vec, int d, int pos, float theta, int is_v3) { orc_rope(vec, d, d, pos, theta, is_v3); }

/* BlockMHA::_attention_impl: src/infer.cpp:934-1049 */
static int attention_mha(orc_model* m, orc_layer* L, int pos, int kv_sink, int kv_pos, int kv_len) {
  const dsk_config* c = &m->c;
  int H = c->n_heads, hd = m->head_dim, rope = c->qk_rope_head_dim, nope = c->qk_nope_head_dim, vd = c->v_head_dim;
  int is_v3 = c->has_moegate_bias;
  if (c->q_lora_rank > 0) {
    if (mm(m, &L->t[DSK_ROLE_WQ_A], m->xb, m->q_a)) return -1;
    orc_rmsnorm(m->q_a, m->q_a, (const float*)L->t[DSK_ROLE_Q_A_NORM].w, c->q_lora_rank, c->norm_eps);
    if (mm(m, &L->t[DSK_ROLE_WQ_B], m->q_a, m->q)) return -1;
  } else {
    if (mm(m, &L->t[DSK_ROLE_WQ], m->xb, m->q)) return -1;
  }
  if (mm(m, &L->t[DSK_ROLE_WKV_A], m->xb, m->kv_a)) return -1;
  for (int h = 0; h < H; h++) rope_any(m->q + h * hd + nope, rope, pos, c->rope_theta, is_v3);
  float* k_rope = m->kv_a + c->kv_lora_rank;
  rope_any(k_rope, rope, pos, c->rope_theta, is_v3);
  orc_rmsnorm(m->kv_a, m->kv_a, (const float*)L->t[DSK_ROLE_KV_A_NORM].w, c->kv_lora_rank, c->norm_eps);
  if (mm(m, &L->t[DSK_ROLE_WKV_B], m->kv_a, m->kv_b)) return -1;
  for (int h = 0; h < H; h++) {
    const float* kvb = m->kv_b + (size_t)h * (nope + vd);
    for (int i = 0; i < nope; i++) m->k[h * hd + i] = kvb[i];
    for (int i = 0; i < rope; i++) m->k[h * hd + nope + i] = k_rope[i];
    for (int i = 0; i < vd; i++) m->v[h * vd + i] = kvb[nope + i];
  }
  uint16_t* kc = L->key_cache + (size_t)kv_pos * H * hd;
  uint16_t* vc = L->value_cache + (size_t)kv_pos * H * vd;
  for (int i = 0; i < H * hd; i++) kc[i] = orc_float_to_half(m->k[i]);
  for (int i = 0; i < H * vd; i++) vc[i] = orc_float_to_half(m->v[i]);
  for (int r = 0; r < kv_sink; r++) { /* sink rotation, src/infer.cpp:1008-1020 */
    uint16_t* key = L->key_cache + (size_t)r * H * hd;
    for (int h = 0; h < H; h++) orc_rope_f16(key + h * hd + nope, rope, rope, 1, c->rope_theta, is_v3);
  }
  for (int h = 0; h < H; h++) {
    orc_attn(m->xb2 + (size_t)h * vd, m->att + (size_t)h * c->max_seq_len, m->q + (size_t)h * hd,
             L->key_cache + h * hd, L->value_cache + h * vd, hd, vd, H, kv_len);
  }
  return mm(m, &L->t[DSK_ROLE_WO], m->xb2, m->hb);
}

/* BlockMLA::_attention_impl: src/infer.cpp:1051-1141 */
static int attention_mla(orc_model* m, orc_layer* L, int pos, int kv_sink, int kv_pos, int kv_len) {
  const dsk_config* c = &m->c;
  int H = c->n_heads, rope = c->qk_rope_head_dim, lora = c->kv_lora_rank, vd = c->v_head_dim;
  int is_v3 = c->has_moegate_bias;
  if (c->q_lora_rank <= 0) return fail("MLA requires q_lora_rank > 0");
  if (mm(m, &L->t[DSK_ROLE_WQ_A], m->xb, m->q_a)) return -1;
  orc_rmsnorm(m->q_a, m->q_a, (const float*)L->t[DSK_ROLE_Q_A_NORM].w, c->q_lora_rank, c->norm_eps);
  if (mm(m, &L->t[DSK_ROLE_WKV_A], m->xb, m->kv_a)) return -1;
  if (mm(m, &L->t[DSK_ROLE_WQ_ROPE_B], m->q_a, m->q_rope)) return -1;
  if (mm(m, &L->t[DSK_ROLE_WC], m->q_a, m->q_c)) return -1;
  for (int h = 0; h < H; h++) rope_any(m->q_rope + h * rope, rope, pos, c->rope_theta, is_v3);
  float* k_rope = m->kv_a + lora;
  rope_any(k_rope, rope, pos, c->rope_theta, is_v3);
  orc_rmsnorm(m->kv_a, m->kv_a, (const float*)L->t[DSK_ROLE_KV_A_NORM].w, lora, c->norm_eps);
  uint16_t* nc = L->kv_nope_cache + (size_t)kv_pos * lora;
  uint16_t* rc = L->kv_rope_cache + (size_t)kv_pos * rope;
  for (int i = 0; i < lora; i++) nc[i] = orc_float_to_half(m->kv_a[i]);
  for (int i = 0; i < rope; i++) rc[i] = orc_float_to_half(k_rope[i]);
  for (int r = 0; r < kv_sink; r++) orc_rope_f16(L->kv_rope_cache + (size_t)r * rope, rope, rope, 1, c->rope_theta, is_v3);
  for (int h = 0; h < H; h++) {
    orc_attn_mla(m->xb2 + (size_t)h * lora, m->att + (size_t)h * c->max_seq_len, m->q_c + (size_t)h * lora,
                 m->q_rope + (size_t)h * rope, L->kv_nope_cache, L->kv_rope_cache, m->head_dim, lora, rope, kv_len);
  }
  /* per-head wv_b, viewed as (H, v_head_dim, kv_lora_rank): src/model.cpp:576-580, src/infer.cpp:1134-1137 */
  orc_tensor wv = L->t[DSK_ROLE_WV_B];
  for (int h = 0; h < H; h++) {
    if (!wv.w) return fail("forward: unbound wv_b");
    if (orc_gemv_expert(wv.quant, wv.w, wv.s, c->block_size, h, vd, lora, m->xb2 + (size_t)h * lora,
                        m->kv_b + (size_t)h * vd))
      return -1;
  }
  return mm(m, &L->t[DSK_ROLE_WO], m->kv_b, m->hb);
}

static void glu_act(const dsk_config* c, float* hb, const float* hb2, int n) {
  if (c->act == DSK_ACT_GELU)
    for (int i = 0; i < n; i++) hb[i] = geluf(hb[i]) * hb2[i];
  else
    for (int i = 0; i < n; i++) hb[i] = siluf(hb[i]) * hb2[i];
}

/* Block::_block_cpu: src/infer.cpp:810-932 */
static int block_forward(orc_model* m, int l, int pos, int kv_sink, int kv_pos, int kv_len) {
  const dsk_config* c = &m->c;
  orc_layer* L = &m->L[l];
  orc_rmsnorm(m->xb, m->x, (const float*)L->t[DSK_ROLE_ATTN_NORM].w, c->dim, c->norm_eps);
  if ((c->use_mla ? attention_mla : attention_mha)(m, L, pos, kv_sink, kv_pos, kv_len)) return -1;
  for (int i = 0; i < c->dim; ++i) m->x[i] += m->hb[i];
  orc_rmsnorm(m->xb, m->x, (const float*)L->t[DSK_ROLE_FFN_NORM].w, c->dim, c->norm_eps);
  int K = c->n_active_routed;
  for (int k = 0; k < K; k++) m->route_e[(size_t)l * imax(1, K) + k] = -1;
  if (c->n_routed_experts > 0 && L->t[DSK_ROLE_MOEGATE].w) {
    const orc_tensor* gate = &L->t[DSK_ROLE_MOEGATE];
    if (orc_gemv(DSK_QUANT_F32, gate->w, NULL, NULL, c->n_routed_experts, c->dim, m->xb, m->moe_weights)) return -1;
    memcpy(m->router_logits + (size_t)l * c->n_routed_experts, m->moe_weights, sizeof(float) * c->n_routed_experts);
    orc_moe_gate(m->moe_weights, (const float*)L->t[DSK_ROLE_MOEGATE_BIAS].w, c->n_routed_experts, K,
                 c->norm_topk_prob, c->routed_scaling_factor, c->scoring_func, c->topk_method, c->n_group,
                 c->topk_group, m->active_e, m->active_w, m->moe_weights);
    for (int k = 0; k < K; k++) {
      m->route_e[(size_t)l * K + k] = m->active_e[k];
      m->route_w[(size_t)l * K + k] = m->active_w[k];
    }
    for (int k = 0; k < K; ++k) {
      int e = m->active_e[k];
      if (mm_e(m, &L->t[DSK_ROLE_W1], e, m->xb, m->hb)) return -1;
      if (mm_e(m, &L->t[DSK_ROLE_W3], e, m->xb, m->hb2)) return -1;
      glu_act(c, m->hb, m->hb2, c->moe_intermediate_size);
      if (mm_e(m, &L->t[DSK_ROLE_W2], e, m->hb, m->xb2)) return -1;
      float w = m->active_w[k];
      for (int i = 0; i < c->dim; ++i) m->x[i] += m->xb2[i] * w;
    }
    if (c->n_shared_experts > 0) {
      if (mm(m, &L->t[DSK_ROLE_SHARED_W1], m->xb, m->hb)) return -1;
      if (mm(m, &L->t[DSK_ROLE_SHARED_W3], m->xb, m->hb2)) return -1;
      glu_act(c, m->hb, m->hb2, c->n_shared_experts * c->moe_intermediate_size);
      if (mm(m, &L->t[DSK_ROLE_SHARED_W2], m->hb, m->xb2)) return -1;
      for (int i = 0; i < c->dim; ++i) m->x[i] += m->xb2[i];
    }
  } else {
    if (mm(m, &L->t[DSK_ROLE_W1], m->xb, m->hb)) return -1;
    if (mm(m, &L->t[DSK_ROLE_W3], m->xb, m->hb2)) return -1;
    glu_act(c, m->hb, m->hb2, c->hidden_dim);
    if (mm(m, &L->t[DSK_ROLE_W2], m->hb, m->xb2)) return -1;
    for (int i = 0; i < c->dim; ++i) m->x[i] += m->xb2[i];
  }
  memcpy(m->trace_x + (size_t)l * c->dim, m->x, sizeof(float) * c->dim);
  return 0;
}

/* Model::_forward_cpu: src/infer.cpp:1265-1317 */
int orc_forward(orc_model* m, int token, int pos, int mode, float* logits) {
  if (!m->finalized) return fail("forward before finalize");
  const dsk_config* c = &m->c;
  const orc_tensor* emb = &m->g[DSK_ROLE_EMBED];
  if (token < 0 || token >= c->vocab_size) return fail("token out of range");
  if (orc_embed_row(emb->quant, emb->w, emb->s, c->block_size, c->dim, token, m->x)) return -1;
  int W = c->rs_original_max_position_embeddings;
  int kv_sink = pos >= W ? KV_SINKS : 0;
  int kv_pos = kv_sink + (pos - kv_sink) % (W - kv_sink);
  int kv_len = pos >= W ? W : pos + 1;
  if (kv_pos >= c->max_seq_len || kv_len > c->max_seq_len) return fail("position exceeds max_seq_len allocation");
  for (int l = 0; l < c->n_layers; l++)
    if (block_forward(m, l, pos, kv_sink, kv_pos, kv_len)) return -1;
  if (mode == DSK_MODE_HYDRATE_KV_CACHE) return 0;
  orc_rmsnorm(m->x, m->x, (const float*)m->g[DSK_ROLE_FINAL_NORM].w, c->dim, c->norm_eps);
  const orc_tensor* cls = &m->g[DSK_ROLE_OUTPUT];
  return orc_gemv(cls->quant, cls->w, cls->s, c->block_size, c->vocab_size, c->dim, m->x, logits);
}

int orc_model_get_routing(orc_model* m, int32_t* experts, float* weights) {
  size_t n = (size_t)m->c.n_layers * imax(1, m->c.n_active_routed);
  memcpy(experts, m->route_e, n * 4);
  memcpy(weights, m->route_w, n * 4);
  return 0;
}
int orc_model_get_trace_x(orc_model* m, int layer, float* x_out) {
  if (layer < 0 || layer >= m->c.n_layers) return fail("bad layer");
  memcpy(x_out, m->trace_x + (size_t)layer * m->c.dim, sizeof(float) * m->c.dim);
  return 0;
}
int orc_model_get_router_logits(orc_model* m, int layer, float* out) {
  if (layer < 0 || layer >= m->c.n_layers) return fail("bad layer");
  memcpy(out, m->router_logits + (size_t)layer * m->c.n_routed_experts, sizeof(float) * m->c.n_routed_experts);
  return 0;
}
