// tile_device.h -- Q2_K x Q8_K row products with the sub-block dots on the matrix pipe (gfx950, wave64).
//
// What is computed is ggml_vec_dot_q2_K_q8_K (src/quant.cpp:666-783): per 256-block
//     isum = sum_j (sc[j] & 0xF) * sum_l q8[16j + l] * q2[16j + l],   summs = sum_j (sc[j] >> 4) * bsums[j]
//     acc += (dx * d) * isum - (dx * dmin) * summs
// with the 16 integer sub-block sums of 16 ROWS taken off the VALU: one v_mfma_i32_16x16x64_i8 computes
// D[16 x 16] += A[16 x 64] * B[64 x 16]; here
//   B = the weights: column n = row n of a 16-row tile; K-group g (16 bytes) = the row's qs bytes [16g, 16g + 16) with ONE 2-bit
//       field (shift 2s) masked in place = elements 128h + 32s + 16lh + t of the block (g = 2h + lh): sub-block
//       j(g, s) = 8h + 2s + lh  (layout of dequantize_row_q2_K, src/quant.cpp:217-247);
//   A = the activations as sub-block SELECTORS: row i of K-group g holds the 16 int8 codes of sub-block j(g, s) if i == j(g, s),
//       zeros otherwise.
// Four MFMAs (s = 0..3) leave D[j][n] = f(s(j)) * sum_t q8[16j + t] * q2_n[16j + t] (f = 1, 4, 16, 16: the in-place masks), exact
// in int32.  In the C/D layout lane (n = lane & 15, g4 = lane >> 4) holds the sums of sub-blocks 4 g4 .. 4 g4 + 3 of ITS row:
// four mul24 / mad24 apply the scales, one int -> f32 and one FMA per lane and block apply d, the min term is two dot4 against
// the split sub-block sums.  ~42 VALU instructions per 16 rows x 256 columns instead of ~70 (64 dot4 items); 15 / 16 of the
// MACs multiply zeros, which the matrix pipe has the headroom for (tools/mfma_gemv_probe.hip: experts' w1/w3 20.7 -> 16.1 us,
// classifier 56.7 -> 48.0 us = 6.3 TB/s).
//
// WEIGHT LAYOUT "tiles": a 16-row x 256-column tile is one contiguous 1344-byte record (= 16 blocks of 84 bytes)
//   [   0, 1024)  qs: 16 bytes at 16 * (n + 16 g): row n's qs bytes [16g, 16g + 16)
//   [1024, 1280)  scales: 4 bytes at 4 * (n + 16 g4): row n's scales[4 g4 .. 4 g4 + 3] (the reference's order)
//   [1280, 1344)  d | dmin << 16 of row n at 4 n
// and the tiles of a 16-row STRIP (all blocks of the rows) are contiguous, strips in row order; rows are padded to a multiple
// of 16 with zero blocks.  Every load instruction of a wave covers whole lines of one contiguous range.
//
// ASSOCIATION (what makes results independent of the launch geometry).  The value of a row is
//     (S0 + S1) + (S2 + S3),   S_g = sum over the row's ITEMS in order of P[item][g]   (left to right, from 0)
//     P[item][g] = fma(accd, lanefac, -accm): accd / accm the FMA chains over the item's blocks in order (from 0)
// an ITEM being 4 consecutive blocks of the row (the last one shorter) when the row has more than 8 blocks, ONE block otherwise.
// Any split of a strip's items over waves / workgroups / launches gives the same bits: the fused expert launch equals the
// two-launch form, a shard equals the unsharded model, whatever the grids.
#pragma once
#include "gemv_device.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define TILE_B 1344   // bytes of a 16-row x 256-column tile
#define TREC 320      // LDS record of a staged 256-block: codes[256] | zeros[16] | (bsum hi[4], lo[4]) x 4 | d | pad
#define TREC_ZERO 256
#define TREC_BS 272
#define TREC_D 304
DEV size_t tile_lds_bytes(int n) { return (size_t)(n >> 8) * TREC; }
// blocks per item / items per strip of a row of nb blocks (the association above)
__host__ __device__ inline constexpr int tile_seg(int nb) { return nb > 8 ? 4 : 1; }
__host__ __device__ inline constexpr int tile_ips(int nb) { return nb > 8 ? (nb + 3) >> 2 : nb; }

struct TLane {
  int aoff[4];     // LDS offset of this lane's selector operand inside a block record, per shift s
  int g;           // lane >> 4: K-group of the operands, sub-block group of the results
  int n;           // lane & 15: row inside the tile
  int shA;         // results of an even group carry factors (1, 1, 4, 4), of an odd group (16, 16, 16, 16)
  float lanefac;   // ... undone once per item: 1/4 or 1/16
  int vq, vs, vd;  // byte offsets of the lane's qs / scales / d|dmin inside a tile
};
DEV TLane tlane_init(int lane) {
  TLane L;
  L.n = lane & 15; L.g = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int j = 8 * (L.g >> 1) + 2 * s + (L.g & 1);
    L.aoff[s] = L.n == j ? 16 * j : TREC_ZERO;
  }
  L.shA = (L.g & 1) ? 0 : 2;
  L.lanefac = (L.g & 1) ? 0.0625f : 0.25f;
  L.vq = lane * 16; L.vs = 1024 + lane * 4; L.vd = 1280 + L.n * 4;
  return L;
}

struct TStep { u32x4 w; u32 scw, dm; };
DEV void tstep_load(TStep& S, rsrc_t W, const TLane& L, int soff) {
  S.w = __builtin_amdgcn_raw_buffer_load_b128(W, L.vq, soff, BUF_NT);
  S.scw = __builtin_amdgcn_raw_buffer_load_b32(W, L.vs, soff, BUF_NT);
  S.dm = __builtin_amdgcn_raw_buffer_load_b32(W, L.vd, soff, BUF_NT);
}
// one tile x one staged block (rec): accd += (dx d) * (4 or 16) isum, accm += (dx dmin) * summs  [this lane's 4 sub-blocks]
DEV void tstep_mac(const TStep& S, const uint8_t* rec, const TLane& L, float& accd, float& accm) {
  const u32x4 w = S.w;
  i32x4 b0, b1, b2, b3;
  b0.x = w.x & 0x03030303u; b0.y = w.y & 0x03030303u; b0.z = w.z & 0x03030303u; b0.w = w.w & 0x03030303u;
  b1.x = w.x & 0x0C0C0C0Cu; b1.y = w.y & 0x0C0C0C0Cu; b1.z = w.z & 0x0C0C0C0Cu; b1.w = w.w & 0x0C0C0C0Cu;
  b2.x = w.x & 0x30303030u; b2.y = w.y & 0x30303030u; b2.z = w.z & 0x30303030u; b2.w = w.w & 0x30303030u;
  b3.x = (w.x >> 2) & 0x30303030u; b3.y = (w.y >> 2) & 0x30303030u; b3.z = (w.z >> 2) & 0x30303030u; b3.w = (w.w >> 2) & 0x30303030u;
  const i32x4 a0 = *reinterpret_cast<const i32x4*>(rec + L.aoff[0]);
  const i32x4 a1 = *reinterpret_cast<const i32x4*>(rec + L.aoff[1]);
  const i32x4 a2 = *reinterpret_cast<const i32x4*>(rec + L.aoff[2]);
  const i32x4 a3 = *reinterpret_cast<const i32x4*>(rec + L.aoff[3]);
  i32x4 D = {0, 0, 0, 0};
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b3, D, 0, 0, 0);
  const u32 scw = S.scw;
  const int d0 = scw & 0xF, d1 = (scw >> 8) & 0xF, d2 = (scw >> 16) & 0xF, d3 = (scw >> 24) & 0xF;
  const int t01 = __mul24(d0, D.x) + __mul24(d1, D.y);   // all products < 2^23
  const int t23 = __mul24(d2, D.z) + __mul24(d3, D.w);
  const int isx = (t01 << L.shA) + t23;                  // even group: 4 isum; odd group: 16 isum
  const u32 m4 = (scw >> 4) & 0x0F0F0F0Fu;
  const u32x2 bs = *reinterpret_cast<const u32x2*>(rec + TREC_BS + 8 * L.g);
  const int summs = (sdot4(m4, bs.x, 0) << 8) + (int)__builtin_amdgcn_udot4(m4, bs.y, 0u, false);
  const float dx = *reinterpret_cast<const float*>(rec + TREC_D);
  const float dd = dx * h2f(S.dm & 0xffff), dmn = dx * h2f(S.dm >> 16);
  accd = fmaf(dd, (float)isx, accd);
  accm = fmaf(dmn, (float)summs, accm);
}
DEV float titem_value(float accd, float accm, const TLane& L) { return fmaf(accd, L.lanefac, -accm); }

#ifndef TILE_G
#define TILE_G 8  // column steps requested together
#endif
// Blocks [b0, b1) of ONE strip (W + soff0 = its first tile, act = the record of block 0 of its activation vector), TILE_G
// column steps requested together (then 4, then what is left); an item's partial goes to red_strip[item * 64 + lane].
// b0 is a multiple of SEG; b1 is a multiple of SEG or the end of the row.  Full groups run straight-line (hipcc counts the
// outstanding loads exactly: step u is multiplied while steps u + 1.. are in flight, and the LDS reads, matrix instructions and
// scale arithmetic of neighbouring steps interleave); a shorter last group requests exactly its steps.
template <int SEG, int N>
DEV void tile_group(rsrc_t W, int soff0, const uint8_t* act, int b, float* red_strip, const TLane& L, int lane) {
  TStep S[N];
#pragma unroll
  for (int u = 0; u < N; ++u) tstep_load(S[u], W, L, soff0 + (b + u) * TILE_B);
  float accd = 0.f, accm = 0.f;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    tstep_mac(S[u], act + (size_t)(b + u) * TREC, L, accd, accm);
    if (SEG == 1 || (u & 3) == 3) {
      red_strip[((SEG == 1 ? b + u : (b + u) >> 2) << 6) + lane] = titem_value(accd, accm, L);
      accd = accm = 0.f;
    }
  }
}
template <int SEG>
DEV void tile_strip_range(rsrc_t W, int soff0, const uint8_t* act, int b0, int b1, float* red_strip, const TLane& L, int lane) {
  int b = b0;
  for (; b + TILE_G <= b1; b += TILE_G) tile_group<SEG, TILE_G>(W, soff0, act, b, red_strip, L, lane);
  if (TILE_G > 4 && b + 4 <= b1) { tile_group<SEG, 4>(W, soff0, act, b, red_strip, L, lane); b += 4; }
  if (b < b1) {
    const int cnt = b1 - b;  // 1..3
    TStep S[3];
    float accd = 0.f, accm = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u < cnt) tstep_load(S[u], W, L, soff0 + (b + u) * TILE_B);
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u < cnt) {
        tstep_mac(S[u], act + (size_t)(b + u) * TREC, L, accd, accm);
        if (SEG == 1 || u == cnt - 1) {
          red_strip[((SEG == 1 ? b + u : b >> 2) << 6) + lane] = titem_value(accd, accm, L);
          accd = accm = 0.f;
        }
      }
  }
}
// Items [i0, i1) of a round of strips x IPS items (item i = strip i / IPS, item i % IPS of it): the wave walks the strips
// its range touches.  strip_of(s, W, soff0, act) resolves strip s (wave-uniform); red = the round's partials [item][64];
// done(s, n) is called after n items of strip s have left their partials.
template <int SEG, typename F, typename G>
DEV void tile_items(int i0, int i1, int ips, int nb, float* red, const TLane& L, int lane, F&& strip_of, G&& done) {
  if (i0 >= i1) return;
  int s = i0 / ips, k0 = i0 - s * ips;
  while (i0 < i1) {
    const int k1 = (i1 - i0) + k0 < ips ? (i1 - i0) + k0 : ips;
    rsrc_t W;
    int soff0;
    const uint8_t* act;
    strip_of(s, W, soff0, act);
    const int bb0 = k0 * SEG, bb1 = k1 * SEG < nb ? k1 * SEG : nb;
    tile_strip_range<SEG>(W, soff0, act, bb0, bb1, red + (size_t)s * ips * 64, L, lane);
    done(s, k1 - k0);
    i0 += k1 - k0;
    ++s;
    k0 = 0;
  }
}

// The value of the wave's row (lane & 15) of a strip from its items' partials, by ONE wave (the association of the header):
// lane (n, g) adds P[item][g] of row n in item order, then (S0 + S1) + (S2 + S3) across the four lane groups.  Every lane of a
// row's four returns the value.  The partials are read eight at a time (one LDS round trip per eight).
DEV float tile_strip_value(const float* red_strip, int ips, int lane) {
  float sg = 0.f;
  for (int k0 = 0; k0 < ips; k0 += 8) {
    float p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = red_strip[((k0 + k < ips ? k0 + k : k0) << 6) + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k0 + k < ips) sg += p[k];
  }
  const float a = sg + __shfl_xor(sg, 16);
  return a + __shfl_xor(a, 32);
}

