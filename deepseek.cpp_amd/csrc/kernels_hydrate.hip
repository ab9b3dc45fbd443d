// kernels_hydrate.hip -- batched prompt ingestion ("hydrate"): P prompt tokens through one block with every
// weight matrix read ONCE, instead of the reference's one forward per prompt token (src/main.cpp:312-319 calling
// Model::forward(.., HYDRATE_KV_CACHE), src/infer.cpp:1284-1287).
//
// The contract is BIT-IDENTITY with P single-token dsk_forward calls on a model whose Q2_K matrices are tile records
// (option "q2k_tiles" 2): same KV-cache rows, same residual stream, same routing, same logits.  That works because
//   * the tiled decode kernels define a row's value by an association that does not depend on the launch geometry
//     (tile_device.h: (S0 + S1) + (S2 + S3), S_g = the in-order sum of the item partials fma-chained over the item's blocks),
//     and the GEMM below produces exactly those partials for every (token, row);
//   * Q8_K quantisation is exact arithmetic (IEEE divide / multiply / rint), so any kernel quantising the same floats gives the
//     same codes;
//   * every other float stage (rmsnorm trees, router GEMV, gate, rope, attention, GLU, combine) runs the decode path's own
//     device functions with the decode path's workgroup sizes, once per token.
//
// The GEMM (hyd_gemm_kernel) is a real i8 GEMM on v_mfma_i32_16x16x64_i8, not the decode kernels' selector trick: a lane
// expands its 16 weight bytes to SCALED int8 weights - (q2 & mask) * scale <= 45, a packed 16-bit multiply of four masked
// codes by one scale cannot carry between bytes - so the matrix instruction returns the reference's
// sum_j scale_j * sum_l q8 * q2 directly.  To keep the decode association the sums are kept per sub-block GROUP g4 (sub-blocks
// 4 g4 .. 4 g4 + 3: what one lane of the decode kernel holds): the 16 rows of the activation operand are 4 tokens x 4 groups,
// row (t, g4) holding token t's codes in the K positions of group g4 and zeros elsewhere.  D[(t, g4)][n] is then the exact
// integer the decode lane converts to float, and lane (n, t) of the result holds all four groups of ITS (token, row):
// the float stage - dx * d, the min term, the per-item partials, the final (S0 + S1) + (S2 + S3) - is lane-local.
#include "dsk_internal.h"
#include "tile_device.h"
#include <algorithm>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#ifndef HYD16_MIN_TOKENS
#define HYD16_MIN_TOKENS 12  // plain matrices: from this many tokens on, 16 tokens per wave (hyd_gemm16_kernel)
#endif
#ifndef HYD16_EXPERT_MIN
#define HYD16_EXPERT_MIN 6   // expert stacks: tasks with at least this many rows take the 16-token form (0: never)
#endif
#define HYD_OOB 0x40000000  // a buffer offset beyond any activation array (all far below 1 GiB): the load returns zeros and touches no memory

DEV rsrc_t make_rsrc_n(const void* p, u32 bytes) {
  const unsigned long long v = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
DEV u32 pkmul16(u32 a, u32 b) {  // two independent 16-bit products (v_pk_mul_lo_u16)
  const u16x2 r = __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b);
  return __builtin_bit_cast(u32, r);
}

struct HydTile { u32x4 w; u32 sc[4]; u32 dm; };
DEV void hyd_tile_load(HydTile& T, rsrc_t W, int lane, int soff) {
  const int n = lane & 15;
  T.w = __builtin_amdgcn_raw_buffer_load_b128(W, lane * 16, soff, BUF_NT);
#pragma unroll
  for (int g = 0; g < 4; ++g) T.sc[g] = __builtin_amdgcn_raw_buffer_load_b32(W, 1024 + 4 * (n + 16 * g), soff, BUF_NT);
  T.dm = __builtin_amdgcn_raw_buffer_load_b32(W, 1280 + 4 * n, soff, BUF_NT);
}
// the four B operands of a tile: field s of the lane's 16 bytes (row n = lane & 15, K-group kg = lane >> 4) times the 4-bit scale
// of its sub-block j(kg, s) = 8 (kg >> 1) + 2 s + (kg & 1) = scale byte (2 (s & 1) + (kg & 1)) of word 2 (kg >> 1) + (s >> 1)
DEV void hyd_expand(const HydTile& T, int kg, i32x4 (&B)[4]) {
  const u32 wlo = (kg & 2) ? T.sc[2] : T.sc[0], whi = (kg & 2) ? T.sc[3] : T.sc[1];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const u32 word = (s >> 1) ? whi : wlo;
    const u32 sc = (word >> (8 * (2 * (s & 1) + (kg & 1)))) & 0xFu;
    const u32 pair = sc | (sc << 16);
    B[s].x = (int)pkmul16((T.w.x >> (2 * s)) & 0x03030303u, pair);
    B[s].y = (int)pkmul16((T.w.y >> (2 * s)) & 0x03030303u, pair);
    B[s].z = (int)pkmul16((T.w.z >> (2 * s)) & 0x03030303u, pair);
    B[s].w = (int)pkmul16((T.w.w >> (2 * s)) & 0x03030303u, pair);
  }
}

// float state of one (token, row) per matrix: the running item chains and the four group sums
struct HydAcc {
  float S[4], ad[4], am[4];
};
DEV void hyd_acc_zero(HydAcc& a) {
#pragma unroll
  for (int g = 0; g < 4; ++g) a.S[g] = a.ad[g] = a.am[g] = 0.f;
}
// the activation side of one block for one token quad: the four selector-row operands, the tokens' sub-block sums and scale
struct HydAct {
  i32x4 a[4];
  u32x4 bs0, bs1;  // the token's 16 sub-block sums (result side; DIG: bs0 = the digit words of ONE group, operand side)
  float dx;
};
// DIG: the sums come as digit words (hyd16_split, launch_hyd_digits) and the min term rides the matrix pipe: row (t, g4) of the
// operand holds token t's four words of group g4 in K-group kg == g4 (goff: out of range for the other lanes)
template <bool DIG>
DEV void hyd_act_load(HydAct& X, const rsrc_t& RA, const rsrc_t& RB, const rsrc_t& RD, int off0, int off1, int boff, int doff, int b) {
  X.a[0] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, off0, b * 256, 0));
  X.a[1] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, off0 + 32, b * 256, 0));
  X.a[2] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, off1, b * 256, 0));
  X.a[3] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, off1 + 32, b * 256, 0));
  if (DIG) {
    X.bs0 = __builtin_amdgcn_raw_buffer_load_b128(RB, boff, b * 64, 0);
  } else {
    X.bs0 = __builtin_amdgcn_raw_buffer_load_b128(RB, boff, b * 32, 0);
    X.bs1 = __builtin_amdgcn_raw_buffer_load_b128(RB, boff + 16, b * 32, 0);
  }
  X.dx = u2f(__builtin_amdgcn_raw_buffer_load_b32(RD, doff, b * 4, 0));
}
// one block of one tile for one token quad: D = the four exact group sums of this lane's (token, row)
// the min operand of a tile's weight side: K-group kg carries sub-blocks 4 kg + i as {m, 8 m, 8 m, 0} (against the digit words)
DEV i32x4 hyd_bm(const HydTile& T, int kg) {
  const u32 scm = kg == 0 ? T.sc[0] : kg == 1 ? T.sc[1] : kg == 2 ? T.sc[2] : T.sc[3];
  i32x4 Bm;
  Bm.x = (int)(((scm >> 4) & 0xFu) * 0x00080801u);
  Bm.y = (int)(((scm >> 12) & 0xFu) * 0x00080801u);
  Bm.z = (int)(((scm >> 20) & 0xFu) * 0x00080801u);
  Bm.w = (int)(((scm >> 28) & 0xFu) * 0x00080801u);
  return Bm;
}
template <bool DIG>
DEV void hyd_block(const HydTile& T, const i32x4 (&B)[4], const i32x4& Bm, const HydAct& X, HydAcc& acc) {
  i32x4 D = {0, 0, 0, 0};
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(X.a[0], B[0], D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(X.a[1], B[1], D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(X.a[2], B[2], D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(X.a[3], B[3], D, 0, 0, 0);
  const int Dg[4] = {D.x, D.y, D.z, D.w};
  const float dd = X.dx * h2f(T.dm & 0xffff), dmn = X.dx * h2f(T.dm >> 16);
  if (DIG) {
    // row (t, g4) x K-group g4 against the mins: lane (n, t) gets the four groups' sum_j m_j b_j - exact (hyd16_split)
    const i32x4 z = {0, 0, 0, 0};
    const i32x4 M = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, X.bs0), Bm, z, 0, 0, 0);
    const int Mg[4] = {M.x, M.y, M.z, M.w};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc.ad[g] = fmaf(dd, (float)Dg[g], acc.ad[g]);
      acc.am[g] = fmaf(dmn, (float)Mg[g], acc.am[g]);
    }
    return;
  }
  const u32 bw[8] = {X.bs0.x, X.bs0.y, X.bs0.z, X.bs0.w, X.bs1.x, X.bs1.y, X.bs1.z, X.bs1.w};  // 16 int16 sub-block sums of the token
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const u32 m4 = (T.sc[g] >> 4) & 0x0F0F0F0Fu;                                   // the group's four 4-bit mins
    const u32 hi = __builtin_amdgcn_perm(bw[2 * g + 1], bw[2 * g], 0x07050301u);  // the sums' high bytes (signed)
    const u32 lo = __builtin_amdgcn_perm(bw[2 * g + 1], bw[2 * g], 0x06040200u);  // low bytes (unsigned)
    const int summs = (sdot4(m4, hi, 0) << 8) + (int)__builtin_amdgcn_udot4(m4, lo, 0u, false);
    acc.ad[g] = fmaf(dd, (float)Dg[g], acc.ad[g]);     // tile_device.h tstep_mac, without the in-place mask factors ...
    acc.am[g] = fmaf(dmn, (float)summs, acc.am[g]);
  }
}
DEV void hyd_item_end(HydAcc& acc) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    acc.S[g] += acc.ad[g] - acc.am[g];  // ... which titem_value undoes exactly: fma(4^k accd, 4^-k, -accm) == accd - accm, one rounding
    acc.ad[g] = acc.am[g] = 0.f;
  }
}
DEV float hyd_value(const HydAcc& acc) { return (acc.S[0] + acc.S[1]) + (acc.S[2] + acc.S[3]); }  // tile_strip_value

// One wave = one 16-row strip (GLU: the same strip of w1 and w3) of one task (a plain matrix, or expert `task` of a stack) times
// 4 * NQ of the task's activation rows per pass over the strip's tiles.
//   SPREAD false (expert stacks: a handful of rows per task): the four waves of a workgroup take four consecutive strips and
//                loop over the task's rows, 4 * NQ at a time;
//   SPREAD true  (plain matrices times all P tokens): the waves of a workgroup (up to 8) take the SAME strip and consecutive
//                chunks of 4 * NQ tokens - they stream the same tiles at the same time (one HBM read, L1 hits for the others) and
//                each wave's chain of dependent loads is short.
// Software pipeline of a pass: the tiles of blocks b + 1 and b + 2 and the activation operands of block b + 1 are in flight
// while block b is multiplied (a wave is a chain of dependent memory round trips otherwise: measured 3 - 15x the streaming time).
// Quads past the task's rows read zeros (out-of-range buffer offsets) and are not stored.
template <bool GLU, int NQ, bool SPREAD, bool DIG = false>
__global__ __launch_bounds__(SPREAD && !GLU ? 512 : 256) void hyd_gemm_kernel(const HydGemmArgs A) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nwaves = (int)blockDim.x >> 6;
  const int strips = (A.rows + 15) >> 4;
  const int ntask = A.n_experts > 0 ? A.n_experts : 1;
  const long long unit = SPREAD ? (long long)blockIdx.x : (long long)blockIdx.x * nwaves + wave;
  if (unit >= (long long)ntask * strips) return;
  const int task = (int)(unit / strips), strip = (int)(unit - (long long)task * strips);
  const int n = A.n, nb = n >> 8;
  const size_t woff = (size_t)task * A.e_bytes + (size_t)strip * nb * TILE_B;
  const rsrc_t W1 = make_rsrc(A.W + woff);
  const rsrc_t W3 = make_rsrc(GLU ? A.W3 + woff : A.W + woff);
  // expert stacks: most tasks may be outside this launch's window - look first; plain matrices: the strip's first tiles are requested
  // before anything else
  if (A.count) {
    const int c0 = __builtin_amdgcn_readfirstlane(A.count[task]);
    if (c0 <= 0 || c0 < A.cnt_min || (A.cnt_max > 0 && c0 > A.cnt_max)) return;
  }
  HydTile T1, T3, N1, N3;
  hyd_tile_load(T1, W1, lane, 0);
  if (GLU) hyd_tile_load(T3, W3, lane, 0);
  if (nb > 1) {
    hyd_tile_load(N1, W1, lane, TILE_B);
    if (GLU) hyd_tile_load(N3, W3, lane, TILE_B);
  }
  const int cnt = A.count ? __builtin_amdgcn_readfirstlane(A.count[task]) : A.m;
  if (cnt <= 0 || cnt < A.cnt_min || (A.cnt_max > 0 && cnt > A.cnt_max)) return;  // (a launch may be restricted to tasks with a row count in [cnt_min, cnt_max])
  const int* list = A.list ? A.list + (size_t)task * A.list_stride : nullptr;
  const bool seg4 = nb > 8;  // tile_seg
  const rsrc_t RA = make_rsrc_n(A.a_qs, (u32)((size_t)A.a_rows * n));
  const rsrc_t RB = DIG ? make_rsrc_n(A.a_dig, (u32)((size_t)A.a_rows * (n >> 4) * 4)) : make_rsrc_n(A.a_bsums, (u32)((size_t)A.a_rows * (n >> 4) * 2));
  const rsrc_t RD = make_rsrc_n(A.a_d, (u32)((size_t)A.a_rows * nb * 4));
  // activation operand: this lane is row (tl, g4l) of K-group kg; it holds data for fields 2c, 2c + 1 (c = g4l & 1) when its
  // group's K half is its own
  const int kg = lane >> 4, tl = (lane & 15) >> 2, g4l = lane & 3, c = g4l & 1;
  const bool lane_ok = (g4l >> 1) == (kg >> 1);
  const int jbase = 16 * (8 * (kg >> 1) + (kg & 1));  // byte offset of sub-block j(kg, 0) in a block's codes
  const int qd = lane >> 4, rown = lane & 15;         // result side: token qd of the quad, row rown of the strip
  // (SPREAD: blockIdx.y deals the token chunks of a strip over several workgroups when the matrix has few strips)
  const int first = SPREAD ? ((int)blockIdx.y * nwaves + wave) * 4 * NQ : 0, step = SPREAD ? (int)gridDim.y * nwaves * 4 * NQ : 4 * NQ;
  bool first_pass = true;
  for (int base = first; base < cnt; base += step) {
    int off0[NQ], off1[NQ], boff[NQ], doff[NQ], outv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int eA = base + 4 * q + tl;
      const bool okA = eA < cnt && lane_ok;
      const int vA = eA < cnt ? (list ? list[eA] : eA) : 0;
      const int arow = vA / A.a_div;
      const int ab = arow * n + jbase;
      off0[q] = okA && c == 0 ? ab : HYD_OOB;        // fields 0, 1 (+ 32 bytes)
      off1[q] = okA && c == 1 ? ab + 64 : HYD_OOB;   // fields 2, 3
      const int eD = base + 4 * q + qd;
      const bool okD = eD < cnt;
      const int vD = okD ? (list ? list[eD] : eD) : 0;
      const int arowD = vD / A.a_div;
      // (DIG: the digit words sit on the OPERAND side - row (tl, g4l), K-group g4l - like the codes)
      boff[q] = DIG ? (okA && kg == g4l ? (arow * (n >> 4) + 4 * g4l) * 4 : HYD_OOB) : (okD ? arowD * (n >> 4) * 2 : HYD_OOB);
      doff[q] = okD ? arowD * nb * 4 : HYD_OOB;
      outv[q] = okD ? vD : -1;
    }
    HydAct X[NQ], Y[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) hyd_act_load<DIG>(X[q], RA, RB, RD, off0[q], off1[q], boff[q], doff[q], 0);
    if (!first_pass) {  // a further pass over the same strip (L2 hits)
      hyd_tile_load(T1, W1, lane, 0);
      if (GLU) hyd_tile_load(T3, W3, lane, 0);
      if (nb > 1) {
        hyd_tile_load(N1, W1, lane, TILE_B);
        if (GLU) hyd_tile_load(N3, W3, lane, TILE_B);
      }
    }
    first_pass = false;
    HydAcc acc1[NQ], acc3[GLU ? NQ : 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { hyd_acc_zero(acc1[q]); if (GLU) hyd_acc_zero(acc3[q]); }
    for (int b = 0; b < nb; ++b) {
      HydTile M1, M3;
      if (b + 2 < nb) {
        hyd_tile_load(M1, W1, lane, (b + 2) * TILE_B);
        if (GLU) hyd_tile_load(M3, W3, lane, (b + 2) * TILE_B);
      }
      if (b + 1 < nb) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) hyd_act_load<DIG>(Y[q], RA, RB, RD, off0[q], off1[q], boff[q], doff[q], b + 1);
      }
      const bool item_end = !seg4 || (b & 3) == 3 || b == nb - 1;  // tile_device.h: items of 4 blocks, or 1 for rows of <= 8
      {
        i32x4 B[4], Bm = {0, 0, 0, 0};
        hyd_expand(T1, kg, B);
        if (DIG) Bm = hyd_bm(T1, kg);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          hyd_block<DIG>(T1, B, Bm, X[q], acc1[q]);
          if (item_end) hyd_item_end(acc1[q]);
        }
        if (GLU) {
          hyd_expand(T3, kg, B);
          if (DIG) Bm = hyd_bm(T3, kg);
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            hyd_block<DIG>(T3, B, Bm, X[q], acc3[q]);
            if (item_end) hyd_item_end(acc3[q]);
          }
        }
      }
      T1 = N1; N1 = M1;
      if (GLU) { T3 = N3; N3 = M3; }
#pragma unroll
      for (int q = 0; q < NQ; ++q) X[q] = Y[q];
    }
    const int row = strip * 16 + rown;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (outv[q] >= 0 && row < A.rows) {
        float* o = A.out + (size_t)outv[q] * A.out_stride + row;
        const float v = hyd_value(acc1[q]);
        if (GLU) *o = act_fn(v, A.act) * hyd_value(acc3[q]);   // src/infer.cpp:859-872
        else if (A.epilogue == EPI_ADD) *o += v;                // src/infer.cpp:832-834, 928-930
        else *o = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Plain matrices times MANY tokens: 16 tokens per wave and pass, the matrix instruction's whole N side.
// The K side of one instruction is one sub-block GROUP g (sub-blocks 4 g .. 4 g + 3: what the float stage keeps apart), so
// D_g[token][row] is the exact group sum with no zero padding: K-group kg' of lane (row n, kg') carries sub-block 4 g + kg' =
// field 2 (g & 1) + (kg' >> 1) of the qs bytes of K-group 2 (g >> 1) + (kg' & 1) - the lane's own 16 bytes or those of the lane
// 32 away (one v_permlane32_swap per dword gives both) - scaled as in hyd_expand; the activation operand of lane (token t, kg') is
// the token's 16 codes of that sub-block in natural order: four plain 16-byte loads per block, no selector rows.
// The min term rides the matrix pipe too: a sub-block sum b (|b| <= 2032) is split b = 8 (v1 + v2) + v0 with all three in int8 range,
// and K slot (sub-block j, {v0, v1, v2, 0}) meets the weight side's {m_j, 8 m_j, 8 m_j, 0} (8 m_j <= 120): the instruction returns
// sum_j m_j b_j exactly; one instruction per group (the activation side masked to the group's lanes).
// In the C/D layout lane (n, q) holds tokens 4 q .. 4 q + 3 of its row: the float stage (hyd_block's, per token) is lane-local.
// Same integers, same float chains as hyd_gemm_kernel: same bits.
// ------------------------------------------------------------------------------------
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
struct Hyd16Acc { float S[4][4], ad[4][4], am[4][4]; };  // [token of the lane's four][group]
DEV void hyd16_acc_zero(Hyd16Acc& a) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) a.S[r][g] = a.ad[r][g] = a.am[r][g] = 0.f;
}
struct Hyd16Act {
  i32x4 a[4];     // the token's codes of sub-block 4 g + kg'
  u32x2_t bs;     // the token's sums of sub-blocks 4 kg' .. 4 kg' + 3 (int16)
  float dx[4];    // result side: the scales of tokens 4 q .. 4 q + 3
};
DEV void hyd16_act_load(Hyd16Act& X, const rsrc_t& RA, const rsrc_t& RB, const rsrc_t& RD, int offA, int offB, const int (&offD)[4], int b) {
#pragma unroll
  for (int g = 0; g < 4; ++g) X.a[g] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, offA + 64 * g, b * 256, 0));
  X.bs = __builtin_amdgcn_raw_buffer_load_b64(RB, offB, b * 32, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) X.dx[r] = u2f(__builtin_amdgcn_raw_buffer_load_b32(RD, offD[r], b * 4, 0));
}
// the weight side of a tile: four scaled group operands and the min operand
DEV void hyd16_expand(const HydTile& T, int kgp, i32x4 (&B)[4], i32x4& Bm) {
  const u32 w[4] = {T.w.x, T.w.y, T.w.z, T.w.w};
  u32 lo[4], hi[4];  // lo: lanes 0-31 their own dword, lanes 32-63 that of lane - 32 (groups 0, 1); hi: lanes 0-31 that of lane + 32 (groups 2, 3)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const auto sw = __builtin_amdgcn_permlane32_swap(w[i], w[i], false, false);
    lo[i] = sw[0];
    hi[i] = sw[1];
  }
  const int fs = kgp >> 1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int sh = 2 * (2 * (g & 1) + fs);
    const u32 sc = (T.sc[g] >> (8 * kgp)) & 0xFu;
    const u32 pair = sc | (sc << 16);
    const u32* src = (g >> 1) ? hi : lo;
    B[g].x = (int)pkmul16((src[0] >> sh) & 0x03030303u, pair);
    B[g].y = (int)pkmul16((src[1] >> sh) & 0x03030303u, pair);
    B[g].z = (int)pkmul16((src[2] >> sh) & 0x03030303u, pair);
    B[g].w = (int)pkmul16((src[3] >> sh) & 0x03030303u, pair);
  }
  const u32 scm = kgp == 0 ? T.sc[0] : kgp == 1 ? T.sc[1] : kgp == 2 ? T.sc[2] : T.sc[3];  // the lane's own group's scale | min bytes
  Bm.x = (int)(((scm >> 4) & 0xFu) * 0x00080801u);
  Bm.y = (int)(((scm >> 12) & 0xFu) * 0x00080801u);
  Bm.z = (int)(((scm >> 20) & 0xFu) * 0x00080801u);
  Bm.w = (int)(((scm >> 28) & 0xFu) * 0x00080801u);
}
DEV u32 hyd16_split(int b) {  // b = 8 (v1 + v2) + v0: bytes {v0, v1, v2, 0}
  const int v0 = b & 7, q = b >> 3, v1 = q >> 1, v2 = q - v1;
  return (u32)v0 | (((u32)v1 & 0xFFu) << 8) | (((u32)v2 & 0xFFu) << 16);
}
// am4: the min term's activation operand of this lane: the digit words of sub-blocks 4 kg' + i of its token
DEV void hyd16_block_d(const HydTile& T, const i32x4 (&B)[4], const i32x4& Bm, const Hyd16Act& X, const i32x4& am4, int kgp, Hyd16Acc& acc) {
  const float d = h2f(T.dm & 0xffff), dmin = h2f(T.dm >> 16);
  float dd[4], dmn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { dd[r] = X.dx[r] * d; dmn[r] = X.dx[r] * dmin; }
  // all eight matrix instructions first (independent: they pipeline back to back), then the float stage
  const i32x4 z = {0, 0, 0, 0};
  i32x4 D[4], M[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) D[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X.a[g], B[g], z, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 4; ++g) M[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kgp == g ? am4 : z, Bm, z, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int Dr[4] = {D[g].x, D[g].y, D[g].z, D[g].w}, Mr[4] = {M[g].x, M[g].y, M[g].z, M[g].w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc.ad[r][g] = fmaf(dd[r], (float)Dr[r], acc.ad[r][g]);
      acc.am[r][g] = fmaf(dmn[r], (float)Mr[r], acc.am[r][g]);
    }
  }
}
DEV void hyd16_block(const HydTile& T, const i32x4 (&B)[4], const i32x4& Bm, const Hyd16Act& X, int kgp, Hyd16Acc& acc) {
  i32x4 am4;
  am4.x = (int)hyd16_split((int)(short)(X.bs.x & 0xFFFFu));
  am4.y = (int)hyd16_split((int)(short)(X.bs.x >> 16));
  am4.z = (int)hyd16_split((int)(short)(X.bs.y & 0xFFFFu));
  am4.w = (int)hyd16_split((int)(short)(X.bs.y >> 16));
  hyd16_block_d(T, B, Bm, X, am4, kgp, acc);
}
DEV void hyd16_item_end(Hyd16Acc& acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc.S[r][g] += acc.ad[r][g] - acc.am[r][g];
      acc.ad[r][g] = acc.am[r][g] = 0.f;
    }
}
DEV float hyd16_value(const Hyd16Acc& acc, int r) { return (acc.S[r][0] + acc.S[r][1]) + (acc.S[r][2] + acc.S[r][3]); }

// The workgroup: up to 8 waves = 8 consecutive STRIPS times the SAME 16 tokens (blockIdx.y = the token chunk).  The tokens' operands
// of HYD16_SB blocks at a time are staged once per workgroup in LDS (double-buffered, one barrier per stage) and every wave reads its
// matrix operands from there: a wave's vector-memory traffic is its own tiles only.  (First form, measured: the waves of a workgroup
// on different token chunks of one strip, each loading its 4.4 KB of operands per block from L2 - SQ counters: 51 % of the wave time
// parked on s_waitcnt, VALU 32 %: 46 KB per CU and block through a memory path that carries ~27 GB/s per CU; wo 108 us at P = 64.)
#define HYD16_SB 4          // blocks per stage
#define HYD16_TOK_B 272     // LDS bytes of a token's codes of one block (256 + 16: the 16 tokens of an operand read spread over the banks)
#define HYD16_STAGE_B (HYD16_SB * (16 * HYD16_TOK_B + 16 * 64 + 16 * 4))   // codes | sub-block sums as digit words (hyd16_split) | scales, per block
template <int R> struct Hyd16Stage { u32x4 c[R]; u32x4 bs; u32x4 d; };
// Plain matrices: blockIdx.x = strip group, blockIdx.y = the chunk of 16 tokens.  Expert stacks (A.n_experts > 0: the task's rows come
// through its list, src/infer.cpp:853-878): blockIdx.x = task x strip groups, the workgroup walks the task's rows 16 at a time (a
// launch may be restricted to tasks with a row count in [cnt_min, cnt_max]: the few-row tasks stay with hyd_gemm_kernel's quads).
template <bool GLU, int NW>
__global__ __launch_bounds__(64 * NW) void hyd_gemm16_kernel(const HydGemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // 2 x HYD16_STAGE_B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwaves = NW, nthreads = 64 * NW;
  constexpr int R = HYD16_SB * 256 / nthreads;  // 16-byte pieces of a stage's codes per thread
  const int strips = (A.rows + 15) >> 4;
  const int groups = (strips + nwaves - 1) / nwaves;
  const int task = A.n_experts > 0 ? (int)blockIdx.x / groups : 0;
  const int group = (int)blockIdx.x - task * groups;
  const int cnt = A.count ? __builtin_amdgcn_readfirstlane(A.count[task]) : A.m;
  if (cnt <= 0 || cnt < A.cnt_min || (A.cnt_max > 0 && cnt > A.cnt_max)) return;
  const int* list = A.list ? A.list + (size_t)task * A.list_stride : nullptr;
  const int strip = group * nwaves + wave;
  const bool live = strip < strips;  // (a wave without a strip still stages and keeps the barriers)
  const int n = A.n, nb = n >> 8;
  const size_t woff = (size_t)task * A.e_bytes + (size_t)(live ? strip : 0) * nb * TILE_B;
  const rsrc_t W1 = make_rsrc(A.W + woff);
  const rsrc_t W3 = make_rsrc(GLU ? A.W3 + woff : A.W + woff);
  const bool seg4 = nb > 8;  // tile_seg
  const rsrc_t RA = make_rsrc_n(A.a_qs, (u32)((size_t)A.a_rows * n));
  const rsrc_t RB = make_rsrc_n(A.a_bsums, (u32)((size_t)A.a_rows * (n >> 4) * 2));
  const rsrc_t RD = make_rsrc_n(A.a_d, (u32)((size_t)A.a_rows * nb * 4));
  const int kgp = lane >> 4, tl = lane & 15;   // operand side: token tl of the chunk, K-group kg'
  const int q = lane >> 4, rown = lane & 15;   // result side: tokens 4 q .. 4 q + 3, row rown of the strip
  auto c_sb = [&](int k) { return (tid + k * nthreads) >> 8; };
  auto c_lds = [&](int k) { const int idx = tid + k * nthreads; return ((idx >> 8) * 16 + ((idx >> 4) & 15)) * HYD16_TOK_B + (idx & 15) * 16; };
  const int b_sb = tid >> 5, b_t = (tid >> 1) & 15, b_piece = tid & 1;
  const int b_lds = HYD16_SB * 16 * HYD16_TOK_B + (b_sb * 16 + b_t) * 64 + b_piece * 32;   // (8 sums -> 8 digit words)
  const int d_lds = HYD16_SB * (16 * HYD16_TOK_B + 16 * 64);
  auto arow_of = [&](int e) { return (list ? list[e] : e) / A.a_div; };  // (e < cnt)
  for (int base = (int)blockIdx.y * 16; base < cnt; base += (int)gridDim.y * 16) {
    // ---- staging: thread -> its pieces of a stage (wave-uniform loop bounds; a row past the count reads zeros) ----
    // codes: HYD16_SB x 16 tokens x 16 pieces of 16 bytes; sums: HYD16_SB x 16 x 2 pieces; scales: 16 tokens x (HYD16_SB floats = 16 bytes)
    int c_off[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int idx = tid + k * nthreads;
      const int t = (idx >> 4) & 15, piece = idx & 15;
      const int e = base + t;
      c_off[k] = e < cnt ? arow_of(e) * n + piece * 16 : HYD_OOB;
    }
    const int b_off = (base + b_t < cnt) ? arow_of(base + b_t) * (n >> 4) * 2 + b_piece * 16 : HYD_OOB;
    const int d_off = (tid < 16 && base + tid < cnt) ? arow_of(base + tid) * nb * 4 : HYD_OOB;
    auto stage_request = [&](Hyd16Stage<R>& G, int b0) {
#pragma unroll
      for (int k = 0; k < R; ++k) G.c[k] = __builtin_amdgcn_raw_buffer_load_b128(RA, c_off[k], (b0 + c_sb(k)) * 256, 0);
      if (tid < HYD16_SB * 32) G.bs = __builtin_amdgcn_raw_buffer_load_b128(RB, b_off, (b0 + b_sb) * 32, 0);
      if (tid < 16) {  // the token's scales of blocks b0 .. b0 + 3 (past the row's end: never multiplied)
        G.d.x = __builtin_amdgcn_raw_buffer_load_b32(RD, d_off, (b0 + 0) * 4, 0);
        G.d.y = b0 + 1 < nb ? __builtin_amdgcn_raw_buffer_load_b32(RD, d_off, (b0 + 1) * 4, 0) : 0u;
        G.d.z = b0 + 2 < nb ? __builtin_amdgcn_raw_buffer_load_b32(RD, d_off, (b0 + 2) * 4, 0) : 0u;
        G.d.w = b0 + 3 < nb ? __builtin_amdgcn_raw_buffer_load_b32(RD, d_off, (b0 + 3) * 4, 0) : 0u;
      }
    };
    auto stage_write = [&](const Hyd16Stage<R>& G, uint8_t* buf) {
#pragma unroll
      for (int k = 0; k < R; ++k) *reinterpret_cast<u32x4*>(buf + c_lds(k)) = G.c[k];
      if (tid < HYD16_SB * 32) {  // the sums go to LDS as the min term's operand words: split ONCE per workgroup, not per wave and block
        const u32 w[4] = {G.bs.x, G.bs.y, G.bs.z, G.bs.w};
        u32x4 lo, hi;
        lo.x = hyd16_split((int)(short)(w[0] & 0xFFFFu)); lo.y = hyd16_split((int)(short)(w[0] >> 16));
        lo.z = hyd16_split((int)(short)(w[1] & 0xFFFFu)); lo.w = hyd16_split((int)(short)(w[1] >> 16));
        hi.x = hyd16_split((int)(short)(w[2] & 0xFFFFu)); hi.y = hyd16_split((int)(short)(w[2] >> 16));
        hi.z = hyd16_split((int)(short)(w[3] & 0xFFFFu)); hi.w = hyd16_split((int)(short)(w[3] >> 16));
        *reinterpret_cast<u32x4*>(buf + b_lds) = lo;
        *reinterpret_cast<u32x4*>(buf + b_lds + 16) = hi;
      }
      if (tid < 16) {
        float* dl = reinterpret_cast<float*>(buf + d_lds);
        dl[0 * 16 + tid] = u2f(G.d.x); dl[1 * 16 + tid] = u2f(G.d.y); dl[2 * 16 + tid] = u2f(G.d.z); dl[3 * 16 + tid] = u2f(G.d.w);
      }
    };
    HydTile T1, T3, N1, N3;
    if (live) {  // (a further pass over the same strip: L2 hits)
      hyd_tile_load(T1, W1, lane, 0);
      if (GLU) hyd_tile_load(T3, W3, lane, 0);
      if (nb > 1) {
        hyd_tile_load(N1, W1, lane, TILE_B);
        if (GLU) hyd_tile_load(N3, W3, lane, TILE_B);
      }
    }
    Hyd16Stage<R> G;
    stage_request(G, 0);
    stage_write(G, lds);
    __syncthreads();
    Hyd16Acc acc1, acc3;
    hyd16_acc_zero(acc1);
    if (GLU) hyd16_acc_zero(acc3);
    for (int b0 = 0; b0 < nb; b0 += HYD16_SB) {
      uint8_t* cur = lds + ((b0 / HYD16_SB) & 1) * HYD16_STAGE_B;
      uint8_t* nxt = lds + (((b0 / HYD16_SB) & 1) ^ 1) * HYD16_STAGE_B;
      const bool more = b0 + HYD16_SB < nb;
      if (more) stage_request(G, b0 + HYD16_SB);
      if (live) {
#pragma unroll
        for (int u = 0; u < HYD16_SB; ++u) {
          const int b = b0 + u;
          if (b < nb) {
            HydTile M1, M3;
            if (b + 2 < nb) {
              hyd_tile_load(M1, W1, lane, (b + 2) * TILE_B);
              if (GLU) hyd_tile_load(M3, W3, lane, (b + 2) * TILE_B);
            }
            Hyd16Act X;
            const uint8_t* cb = cur + (u * 16 + tl) * HYD16_TOK_B + 16 * kgp;
#pragma unroll
            for (int g = 0; g < 4; ++g) X.a[g] = *reinterpret_cast<const i32x4*>(cb + 64 * g);
            const i32x4 am4 = *reinterpret_cast<const i32x4*>(cur + HYD16_SB * 16 * HYD16_TOK_B + (u * 16 + tl) * 64 + 16 * kgp);
            const f32x4 dq = *reinterpret_cast<const f32x4*>(cur + d_lds + (u * 16 + 4 * q) * 4);
            X.dx[0] = dq.x; X.dx[1] = dq.y; X.dx[2] = dq.z; X.dx[3] = dq.w;
            const bool item_end = !seg4 || (b & 3) == 3 || b == nb - 1;
            i32x4 B[4], Bm;
            hyd16_expand(T1, kgp, B, Bm);
            hyd16_block_d(T1, B, Bm, X, am4, kgp, acc1);
            if (item_end) hyd16_item_end(acc1);
            if (GLU) {
              hyd16_expand(T3, kgp, B, Bm);
              hyd16_block_d(T3, B, Bm, X, am4, kgp, acc3);
              if (item_end) hyd16_item_end(acc3);
            }
            T1 = N1; N1 = M1;
            if (GLU) { T3 = N3; N3 = M3; }
          }
        }
      }
      if (more) stage_write(G, nxt);
      __syncthreads();  // (also in front of the next chunk's first stage: every wave is done with both buffers)
    }
    if (live) {
      const int row = strip * 16 + rown;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int eD = base + 4 * q + r;
        if (eD < cnt && row < A.rows) {
          const int vD = list ? list[eD] : eD;
          float* o = A.out + (size_t)vD * A.out_stride + row;
          const float v = hyd16_value(acc1, r);
          if (GLU) *o = act_fn(v, A.act) * hyd16_value(acc3, r);   // src/infer.cpp:859-872
          else if (A.epilogue == EPI_ADD) *o += v;                  // src/infer.cpp:832-834, 928-930
          else *o = v;
        }
      }
    }
  }
}

// Long rows, few strips (the first-stage projections: 96 + 36 strips of 28 blocks; the shared expert's w1 / w3): a strip's row is a
// chain of 28 dependent block steps whatever the grid - 25 us per launch at P = 64 with most of the chip idle.  The association of
// tile_device.h is made for this: a row's value is the in-order sum of its ITEM partials (4 blocks each), so the items of one
// strip x 16 tokens go to the WAVES of a workgroup (wave w = item w), every wave leaves ad - am of its item in LDS and wave 0 adds
// them in item order - the additions hyd16_item_end performs, in the same order: same bits.  Operands straight from L2 (each wave
// reads other blocks: nothing to share).
template <bool GLU>
__global__ __launch_bounds__(512) void hyd_gemm16k_kernel(const HydGemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // partials [matrix][item][value 0..15][lane]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nitems = (int)blockDim.x >> 6;
  const int strip = (int)blockIdx.x;
  const int n = A.n, nb = n >> 8;
  const int b_lo = wave * 4, b_hi = b_lo + 4 < nb ? b_lo + 4 : nb;
  const size_t woff = (size_t)strip * nb * TILE_B;
  const rsrc_t W1 = make_rsrc(A.W + woff);
  const rsrc_t W3 = make_rsrc(GLU ? A.W3 + woff : A.W + woff);
  HydTile T1, T3, N1, N3;
  hyd_tile_load(T1, W1, lane, b_lo * TILE_B);
  if (GLU) hyd_tile_load(T3, W3, lane, b_lo * TILE_B);
  if (b_lo + 1 < b_hi) {
    hyd_tile_load(N1, W1, lane, (b_lo + 1) * TILE_B);
    if (GLU) hyd_tile_load(N3, W3, lane, (b_lo + 1) * TILE_B);
  }
  const int cnt = A.m;
  const rsrc_t RA = make_rsrc_n(A.a_qs, (u32)((size_t)A.a_rows * n));
  const rsrc_t RB = make_rsrc_n(A.a_bsums, (u32)((size_t)A.a_rows * (n >> 4) * 2));
  const rsrc_t RD = make_rsrc_n(A.a_d, (u32)((size_t)A.a_rows * nb * 4));
  const int kgp = lane >> 4, tl = lane & 15, q = lane >> 4, rown = lane & 15;
  const int base = (int)blockIdx.y * 16;
  const int eA = base + tl;
  const bool okA = eA < cnt;
  const int arow = okA ? eA / A.a_div : 0;
  const int offA = okA ? arow * n + 16 * kgp : HYD_OOB;
  const int offB = okA ? arow * (n >> 4) * 2 + 8 * kgp : HYD_OOB;
  int offD[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int eD = base + 4 * q + r;
    offD[r] = eD < cnt ? (eD / A.a_div) * nb * 4 : HYD_OOB;
  }
  Hyd16Act X, Y;
  hyd16_act_load(X, RA, RB, RD, offA, offB, offD, b_lo);
  Hyd16Acc acc1, acc3;
  hyd16_acc_zero(acc1);
  if (GLU) hyd16_acc_zero(acc3);
  for (int b = b_lo; b < b_hi; ++b) {
    HydTile M1, M3;
    if (b + 2 < b_hi) {
      hyd_tile_load(M1, W1, lane, (b + 2) * TILE_B);
      if (GLU) hyd_tile_load(M3, W3, lane, (b + 2) * TILE_B);
    }
    if (b + 1 < b_hi) hyd16_act_load(Y, RA, RB, RD, offA, offB, offD, b + 1);
    i32x4 B[4], Bm;
    hyd16_expand(T1, kgp, B, Bm);
    hyd16_block(T1, B, Bm, X, kgp, acc1);
    if (GLU) {
      hyd16_expand(T3, kgp, B, Bm);
      hyd16_block(T3, B, Bm, X, kgp, acc3);
    }
    T1 = N1; N1 = M1;
    if (GLU) { T3 = N3; N3 = M3; }
    X = Y;
  }
  float* part = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      part[((size_t)wave * 16 + r * 4 + g) * 64 + lane] = acc1.ad[r][g] - acc1.am[r][g];
      if (GLU) part[((size_t)(nitems + wave) * 16 + r * 4 + g) * 64 + lane] = acc3.ad[r][g] - acc3.am[r][g];
    }
  __syncthreads();
  if (wave != 0) return;
  float S1[4][4], S3[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) { S1[r][g] = 0.f; S3[r][g] = 0.f; }
  for (int it = 0; it < nitems; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        S1[r][g] += part[((size_t)it * 16 + r * 4 + g) * 64 + lane];
        if (GLU) S3[r][g] += part[((size_t)(nitems + it) * 16 + r * 4 + g) * 64 + lane];
      }
  const int row = strip * 16 + rown;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int eD = base + 4 * q + r;
    if (eD < cnt && row < A.rows) {
      float* o = A.out + (size_t)eD * A.out_stride + row;
      const float v = (S1[r][0] + S1[r][1]) + (S1[r][2] + S1[r][3]);
      if (GLU) *o = act_fn(v, A.act) * ((S3[r][0] + S3[r][1]) + (S3[r][2] + S3[r][3]));   // src/infer.cpp:859-872
      else if (A.epilogue == EPI_ADD) *o += v;                                              // src/infer.cpp:832-834, 928-930
      else *o = v;
    }
  }
}

template <bool GLU, int NQ>
static void hyd_gemm_launch(hipStream_t st, const HydGemmArgs& A) {
  const long long units = (long long)(A.n_experts > 0 ? A.n_experts : 1) * ((A.rows + 15) >> 4);
  // expert stacks: the tasks with HYD16_EXPERT_MIN rows or more go through the 16-token form (their rows share the expanded tile 16 at
  // a time instead of 4), the others stay with the quads below
  if (A.n_experts > 0 && A.count && A.list && HYD16_EXPERT_MIN > 0 && A.cnt_min == 0 && A.cnt_max == 0) {
    HydGemmArgs B = A;
    B.cnt_min = HYD16_EXPERT_MIN;
    const int strips = (A.rows + 15) >> 4;
    constexpr int NW = GLU ? 4 : 8;
    const int groups = (strips + NW - 1) / NW;
    hipLaunchKernelGGL((hyd_gemm16_kernel<GLU, NW>), dim3((unsigned)(A.n_experts * groups), 1u), dim3(64 * NW), 2 * HYD16_STAGE_B, st, B);
    HydGemmArgs C = A;
    C.cnt_max = HYD16_EXPERT_MIN - 1;
    if (C.a_dig) hipLaunchKernelGGL((hyd_gemm_kernel<GLU, NQ, false, true>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, C);
    else hipLaunchKernelGGL((hyd_gemm_kernel<GLU, NQ, false>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, C);
    return;
  }
  // plain matrices times many tokens: 16 tokens per wave (hyd_gemm16_kernel), the waves of a workgroup on consecutive chunks of one strip
  if (A.n_experts == 0 && A.list == nullptr && A.m >= HYD16_MIN_TOKENS) {
    const int chunks = (A.m + 15) / 16;
    const int nb_ = A.n >> 8, ips_ = tile_ips(nb_);
    if (nb_ > 8 && ips_ <= 8 && units * chunks <= 768) {  // long rows, few strips: the items of a strip to the waves of a workgroup
      const int lds_k = (GLU ? 2 : 1) * ips_ * 16 * 64 * 4;  // (64 KB for a GLU pair of 8 items: above the default dynamic limit's comfort)
      if (lds_k > 48 * 1024) hipFuncSetAttribute((const void*)hyd_gemm16k_kernel<GLU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k);
      hipLaunchKernelGGL((hyd_gemm16k_kernel<GLU>), dim3((unsigned)units, (unsigned)chunks), dim3(64 * ips_), lds_k, st, A);
      return;
    }
    // strips per workgroup: as many as still give most CUs a workgroup (the staging is paid once per workgroup, and every choice
    // puts the same number of waves on the chip); at least 2 (the staging indices assume >= 128 threads)
    int nw = GLU ? 4 : 8;
    while (nw > 2 && ((units + nw - 1) / nw) * chunks < 192) nw >>= 1;
    const dim3 grid((unsigned)((units + nw - 1) / nw), (unsigned)chunks);
    if (nw == 8) hipLaunchKernelGGL((hyd_gemm16_kernel<GLU, 8>), grid, dim3(512), 2 * HYD16_STAGE_B, st, A);
    else if (nw == 4) hipLaunchKernelGGL((hyd_gemm16_kernel<GLU, 4>), grid, dim3(256), 2 * HYD16_STAGE_B, st, A);
    else hipLaunchKernelGGL((hyd_gemm16_kernel<GLU, 2>), grid, dim3(128), 2 * HYD16_STAGE_B, st, A);
    return;
  }
  // plain matrices with more rows of activations than one wave takes per pass: the waves of a workgroup share a strip
  if (A.n_experts == 0 && A.m > 4 * NQ) {
    const int chunks = (A.m + 4 * NQ - 1) / (4 * NQ);
    int nw = chunks;
    const int maxw = GLU ? 4 : 8;  // (a GLU pair holds two matrices' tiles: 4 waves keep 512 registers per lane in reach)
    if (nw > maxw) nw = maxw;
    // a matrix of few strips (the first-stage projections: 96 + 36) would occupy a fraction of the chip with 8-wave workgroups:
    // its chunks are dealt over several smaller workgroups per strip instead (they still stream the strip at about the same time)
    int ny = 1;
    while (units * ny < 256 && nw > 1) { nw = (nw + 1) / 2; ny = (chunks + nw - 1) / nw; }
    hipLaunchKernelGGL((hyd_gemm_kernel<GLU, NQ, true>), dim3((unsigned)units, (unsigned)ny), dim3(64 * nw), 0, st, A);
  } else {
    hipLaunchKernelGGL((hyd_gemm_kernel<GLU, NQ, false>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, A);
  }
}
// sub-block sums -> digit words (hyd16_split): the operand of the quads' min-term matrix instruction
__global__ __launch_bounds__(256) void hyd_digits_kernel(const int16_t* __restrict__ bsums, u32* __restrict__ dig, size_t count) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) dig[i] = hyd16_split((int)bsums[i]);
}
int launch_hyd_digits(hipStream_t st, const int16_t* bsums, unsigned* dig, size_t count) {
  hipLaunchKernelGGL(hyd_digits_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, bsums, dig, count);
  return DSK_OK;
}
int launch_hyd_gemm(hipStream_t st, const HydGemmArgs& A, int nq) {
  if (A.n % 256 || A.rows < 1 || A.a_rows < 1 || A.a_div < 1) DSK_FAIL(DSK_ERR_INVALID, "hyd_gemm: bad shape");
  if ((size_t)A.a_rows * A.n >= (size_t)HYD_OOB) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_gemm: activation array too large for 30-bit offsets");
  const bool glu = A.W3 != nullptr;
  // (16 tokens per wave and pass - NQ 4 - needs 260 / 366 registers with the pipeline's double buffers: not built)
  if (nq >= 2) { if (glu) hyd_gemm_launch<true, 2>(st, A); else hyd_gemm_launch<false, 2>(st, A); }
  else { if (glu) hyd_gemm_launch<true, 1>(st, A); else hyd_gemm_launch<false, 1>(st, A); }
  return DSK_OK;
}

// ------------------------------------------------------------------------------------
// rmsnorm + Q8_K of P rows (src/infer.cpp:601-611, quant.cpp:616-653) with the staging routine - and therefore the
// sum-of-squares tree - of the decode launch that consumes the vector (gemv_device.h stage_q8 at the launch's NW)
// ------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void hyd_norm_q8_kernel(const float* __restrict__ X, int n, const float* __restrict__ norm_w, float eps,
                                                                int8_t* __restrict__ qs, float* __restrict__ d, int16_t* __restrict__ bsums) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  const int p = blockIdx.x, tid = threadIdx.x;
  ActSrc S;
  S.act_mode = ACT_F32_NORM; S.n = n; S.eps = eps; S.pre_scale = 0.f;
  S.a_f32 = X + (size_t)p * n; S.norm_w = norm_w;
  S.a_qs = nullptr; S.a_d = nullptr; S.a_bsums = nullptr;
  stage_q8<LAY_TILE, NW>(S, smem, tid, scratch);
  __syncthreads();
  dump_staged_q8<LAY_TILE>(smem, n, qs + (size_t)p * n, d + (size_t)p * (n >> 8), tid, NW * 64);
  for (int i = tid; i < (n >> 4); i += NW * 64) {
    const int b = i >> 4, j = i & 15;
    const uint8_t* rec = smem + (size_t)b * TREC;
    const int hi = (int8_t)rec[TREC_BS + 8 * (j >> 2) + (j & 3)], lo = rec[TREC_BS + 4 + 8 * (j >> 2) + (j & 3)];
    bsums[(size_t)p * (n >> 4) + i] = (int16_t)((hi << 8) | lo);
  }
}
int launch_hyd_norm_q8(hipStream_t st, int NW, const float* X, int P, int n, const float* norm_w, float eps, int8_t* qs, float* d, int16_t* bsums) {
  const size_t lds = (size_t)(n >> 8) * TREC;
  if (n % 256 || lds > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_norm_q8: n=%d", n);
#define HN(NWV)                                                                                                         \
  do {                                                                                                                  \
    auto k = hyd_norm_q8_kernel<NWV>;                                                                                   \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    hipLaunchKernelGGL(k, dim3(P), dim3(NWV * 64), lds, st, X, n, norm_w, eps, qs, d, bsums);                           \
  } while (0)
  if (NW == 16) HN(16);
  else if (NW == 8) HN(8);
  else if (NW == 4) HN(4);
  else DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_norm_q8: %d waves", NW);
#undef HN
  return DSK_OK;
}

// ------------------------------------------------------------------------------------
// the two latents of the MHA path, normed and quantised as head_attn_kernel does it (kernels_gemv.hip latent_finish: wave w
// owns block w of q_a || kv_a[:lora], the sums of squares meet in LDS in block order), once per token instead of once per head
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hyd_latent_q8_kernel(const HydLatentArgs A) {
  __shared__ float scratch[16];
  const int p = blockIdx.x, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nbq = A.nq >> 8, nbkv = A.nkv >> 8;
  const bool mine = wave < nbq + nbkv, is_q = wave < nbq;
  const int b = mine ? (is_q ? wave : wave - nbq) : 0;
  const float* src = is_q ? A.q_a + (size_t)p * A.q_stride : A.kv_a + (size_t)p * A.kv_stride;
  const float* nw = is_q ? A.q_norm : A.kv_norm;
  const f32x4 t = *reinterpret_cast<const f32x4*>(src + b * 256 + lane * 4);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(nw + b * 256 + lane * 4);
  float ss = fmaf(t.x, t.x, fmaf(t.y, t.y, fmaf(t.z, t.z, t.w * t.w)));
  ss = wave_sum(ss);
  if (lane == 0) scratch[wave] = mine ? ss : 0.f;
  __syncthreads();
  float total = 0.f;
  if (is_q) for (int i = 0; i < nbq; ++i) total += scratch[i];
  else for (int i = nbq; i < nbq + nbkv; ++i) total += scratch[i];
  if (!mine) return;
  const int nn = is_q ? A.nq : A.nkv;
  const float scale = 1.0f / sqrtf(total / (float)nn + A.eps);
  const float v[4] = {t.x * scale * wv.x, t.y * scale * wv.y, t.z * scale * wv.z, t.w * scale * wv.w};
  if (is_q) ad::q8k_block(v, lane, A.qq_qs + (size_t)p * A.nq + b * 256, A.qq_d + (size_t)p * nbq + b, A.qq_bsums + (size_t)p * (A.nq >> 4) + b * 16);
  else ad::q8k_block(v, lane, A.kq_qs + (size_t)p * A.nkv + b * 256, A.kq_d + (size_t)p * nbkv + b, A.kq_bsums + (size_t)p * (A.nkv >> 4) + b * 16);
}
int launch_hyd_latent_q8(hipStream_t st, const HydLatentArgs& A, int P) {
  if (A.nq % 256 || A.nkv % 256 || (A.nq >> 8) + (A.nkv >> 8) > 16) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_latent_q8: latents of %d + %d", A.nq, A.nkv);
  hipLaunchKernelGGL(hyd_latent_q8_kernel, dim3(P), dim3(1024), 0, st, A);
  return DSK_OK;
}

// ------------------------------------------------------------------------------------
// MHA: this position's key / value rows of every head (src/infer.cpp:978-1006; attn_device.h rope_kv_from_lds without the
// query).  All P rows are in the cache before any token's attention reads them.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hyd_kv_write_kernel(const AttnMhaArgs a, const StepParams* __restrict__ sps, const float* __restrict__ kv_b,
                                                           int kvb_stride, const float* __restrict__ kv_a, int kva_stride) {
  const int h = blockIdx.x, p = blockIdx.y, tid = threadIdx.x;
  const StepParams* sp = sps + p;
  const int hd = a.head_dim, nope = a.nope, rope = a.rope, vd = a.v_dim;
  const int kv_pos = sp->kv_pos;
  const float* kvb = kv_b + (size_t)p * kvb_stride + (size_t)h * (nope + vd);
  uint16_t* kc = a.key_cache + ((size_t)kv_pos * a.n_heads + h) * hd;
  uint16_t* vc = a.value_cache + ((size_t)kv_pos * a.n_heads + h) * vd;
  for (int i = tid; i < nope; i += 256) kc[i] = ad::f2h(kvb[i]);
  for (int i = tid; i < vd; i += 256) vc[i] = ad::f2h(kvb[nope + i]);
  if (tid < rope / 2) {
    const float* kr = kv_a + (size_t)p * kva_stride + a.lora;
    const float v0 = kr[2 * tid], v1 = kr[2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    float re, im;
    ad::rope_rot(v0, v1, c, s, re, im);
    if (a.is_v3) {
      kc[nope + 2 * tid] = ad::f2h(re);
      kc[nope + 2 * tid + 1] = ad::f2h(im);
    } else {
      kc[nope + tid] = ad::f2h(re);
      kc[nope + tid + rope / 2] = ad::f2h(im);
    }
  }
}
// MHA: rope of the query + attention over positions [0, kv_len) of the cache, per (head, token): head_attn_kernel's second half
// (same attn_mha_body<1024>: same score / softmax / value-mix trees)
// (64 VGPRs - 12 dwords of scratch - so that TWO workgroups share a CU: the launch is a chain of dependent round trips per (head,
// token), 8192 of them at P = 64: 115 -> 87 us per block.  The MLA kernel below measured slower with the same cap: 30 dwords spilled.)
// From split_min cached positions on the decode step runs n_split workgroups per head over pieces of the context and merges their
// un-normalised partials (head_attn_kernel, kernels_gemv.hip): another float association than one softmax over the whole context.
// Here ONE workgroup walks the same pieces [kv_len s / S, kv_len (s + 1) / S) in turn, keeps the partials in LDS and merges them with
// the decode launch's statements: the same bits.
// (SPLIT is an instantiation of its own, launched only for chunks that reach the regime: the partials' 16 KB of LDS and the merge
// cost the short-context instantiation a third of its speed - 87 -> 129 us per block at P = 64 - when they were one kernel)
template <bool SPLIT>
__global__ __launch_bounds__(1024, 8) void hyd_attn_kernel(const AttnMhaArgs a, const StepParams* __restrict__ sps, const float* __restrict__ q, int q_stride,
                                                        float* __restrict__ out, int out_stride, int n_split, int split_min, int n_tok) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ __attribute__((aligned(16))) float q_s[256];
  __shared__ __attribute__((aligned(16))) float part[4096];
  // Workgroup -> (head, token).  The hardware deals workgroups to the 8 XCDs round-robin by linear index, and each XCD has its own
  // L2: with a (head, token) grid a head's K / V rows (up to 640 B x 512 positions) were read by all 8 XCDs for every token - 10.7 GB
  // per block at P = 512, the launch bound by that (EXPERIMENTS.md 6.4: 1.36 ms of a block's 5).  1-D grid (gridDim.y == 1, n_heads a
  // multiple of 8): index i lives on XCD i % 8, which walks heads i % 8 + 8 k, for each head its tokens in order - a head's rows
  // are read from HBM once per chunk and served from ONE L2 for all its tokens.
  int h, p;
  if (gridDim.y == 1) {
    const int i = (int)blockIdx.x, xcd = i & 7, j = i >> 3;
    const int hh = j / n_tok;
    h = hh * 8 + xcd; p = j - hh * n_tok;
  } else {
    h = blockIdx.x; p = blockIdx.y;
  }
  const int tid = threadIdx.x;
  const StepParams* sp = sps + p;
  float* att = reinterpret_cast<float*>(smem);
  const int hd = a.head_dim, nope = a.nope, rope = a.rope, vd = a.v_dim;
  for (int i = tid; i < hd; i += 1024) q_s[i] = q[(size_t)p * q_stride + (size_t)h * hd + i];
  __syncthreads();
  float qre = 0.f, qim = 0.f;
  if (tid < rope / 2) {  // rope_kv_from_lds, the query part
    const float v0 = q_s[nope + 2 * tid], v1 = q_s[nope + 2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    ad::rope_rot(v0, v1, c, s, qre, qim);
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
  __syncthreads();
  const int kv_len = sp->kv_len;
  if (!SPLIT || n_split <= 1 || kv_len < split_min) {
    const float o = ad::attn_mha_body<1024>(a, q_s, 0, kv_len, h, tid, att, scratch, part);
    if (tid < vd) out[(size_t)p * out_stride + (size_t)h * vd + tid] = o;
    return;
  }
  if constexpr (SPLIT) {
  __shared__ float ml_s[2];
  __shared__ float sp_s[MHA_SPLIT_MAX][260];  // a piece's un-normalised mix [v_dim] | its maximum | its sum
  const int S = n_split;
  for (int j = 0; j < S; ++j) {
    const int t_lo = (int)((long long)kv_len * j / S), t_hi = (int)((long long)kv_len * (j + 1) / S);
    const float oj = ad::attn_mha_body<1024>(a, q_s, t_lo, t_hi, h, tid, att, scratch, part, ml_s);
    if (tid < vd) sp_s[j][tid] = oj;
    __syncthreads();  // ml_s
    if (tid < 2) sp_s[j][vd + tid] = ml_s[tid];
    __syncthreads();
  }
  // O = sum_s O_s e^{m_s - M} / sum_s l_s e^{m_s - M}, pieces in order (head_attn_kernel's merge)
  float M = -INFINITY;
  for (int j = 0; j < S; ++j) M = fmaxf(M, sp_s[j][vd]);
  float Lsum = 0.f, o = 0.f;
  for (int j = 0; j < S; ++j) {
    const float w = expf(sp_s[j][vd] - M);
    Lsum = fmaf(sp_s[j][vd + 1], w, Lsum);
    if (tid < vd) o = fmaf(sp_s[j][tid], w, o);
  }
  o /= Lsum;
  if (tid < vd) out[(size_t)p * out_stride + (size_t)h * vd + tid] = o;
  }
}
int launch_hyd_kv_write(hipStream_t st, const AttnMhaArgs& a, const StepParams* sps, int P, const float* kv_b, int kvb_stride, const float* kv_a, int kva_stride) {
  hipLaunchKernelGGL(hyd_kv_write_kernel, dim3(a.n_heads, P), dim3(256), 0, st, a, sps, kv_b, kvb_stride, kv_a, kva_stride);
  return DSK_OK;
}
int launch_hyd_attn(hipStream_t st, const AttnMhaArgs& a, const StepParams* sps, int P, int max_kv, const float* q, int q_stride, float* out, int out_stride,
                    int n_split, int split_min) {
  const size_t lds = (size_t)max_kv * 4;
  if (lds > 96 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_attn: kv_len %d does not fit LDS", max_kv);
  if (a.head_dim > 256 || a.v_dim > 256) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_attn: head_dim %d / v_head_dim %d", a.head_dim, a.v_dim);
  if (n_split > MHA_SPLIT_MAX) n_split = MHA_SPLIT_MAX;  // (launch_head_attn's clamp)
  const bool split = n_split > 1 && max_kv >= split_min;
  auto k = split ? hyd_attn_kernel<true> : hyd_attn_kernel<false>;
  if (lds > 32 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const bool by_xcd = a.n_heads % 8 == 0;  // (see the kernel: one head's tokens on one XCD)
  hipLaunchKernelGGL(k, by_xcd ? dim3(a.n_heads * P) : dim3(a.n_heads, P), dim3(1024), lds, st, a, sps, q, q_stride, out, out_stride, n_split, split_min, P);
  return DSK_OK;
}

// ------------------------------------------------------------------------------------
// MLA (src/infer.cpp:1051-1141): the latent's cache entries of the P positions (one workgroup per token running the decode
// launch's own mla_kv_write_body), then per (head, token) the attention of mla_head_kernel's short-context path - rope of q_rope,
// scores over the shared latent cache, softmax, the latent value mix - leaving the head's latent output; the per-head wv_b rows
// are a block-diagonal GEMM over all tokens afterwards (launch_hyd_gemm with one task per head).  The code below is the decode
// kernel's (kernels_gemv.hip mla_head_kernel, `!merged` branch), statement for statement: the same trees, the same bits.
// Contexts from MLA_FLASH_MIN_KV positions on take the matrix-core path in decode (its own association): for those tokens the host
// launches decode's own mla_flash_kernel with the token as a third grid dimension (kernels_misc.hip) and hyd_mla_merge_kernel
// below - the merge of mla_head_kernel's `merged` branch, one body for both (attn_device.h mla_merge_partials).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hyd_mla_kv_write_kernel(MlaKvArgs kv, const StepParams* __restrict__ sps, int kva_stride) {
  const int p = blockIdx.x;
  kv.kv_a += (size_t)p * kva_stride;
  rd::mla_kv_write_body(kv, sps + p, threadIdx.x, 1024);
}
__global__ __launch_bounds__(1024) void hyd_mla_attn_kernel(const AttnMlaArgs a, const StepParams* __restrict__ sps, const float* __restrict__ q_c,
                                                            int qc_stride, const float* __restrict__ q_rope, int qr_stride, float* __restrict__ latent,
                                                            int lat_stride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ __attribute__((aligned(16))) float q_s[768];    // q_c | rotated q_rope
  __shared__ __attribute__((aligned(16))) float part[4096];
  constexpr int NT = 1024, NW = 16;
  typedef ad::f16x4 h4;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = lane >> 4, sl = lane & 15;
  const int h = blockIdx.x, p = blockIdx.y;
  const StepParams* sp = sps + p;
  const int lora = a.lora, rope = a.rope, kv_len = sp->kv_len;
  float* att = reinterpret_cast<float*>(smem);
  const int nj = lora >> 6;
  h4 kc[2][8], kr[2];
  auto request_rows = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = t0 + u * 4 + grp;
      if (t < kv_len) {
        const uint16_t* c = a.nope_cache + (size_t)t * lora;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nj) kc[u][j] = *reinterpret_cast<const h4*>(c + 64 * j + sl * 4);
        if (sl * 4 < rope) kr[u] = *reinterpret_cast<const h4*>(a.rope_cache + (size_t)t * rope + sl * 4);
      }
    }
  };
  request_rows(wave * 8);
  // ---- q: latent part as is, rope part rotated (src/infer.cpp:1075-1084) ----
  for (int i = tid; i < lora; i += NT) q_s[i] = q_c[(size_t)p * qc_stride + (size_t)h * lora + i];
  if (tid < rope / 2) {
    const float* qr = q_rope + (size_t)p * qr_stride + (size_t)h * rope;
    const float v0 = qr[2 * tid], v1 = qr[2 * tid + 1];
    const float c = sp->rope_cs[2 * tid], s = sp->rope_cs[2 * tid + 1];
    float re, im;
    ad::rope_rot(v0, v1, c, s, re, im);
    if (a.is_v3) {
      q_s[lora + 2 * tid] = re;
      q_s[lora + 2 * tid + 1] = im;
    } else {
      q_s[lora + tid] = re;
      q_s[lora + tid + rope / 2] = im;
    }
  }
  __syncthreads();
  // ---- scores: 16 lanes per position, 2 positions per group and step ----
  float qv[8][4], qr4[4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) qv[j][i] = j < nj ? q_s[64 * j + sl * 4 + i] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) qr4[i] = sl * 4 + i < rope ? q_s[lora + sl * 4 + i] : 0.f;
  const float inv = sqrtf((float)a.head_dim);
  for (int t0 = wave * 8; t0 < kv_len; t0 += NW * 8) {
    if (t0 != wave * 8) request_rows(t0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = t0 + u * 4 + grp;
      float pp = 0.f;
      if (t < kv_len) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nj) {
            pp = fmaf(qv[j][0], (float)kc[u][j].x, pp);
            pp = fmaf(qv[j][1], (float)kc[u][j].y, pp);
            pp = fmaf(qv[j][2], (float)kc[u][j].z, pp);
            pp = fmaf(qv[j][3], (float)kc[u][j].w, pp);
          }
        if (sl * 4 < rope) {
          pp = fmaf(qr4[0], (float)kr[u].x, pp);
          pp = fmaf(qr4[1], (float)kr[u].y, pp);
          pp = fmaf(qr4[2], (float)kr[u].z, pp);
          pp = fmaf(qr4[3], (float)kr[u].w, pp);
        }
      }
      pp = ad::row16_sum(pp);
      if (sl == 0 && t < kv_len) att[t] = pp / inv;
    }
  }
  __syncthreads();
  // ---- softmax (src/infer.cpp:472-487) ----
  if (kv_len <= 64) {
    // short contexts: every position lives in wave 0's lanes, so the block-wide max / sum of the general path below ARE wave 0's
    // DPP trees (the other waves contribute -inf and exact zeros): same bits, one barrier instead of five
    if (wave == 0) {
      const float a0 = lane < kv_len ? att[lane] : -INFINITY;
      const float mx0 = ad::wave_max_dpp(a0);
      const float e0 = lane < kv_len ? expf(a0 - mx0) : 0.f;
      const float sum0 = ad::wave_sum_dpp(e0);
      if (lane < kv_len) att[lane] = e0 / sum0;
    }
    __syncthreads();
  } else {
  float mx = -INFINITY;
  for (int t = tid; t < kv_len; t += NT) mx = fmaxf(mx, att[t]);
  mx = ad::block_max(mx, scratch, tid, NT);
  float sum = 0.f;
  for (int t = tid; t < kv_len; t += NT) {
    const float e = expf(att[t] - mx);
    att[t] = e;
    sum += e;
  }
  sum = ad::block_sum(sum, scratch, tid, NT);
  for (int t = tid; t < kv_len; t += NT) att[t] = att[t] / sum;
  __syncthreads();
  }
  // ---- latent values: lora / 4 threads per position, NT / (lora / 4) positions in flight, 4 rows per thread ----
  {
    const int tpp = lora >> 2, TG = NT / tpp;
    const int g = tid / tpp, i4 = tid - g * tpp;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < TG) {
      for (int t0 = g; t0 < kv_len; t0 += 4 * TG) {
        h4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t0 + k * TG < kv_len) v[k] = *reinterpret_cast<const h4*>(a.nope_cache + (size_t)(t0 + k * TG) * lora + i4 * 4);
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
      *reinterpret_cast<f32x4*>(part + (size_t)g * lora + i4 * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
    __syncthreads();
    if (tid < lora) {
      float o = 0.f;
      for (int gg = 0; gg < TG; ++gg) o += part[gg * lora + tid];
      latent[(size_t)p * lat_stride + (size_t)h * lora + tid] = o;
    }
  }
}
// the long-context tokens of a chunk: per (head, token) the merge of mla_flash_kernel's chunk partials (mla_head_kernel, `merged`)
__global__ __launch_bounds__(1024) void hyd_mla_merge_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_o, int H, int lora, int n_chunks,
                                                             const StepParams* __restrict__ sps, float* __restrict__ latent, int lat_stride) {
  __shared__ __attribute__((aligned(16))) float part[64 + 1024];
  __shared__ __attribute__((aligned(16))) float o_s[512];
  const int tid = threadIdx.x, h = blockIdx.x, p = blockIdx.y;
  const int kv_len = sps[p].kv_len;
  const int fcl = MLA_FL_CHUNK(kv_len, n_chunks);
  const int nc = min(n_chunks, (kv_len + fcl - 1) / fcl);
  ad::mla_merge_partials(part_ml + (size_t)p * n_chunks * H * 2, part_o + (size_t)p * n_chunks * H * lora, H, h, lora, nc, tid, part, o_s);
  if (tid < lora && tid < 512) latent[(size_t)p * lat_stride + (size_t)h * lora + tid] = o_s[tid];
}
int launch_hyd_mla_merge(hipStream_t st, const MlaFlashArgs& f, const StepParams* sps, int n_tokens, float* latent, int lat_stride) {
  if (f.lora > 512 || f.n_chunks > 64 || f.chunk_len != 0) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_mla_merge: kv_lora_rank %d / %d chunks", f.lora, f.n_chunks);
  hipLaunchKernelGGL(hyd_mla_merge_kernel, dim3(f.n_heads, n_tokens), dim3(1024), 0, st, f.part_ml, f.part_o, f.n_heads, f.lora, f.n_chunks, sps, latent, lat_stride);
  return DSK_OK;
}
// list[h][p] = p * H + h (the rows of head h in the (token, head)-major latent array), count[h] = P: wv_b as one task per head
__global__ void hyd_head_list_kernel(int* __restrict__ list, int* __restrict__ count, int H, int P, int stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < H * P) {
    const int h = i / P, p = i - h * P;
    list[(size_t)h * stride + p] = p * H + h;
  }
  if (i < H) count[i] = P;
}
int launch_hyd_mla_kv_write(hipStream_t st, const MlaKvArgs& kv, const StepParams* sps, int P, int kva_stride) {
  if (kv.rope > 128 || (kv.rope & 1)) DSK_FAIL(DSK_ERR_UNSUPPORTED, "rope dim %d (max 128, even)", kv.rope);
  hipLaunchKernelGGL(hyd_mla_kv_write_kernel, dim3(P), dim3(1024), 0, st, kv, sps, kva_stride);
  return DSK_OK;
}
int launch_hyd_mla_attn(hipStream_t st, const AttnMlaArgs& a, const StepParams* sps, int P, int max_kv, const float* q_c, int qc_stride, const float* q_rope,
                        int qr_stride, float* latent, int lat_stride) {
  if (a.lora > 512 || a.lora % 64 || a.rope > 64 || a.rope % 4) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_mla_attn: kv_lora_rank %d / rope %d", a.lora, a.rope);
  const size_t lds = (size_t)max_kv * 4;
  if (lds > 96 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "hyd_mla_attn: kv_len %d does not fit LDS", max_kv);
  auto k = hyd_mla_attn_kernel;
  if (lds > 32 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(a.n_heads, P), dim3(1024), lds, st, a, sps, q_c, qc_stride, q_rope, qr_stride, latent, lat_stride);
  return DSK_OK;
}
int launch_hyd_head_list(hipStream_t st, int* list, int* count, int H, int P, int stride) {
  hipLaunchKernelGGL(hyd_head_list_kernel, dim3((H * P + 255) / 256), dim3(256), 0, st, list, count, H, P, stride);
  return DSK_OK;
}

// ------------------------------------------------------------------------------------
// router + gate of P tokens (src/infer.cpp:839-851, 493-599).  The decode launch (router_device.h router_body) recomputes the
// norm and re-reads the 7.3 MB of router weights per token; here
//   1. hyd_router_norm_kernel  per token: the router's own rmsnorm scale (router_norm_scale: the 1024-thread tree), the normed
//      vector y = x * scale * w as f32, and its Q8_K copy - what the experts AND the shared expert's rider consume in decode;
//   2. hyd_router_rows_kernel  per workgroup RW rows x 16 / RW column slices like router_body, the weights of a (row, slice)
//      held in REGISTERS across all P tokens: per token the same fma chain over the same values, the same wave tree, the
//      same slice-order sum;
//   3. hyd_gate_kernel         per token: gate_body on the scores.
// Falls back to router_body per token (hyd_router_kernel) when a slice does not fit the registers.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hyd_router_norm_kernel(RouterArgs a0, float* __restrict__ Y) {
  __shared__ float scratch[16];
  RouterArgs a = a0;
  const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, dim = a.dim;
  a.x += (size_t)p * dim;
  const float scale = rd::router_norm_scale(a, tid, scratch);
  float* y = Y + (size_t)p * dim;
  for (int b = wave; b < (dim >> 8); b += 16) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + b * 256 + lane * 4);
    const f32x4 nw = *reinterpret_cast<const f32x4*>(a.norm_w + b * 256 + lane * 4);
    const float v[4] = {xv.x * scale * nw.x, xv.y * scale * nw.y, xv.z * scale * nw.z, xv.w * scale * nw.w};
    *reinterpret_cast<f32x4*>(y + b * 256 + lane * 4) = f32x4{v[0], v[1], v[2], v[3]};
    if (a.q_qs) ad::q8k_block(v, lane, a.q_qs + (size_t)p * dim + b * 256, a.q_d + (size_t)p * (dim >> 8) + b, a.q_bsums + (size_t)p * (dim >> 4) + b * 16);
  }
}
template <int RW>
__global__ __launch_bounds__(1024) void hyd_router_rows_kernel(const float* __restrict__ W, const float* __restrict__ Y, int dim, int E, int P,
                                                               float* __restrict__ partial) {
  constexpr int SL = 16 / RW;
  extern __shared__ float part_s[];  // [P][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bid = blockIdx.x;
  const int r = wave % RW, sl = wave / RW;
  const int row = bid * RW + r;
  const int chunk = ((dim / 4 + SL - 1) / SL + 63) / 64 * 64 * 4;  // floats per column slice (router_body)
  const int k0 = sl * chunk, k1 = min(dim, k0 + chunk);
  const int i0 = k0 + lane * 4;
  f32x4 wv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    wv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < E && i0 + k * 256 < k1) wv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)row * dim + i0 + k * 256));
  }
  for (int t = 0; t < P; ++t) {
    const float* y = Y + (size_t)t * dim;
    float acc = 0.f;
    if (row < E) {
      f32x4 yv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k * 256 < k1) yv[k] = *reinterpret_cast<const f32x4*>(y + i0 + k * 256);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k * 256 < k1) {
          acc = fmaf(wv[k].x, yv[k].x, acc);
          acc = fmaf(wv[k].y, yv[k].y, acc);
          acc = fmaf(wv[k].z, yv[k].z, acc);
          acc = fmaf(wv[k].w, yv[k].w, acc);
        }
      acc = ad::wave_sum_dpp(acc);
    }
    if (lane == 0) part_s[t * 16 + wave] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < P * RW; idx += 1024) {
    const int t = idx / RW, rr = idx - t * RW;
    if (bid * RW + rr < E) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < SL; ++k) v += part_s[t * 16 + k * RW + rr];  // slice order, like router_body
      partial[(size_t)t * E + bid * RW + rr] = v;
    }
  }
}
__global__ __launch_bounds__(1024) void hyd_gate_kernel(const RouterArgs a, int K) {
  __shared__ __attribute__((aligned(16))) float s[512];
  __shared__ __attribute__((aligned(16))) int surv[256];
  __shared__ float scratch[16];
  __shared__ int sel[256];
  const int p = blockIdx.x, tid = threadIdx.x, E = a.n_routed;
  float v = 0.f;
  if (tid < E) v = a.partial[(size_t)p * E + tid];
  rd::gate_body(tid, v, a.bias, E, a.n_active, a.norm_topk_prob, a.scaling, a.scoring, a.topk_method, a.n_group, a.topk_group,
                a.active_experts + (size_t)p * K, a.active_weights + (size_t)p * K, nullptr, s, surv, sel, scratch, 1024);
}
template <int RW>
__global__ __launch_bounds__(1024) void hyd_router_kernel(const RouterArgs a0, int K) {
  RouterArgs a = a0;
  const int p = blockIdx.y;
  const int dim = a.dim, E = a.n_routed;
  a.x += (size_t)p * dim;
  a.partial += (size_t)p * E;
  a.counter += p;
  a.active_experts += (size_t)p * K;
  a.active_weights += (size_t)p * K;
  if (a.scores_out) a.scores_out += (size_t)p * E;
  if (a.q_qs) { a.q_qs += (size_t)p * dim; a.q_d += (size_t)p * (dim >> 8); a.q_bsums += (size_t)p * (dim >> 4); }
  rd::router_body<RW>(a, (int)blockIdx.x, (int)gridDim.x);
}
// Y: P x dim floats of scratch (the normed vectors); a.partial: P x E
int launch_hyd_router(hipStream_t st, const RouterArgs& a, int P, float* Y) {
  const int RW = a.ksplit >= 8 ? 2 : 4, SL = 16 / RW;
  const int chunk = ((a.dim / 4 + SL - 1) / SL + 63) / 64 * 64 * 4;
  const size_t lds = (size_t)P * 16 * 4;
  if (Y && a.norm_w && a.dim % 256 == 0 && chunk <= 2048 && lds <= 60 * 1024 && a.n_routed <= 1024) {
    hipLaunchKernelGGL(hyd_router_norm_kernel, dim3(P), dim3(1024), 0, st, a, Y);
    if (RW == 2) hipLaunchKernelGGL(hyd_router_rows_kernel<2>, dim3((a.n_routed + 1) / 2), dim3(1024), lds, st, a.w, Y, a.dim, a.n_routed, P, a.partial);
    else hipLaunchKernelGGL(hyd_router_rows_kernel<4>, dim3((a.n_routed + 3) / 4), dim3(1024), lds, st, a.w, Y, a.dim, a.n_routed, P, a.partial);
    hipLaunchKernelGGL(hyd_gate_kernel, dim3(P), dim3(1024), 0, st, a, a.n_active);
    return DSK_OK;
  }
  if (RW == 2) hipLaunchKernelGGL(hyd_router_kernel<2>, dim3((a.n_routed + 1) / 2, P), dim3(1024), 0, st, a, a.n_active);
  else hipLaunchKernelGGL(hyd_router_kernel<4>, dim3((a.n_routed + 3) / 4, P), dim3(1024), 0, st, a, a.n_active);
  return DSK_OK;
}

#ifdef DSK_AB
// MEASUREMENT ONLY, -DDSK_AB builds (tools/ab_build.sh, env DSK_HYD_ROUTE_SEED): replace the gate's choice by K distinct experts per token drawn uniformly by a
// hash of (seed, layer, token, slot).  The synthetic benchmark model routes most tokens of a chunk to the same ~125 experts (its
// random router sees strongly correlated inputs), which flatters a batched prompt: a trained model balances its experts, and the
// bytes a chunk touches are what the prompt phase is made of.  Never used by the parity tests.
__global__ void hyd_route_override_kernel(int* __restrict__ route_e, int P, int K, int E, unsigned seed) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  for (int k = 0; k < K; ++k) {
    unsigned h = seed * 2654435761u ^ (unsigned)(p * 7919 + k * 104729 + 1);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    int e = (int)(h % (unsigned)E);
    for (;;) {  // distinct within the token: linear probing
      bool dup = false;
      for (int j = 0; j < k; ++j) dup = dup || route_e[p * K + j] == e;
      if (!dup) break;
      e = (e + 1) % E;
    }
    route_e[p * K + k] = e;
  }
}
int launch_hyd_route_override(hipStream_t st, int* route_e, int P, int K, int E, unsigned seed) {
  hipLaunchKernelGGL(hyd_route_override_kernel, dim3((P + 63) / 64), dim3(64), 0, st, route_e, P, K, E, seed);
  return DSK_OK;
}
#endif

// tokens grouped by expert: list[e] = the (token, slot) pairs p * K + k routed to expert e, in pair order; count[e].
// One workgroup per expert scans the pairs 256 at a time (ballot + prefix counts: the order is the pair order, deterministic).
__global__ __launch_bounds__(256) void hyd_group_kernel(const int* __restrict__ route_e, int pairs, int* __restrict__ list, int list_stride,
                                                        int* __restrict__ count) {
  __shared__ int wcnt[4];
  const int e = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int base = 0;
  for (int i0 = 0; i0 < pairs; i0 += 256) {
    const int i = i0 + tid;
    const bool hit = i < pairs && route_e[i] == e;
    const unsigned long long m = __ballot(hit);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) wbase += wcnt[w];
      total += wcnt[w];
    }
    if (hit) list[(size_t)e * list_stride + base + wbase + before] = i;
    base += total;
    __syncthreads();
  }
  if (tid == 0) count[e] = base;
}
int launch_hyd_group(hipStream_t st, const int* route_e, int pairs, int E, int* list, int list_stride, int* count) {
  hipLaunchKernelGGL(hyd_group_kernel, dim3(E), dim3(256), 0, st, route_e, pairs, list, list_stride, count);
  return DSK_OK;
}

// x[p] += sum_k w[p][k] * eout[p * K + k] in k order, then the shared expert's row (src/infer.cpp:874-877, 900-903; the
// multiply-adds of the decode launches' fused combine)
__global__ __launch_bounds__(256) void hyd_combine_kernel(float* __restrict__ X, const float* __restrict__ eout, const float* __restrict__ w,
                                                          const float* __restrict__ eout_sh, int K, int n) {
  const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float xv = X[(size_t)p * n + i];
  for (int k = 0; k < K; ++k) xv = fmaf(eout[((size_t)p * K + k) * n + i], w[(size_t)p * K + k], xv);
  if (eout_sh) xv += eout_sh[(size_t)p * n + i];
  X[(size_t)p * n + i] = xv;
}
int launch_hyd_combine(hipStream_t st, float* X, const float* eout, const float* w, const float* eout_sh, int P, int K, int n) {
  hipLaunchKernelGGL(hyd_combine_kernel, dim3((n + 255) / 256, P), dim3(256), 0, st, X, eout, w, eout_sh, K, n);
  return DSK_OK;
}
