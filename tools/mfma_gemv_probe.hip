// mfma_gemv_probe.hip -- go / no-go probe: the Q2_K x Q8_K GEMV (src/quant.cpp:666-783) with the 16 sub-block dot
// products of a 256-block taken off the VALU and put on the matrix pipe (v_mfma_i32_16x16x64_i8), as a stand-alone
// micro-kernel on the DeepSeek-V3 decode shapes.  No engine integration: its own weight layout, its own check.
//
// The segmented dot.  One MFMA computes D[16 x 16] += A[16 x 64] * B[64 x 16] in int8 -> int32.  Here
//   B (64 x 16) = the WEIGHTS: column n = weight row n of a 16-row tile; K-group g (16 bytes) = the row's qs bytes
//                 [16g, 16g+16) with ONE 2-bit field (shift s) masked in place: elements 128h + 32s + 16lh + t of the
//                 block (g = 2h + lh), i.e. sub-block j(g, s) = 8h + 2s + lh  (dequantize_row_q2_K, src/quant.cpp:217-247)
//   A (16 x 64) = the ACTIVATIONS as sub-block selectors: row i of K-group g holds the 16 int8 codes of sub-block
//                 j(g, s) if i == j(g, s), zeros otherwise.
// Four MFMAs (s = 0..3) accumulate D[j][n] = sum_t q8[16j + t] * q2_n[16j + t] * f(s(j)) for all 16 sub-blocks of 16
// rows: the exact int32 sub-sums of quant.cpp:746-780 (f = 1, 4, 16, 16: the in-place masks' factors).  15/16 of the MACs
// multiply zeros; the matrix pipe has the headroom.  In the C/D layout lane (n = lane & 15, g4 = lane >> 4) holds the
// sums of sub-blocks 4 g4 .. 4 g4 + 3 of ITS row n: the scale products are 4 mad24 of one lane, the int -> f32 and the
// FMA with d happen once per lane and block, the min term is two dot4 against the split bsums as before.  Per 1 KiB of
// qs (16 rows x one block): 20 VALU for the unpack + ~22 for scales / min / float, against ~70 for 64 dot4 items.
//
// Weight layout ("tiles-v1"): a 16-row x 256-column tile is one contiguous 1344-byte record
//   [   0, 1024)  qs: lane l = n + 16 g reads 16 bytes at 16 l: row n's qs bytes [16g, 16g+16)
//   [1024, 1280)  scales: lane l = n + 16 g4 reads 4 bytes at 4 l: row n's scales[4 g4 .. 4 g4 + 3] (unpermuted)
//   [1280, 1344)  d | dmin << 16 of row n at 4 n
// tiles of a row strip (16 rows, all blocks) are contiguous: every load instruction of a wave covers whole lines.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_probe tools/mfma_gemv_probe.hip
// Run:   /tmp/mfma_probe            (all shapes, check + timing)     /tmp/mfma_probe check    (small shapes, exact check only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x)                                                                                                      \
  do {                                                                                                             \
    hipError_t e_ = (x);                                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
#define TILE_B 1344
#define REC_B 320  // LDS record of one staged 256-block: codes[256] | zeros[16] | bsum hi/lo per g4 [32] | d | pad

typedef __amdgpu_buffer_rsrc_t rsrc_t;
DEV rsrc_t make_rsrc(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
}
DEV float h2f(u32 bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); }
DEV float u2f(u32 v) { return __builtin_bit_cast(float, v); }

__host__ __device__ inline u32 hash32(u32 x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// word `idx` (4-byte units) of a weight buffer made of tiles: random bytes, except the d | dmin words: f16 of U(0.5, 1.5) * 1e-2
__host__ __device__ inline u32 weight_word(size_t idx, u32 seed) {
  const u32 h = hash32((u32)idx * 2654435761u ^ seed ^ (u32)(idx >> 32));
  const u32 in_tile = (u32)(idx % (TILE_B / 4));
  if (in_tile < 1280 / 4) return h;
  const _Float16 d = (_Float16)((0.5f + (float)(h & 0xff) / 256.0f) * 0.01f);
  const _Float16 m = (_Float16)((0.5f + (float)((h >> 8) & 0xff) / 256.0f) * 0.01f);
  unsigned short db, mb;
  memcpy(&db, &d, 2); memcpy(&mb, &m, 2);
  return (u32)db | ((u32)mb << 16);
}
__global__ void fill_weights(u32* w, size_t nwords, u32 seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) w[i] = weight_word(i, seed);
}

// ---------------------------------------------------------------------------------------------------------------
// one column step: a 16-row tile x one 256-block.  rec = the block's LDS record; aoff[s] = this lane's selector offset.
// accd accumulates dd * (isum * 4 or * 16, see lanefac), accm accumulates dmin * summs.
// ---------------------------------------------------------------------------------------------------------------
struct Step { u32x4 w; u32 scw, dm; };

DEV void load_step(Step& S, rsrc_t W, int vq, int vs, int vd, int soff) {
  S.w = __builtin_amdgcn_raw_buffer_load_b128(W, vq, soff, 2);
  S.scw = __builtin_amdgcn_raw_buffer_load_b32(W, vs, soff, 2);
  S.dm = __builtin_amdgcn_raw_buffer_load_b32(W, vd, soff, 2);
}

template <bool DBG>
DEV void compute_step(const Step& S, const uint8_t* rec, const int (&aoff)[4], int g4, int shA, float& accd, float& accm, int* dbg) {
  const u32x4 w = S.w;
  i32x4 b0, b1, b2, b3;
  b0.x = w.x & 0x03030303u; b0.y = w.y & 0x03030303u; b0.z = w.z & 0x03030303u; b0.w = w.w & 0x03030303u;
  b1.x = w.x & 0x0C0C0C0Cu; b1.y = w.y & 0x0C0C0C0Cu; b1.z = w.z & 0x0C0C0C0Cu; b1.w = w.w & 0x0C0C0C0Cu;
  b2.x = w.x & 0x30303030u; b2.y = w.y & 0x30303030u; b2.z = w.z & 0x30303030u; b2.w = w.w & 0x30303030u;
  b3.x = (w.x >> 2) & 0x30303030u; b3.y = (w.y >> 2) & 0x30303030u; b3.z = (w.z >> 2) & 0x30303030u; b3.w = (w.w >> 2) & 0x30303030u;
  const i32x4 a0 = *reinterpret_cast<const i32x4*>(rec + aoff[0]);
  const i32x4 a1 = *reinterpret_cast<const i32x4*>(rec + aoff[1]);
  const i32x4 a2 = *reinterpret_cast<const i32x4*>(rec + aoff[2]);
  const i32x4 a3 = *reinterpret_cast<const i32x4*>(rec + aoff[3]);
  i32x4 D = {0, 0, 0, 0};
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, D, 0, 0, 0);
  D = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b3, D, 0, 0, 0);
  const u32 scw = S.scw;
  const int d0 = scw & 0xF, d1 = (scw >> 8) & 0xF, d2 = (scw >> 16) & 0xF, d3 = (scw >> 24) & 0xF;
  const int t01 = __mul24(d0, D.x) + __mul24(d1, D.y);
  const int t23 = __mul24(d2, D.z) + __mul24(d3, D.w);
  const int isx = (t01 << shA) + t23;  // even g4: 4 * isum; odd g4: 16 * isum
  const u32 m4 = (scw >> 4) & 0x0F0F0F0Fu;
  const unsigned long long bs = *reinterpret_cast<const unsigned long long*>(rec + 272 + 8 * g4);
  const int summs = (__builtin_amdgcn_sdot4((int)m4, (int)(u32)bs, 0, false) << 8) + (int)__builtin_amdgcn_udot4(m4, (u32)(bs >> 32), 0u, false);
  const float dx = *reinterpret_cast<const float*>(rec + 304);
  const float dd = dx * h2f(S.dm & 0xffff), dmn = dx * h2f(S.dm >> 16);
  accd = fmaf(dd, (float)isx, accd);
  accm = fmaf(dmn, (float)summs, accm);
  if (DBG) { dbg[0] += isx; dbg[1] += summs; }
}

// stage a ready Q8_K vector (codes, block scales, sub-block sums) into LDS records
DEV void stage_act(uint8_t* lds, const int8_t* aq, const float* ad, const short* ab, int nb, int tid, int nthreads) {
  for (int i = tid; i < nb * 16; i += nthreads) {  // 16-byte runs = sub-blocks
    const int b = i >> 4, j = i & 15;
    *reinterpret_cast<u32x4*>(lds + b * REC_B + j * 16) = reinterpret_cast<const u32x4*>(aq)[i];
    const int bsum = ab[i];
    lds[b * REC_B + 272 + 8 * (j >> 2) + (j & 3)] = (uint8_t)(bsum >> 8);
    lds[b * REC_B + 272 + 8 * (j >> 2) + 4 + (j & 3)] = (uint8_t)(bsum & 0xff);
  }
  for (int b = tid; b < nb; b += nthreads) {
    *reinterpret_cast<u32x4*>(lds + b * REC_B + 256) = u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<float*>(lds + b * REC_B + 304) = ad[b];
  }
}

// NW waves per workgroup = TW row tiles x KS splits of the row; each wave: BPS consecutive blocks (all requested at once)
template <int NW, int KS, int BPS, bool DBG>
__global__ __launch_bounds__(NW * 64) void q2k_mfma_gemv(const uint8_t* W, const int8_t* aq, const float* ad, const short* ab, float* out, int rows, int* dbg) {
  constexpr int TW = NW / KS, NB = KS * BPS;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  float* red = reinterpret_cast<float*>(lds + NB * REC_B);  // [2][NW][64]
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  stage_act(lds, aq, ad, ab, NB, tid, NW * 64);
  __syncthreads();
  const int n = lane & 15, g = lane >> 4;
  int aoff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int j = 8 * (g >> 1) + 2 * s + (g & 1);
    aoff[s] = n == j ? 16 * j : 256;
  }
  const int shA = (g & 1) ? 0 : 2;
  const float lanefac = (g & 1) ? 0.0625f : 0.25f;
  const rsrc_t Wr = make_rsrc(W);
  const int vq = lane * 16, vs = 1024 + lane * 4, vd = 1280 + n * 4;
  const int tl = wave / KS, ks = wave - tl * KS;
  const int ntiles = rows >> 4;
  int par = 0;
  for (int t0 = blockIdx.x * TW; t0 < ntiles; t0 += gridDim.x * TW, par ^= 1) {
    const int t = t0 + tl;
    float accd = 0.f, accm = 0.f;
    int dbgv[2] = {0, 0};
    if (t < ntiles) {  // wave-uniform
      Step S[BPS];
      const int soff0 = (t * NB + ks * BPS) * TILE_B;
#pragma unroll
      for (int u = 0; u < BPS; ++u) load_step(S[u], Wr, vq, vs, vd, soff0 + u * TILE_B);
#pragma unroll
      for (int u = 0; u < BPS; ++u) compute_step<DBG>(S[u], lds + (ks * BPS + u) * REC_B, aoff, g, shA, accd, accm, dbgv);
    }
    red[(par * NW + wave) * 64 + lane] = accd * lanefac - accm;
    if (DBG && t < ntiles) {  // per (row, k split, g4): the integer partial sums (isx is 4x / 16x, see lanefac)
      dbg[(((size_t)(t * 16 + n) * KS + ks) * 4 + g) * 2] = dbgv[0];
      dbg[(((size_t)(t * 16 + n) * KS + ks) * 4 + g) * 2 + 1] = dbgv[1];
    }
    __syncthreads();
    if (tid < TW * 16) {
      const int tt = tid >> 4, nn = tid & 15;
      if (t0 + tt < ntiles) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) tot += red[(par * NW + tt * KS + k) * 64 + q * 16 + nn];
        out[(t0 + tt) * 16 + nn] = tot;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host: reference of one row from the same hash (src/quant.cpp:746-780 scalar branch, double accumulation)
// ---------------------------------------------------------------------------------------------------------------
struct RowRef { double val; std::vector<long long> isum, summs; };
static RowRef ref_row(int r, int nb, u32 seed, const std::vector<int8_t>& aq, const std::vector<float>& ad, const std::vector<short>& ab) {
  RowRef R;
  R.val = 0;
  const int t = r >> 4, n = r & 15;
  for (int b = 0; b < nb; ++b) {
    const size_t T = ((size_t)t * nb + b) * TILE_B;
    auto byte_at = [&](size_t off) { return (uint8_t)(weight_word((T + off) >> 2, seed) >> (8 * ((T + off) & 3))); };
    uint8_t qs[64], sc[16];
    for (int i = 0; i < 64; ++i) qs[i] = byte_at(256 * (i / 16) + 16 * n + (i % 16));
    for (int j = 0; j < 16; ++j) sc[j] = byte_at(1024 + 4 * (n + 16 * (j / 4)) + (j % 4));
    const u32 dmw = weight_word((T + 1280 + 4 * n) >> 2, seed);
    unsigned short db = dmw & 0xffff, mb = dmw >> 16;
    _Float16 dh, mh;
    memcpy(&dh, &db, 2); memcpy(&mh, &mb, 2);
    long long isum = 0, summs = 0;
    for (int j = 0; j < 16; ++j) {
      long long x = 0;
      for (int l = 0; l < 16; ++l) {
        const int e = 16 * j + l;
        const int q2 = (qs[32 * (e / 128) + e % 32] >> (2 * ((e % 128) / 32))) & 3;
        x += (long long)aq[b * 256 + e] * q2;
      }
      isum += (long long)(sc[j] & 0xF) * x;
      summs += (long long)(sc[j] >> 4) * ab[b * 16 + j];
    }
    R.isum.push_back(isum);
    R.summs.push_back(summs);
    R.val += (double)ad[b] * (double)(float)dh * (double)isum - (double)ad[b] * (double)(float)mh * (double)summs;
  }
  return R;
}

struct Shape { const char* name; int rows, n; };

template <int NW, int KS, int BPS>
static double run_variant(const Shape& sh, int grid, int iters, int nsets, const uint8_t* dW, size_t set_bytes, const int8_t* daq, const float* dad,
                          const short* dab, float* dout) {
  const size_t lds = (size_t)KS * BPS * REC_B + 2 * NW * 64 * 4;
  auto k = q2k_mfma_gemv<NW, KS, BPS, false>;
  if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, dW + (size_t)(i % nsets) * set_bytes, daq, dad, dab, dout, sh.rows, nullptr);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, dW + (size_t)(i % nsets) * set_bytes, daq, dad, dab, dout, sh.rows, nullptr);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)ms * 1000.0 / iters;
}

template <int NW, int KS, int BPS>
static int check_variant(const Shape& sh, int grid, const uint8_t* dW, u32 seed, const int8_t* daq, const float* dad, const short* dab, float* dout,
                         const std::vector<int8_t>& aq, const std::vector<float>& ad, const std::vector<short>& ab, int nsample) {
  const int nb = sh.n / 256;
  const size_t lds = (size_t)KS * BPS * REC_B + 2 * NW * 64 * 4;
  auto k = q2k_mfma_gemv<NW, KS, BPS, true>;
  if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int* ddbg;
  const size_t dbg_n = (size_t)sh.rows * KS * 4 * 2;
  CK(hipMalloc(&ddbg, dbg_n * 4));
  CK(hipMemset(ddbg, 0, dbg_n * 4));
  CK(hipMemset(dout, 0xff, (size_t)sh.rows * 4));
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, dW, daq, dad, dab, dout, sh.rows, ddbg);
  CK(hipDeviceSynchronize());
  std::vector<float> out(sh.rows);
  std::vector<int> dbg(dbg_n);
  CK(hipMemcpy(out.data(), dout, (size_t)sh.rows * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(dbg.data(), ddbg, dbg_n * 4, hipMemcpyDeviceToHost));
  CK(hipFree(ddbg));
  int bad_int = 0, bad_f = 0;
  double worst = 0;
  for (int i = 0; i < nsample; ++i) {
    const int r = i < 48 ? i : (int)(hash32(i * 977 + 5) % (u32)sh.rows);
    const RowRef R = ref_row(r, nb, seed, aq, ad, ab);
    // integers: sum over the row's blocks of isum / summs against the device's lane partials (x4 / x16 factors undone)
    long long isum = 0, summs = 0, disum = 0, dsumms = 0;
    for (int b = 0; b < nb; ++b) { isum += R.isum[b]; summs += R.summs[b]; }
    for (int ks = 0; ks < KS; ++ks)
      for (int g = 0; g < 4; ++g) {
        const int v = dbg[(((size_t)r * KS + ks) * 4 + g) * 2];
        disum += (g & 1) ? v : 4LL * v;   // in units of 16 * isum
        dsumms += dbg[(((size_t)r * KS + ks) * 4 + g) * 2 + 1];
      }
    if (disum != 16 * isum || dsumms != summs) {
      if (bad_int < 4) fprintf(stderr, "  row %d: integer sums differ: isum16 %lld vs %lld, summs %lld vs %lld\n", r, disum, 16 * isum, dsumms, summs);
      ++bad_int;
    }
    const double err = fabs((double)out[r] - R.val) / (fabs(R.val) + 1e-3);
    if (err > worst) worst = err;
    if (!(err < 2e-5)) {
      if (bad_f < 4) fprintf(stderr, "  row %d: %.9g vs %.9g\n", r, out[r], R.val);
      ++bad_f;
    }
  }
  printf("  check <%d waves, %d splits x %d blocks> %s %dx%d: %d rows sampled, integer mismatches %d, float mismatches %d, worst rel err %.2e\n",
         NW, KS, BPS, sh.name, sh.rows, sh.n, nsample, bad_int, bad_f, worst);
  return bad_int + bad_f;
}

int main(int argc, char** argv) {
  const bool check_only = argc > 1 && !strcmp(argv[1], "check");
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; mfma_gemv_probe: Q2_K GEMV with the sub-block dots on v_mfma_i32_16x16x64_i8 (ready Q8_K input)\n", prop.gcnArchName, prop.multiProcessorCount);
  const Shape shapes[] = {{"experts_w13", 9 * 2 * 2048, 7168}, {"lm_head", 129280, 7168}, {"wq_b", 24576, 1536}, {"wo", 7168, 16384}, {"experts_w2", 9 * 7168, 2048}};
  const u32 seed = 0x1234567u;
  int failures = 0;
  for (const Shape& sh : shapes) {
    const int nb = sh.n / 256;
    const size_t set_bytes = (size_t)(sh.rows / 16) * nb * TILE_B;
    int nsets = (int)(1.0e9 / (double)set_bytes) + 1;
    if (nsets < 2) nsets = 2;
    if (nsets > 64) nsets = 64;
    if (check_only) nsets = 1;
    uint8_t* dW;
    CK(hipMalloc(&dW, set_bytes * nsets));
    // one hash stream over all sets: set 0 is what the host reference reproduces
    hipLaunchKernelGGL(fill_weights, dim3(2048), dim3(256), 0, 0, reinterpret_cast<u32*>(dW), set_bytes * nsets / 4, seed);
    std::vector<int8_t> aq(sh.n);
    std::vector<float> ad(nb);
    std::vector<short> ab(sh.n / 16);
    for (int i = 0; i < sh.n; ++i) aq[i] = (int8_t)((int)(hash32(i * 31 + 7) % 255) - 127);
    for (int b = 0; b < nb; ++b) ad[b] = 0.01f + (float)(hash32(b + 99) % 1000) * 1e-5f;
    for (int j = 0; j < sh.n / 16; ++j) { int s = 0; for (int l = 0; l < 16; ++l) s += aq[16 * j + l]; ab[j] = (short)s; }
    int8_t* daq; float* dad; short* dab; float* dout;
    CK(hipMalloc(&daq, sh.n)); CK(hipMalloc(&dad, nb * 4)); CK(hipMalloc(&dab, sh.n / 8)); CK(hipMalloc(&dout, (size_t)sh.rows * 4));
    CK(hipMemcpy(daq, aq.data(), sh.n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dad, ad.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dab, ab.data(), sh.n / 8, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    const double mb = (double)set_bytes / 1e6;
    printf("== %s %d x %d: %.2f MB per launch, %d rotating weight sets\n", sh.name, sh.rows, sh.n, mb, nsets);
    const int ntiles = sh.rows / 16;
#define RUN(NW, KS, BPS, GRID)                                                                                                        \
  do {                                                                                                                                \
    const int grid_ = (GRID);                                                                                                         \
    failures += check_variant<NW, KS, BPS>(sh, grid_, dW, seed, daq, dad, dab, dout, aq, ad, ab, 96);                                 \
    if (!check_only) {                                                                                                                \
      const double us = run_variant<NW, KS, BPS>(sh, grid_, 40, nsets, dW, set_bytes, daq, dad, dab, dout);                           \
      printf("  <%2d waves, %d splits x %2d blocks> grid %5d: %8.2f us  %7.1f GB/s\n", NW, KS, BPS, grid_, us, mb / us * 1e3);        \
    }                                                                                                                                 \
  } while (0)
    if (nb == 28) {
      RUN(16, 4, 7, (ntiles + 3) / 4 < 256 ? (ntiles + 3) / 4 : 256);
      RUN(16, 4, 7, (ntiles + 3) / 4 < 512 ? (ntiles + 3) / 4 : 512);
      RUN(16, 4, 7, (ntiles + 3) / 4);
      RUN(16, 2, 14, (ntiles + 7) / 8 < 256 ? (ntiles + 7) / 8 : 256);
      RUN(8, 4, 7, (ntiles + 1) / 2 < 512 ? (ntiles + 1) / 2 : 512);
      RUN(8, 4, 7, (ntiles + 1) / 2);
      RUN(4, 4, 7, ntiles < 1024 ? ntiles : 1024);
      RUN(4, 4, 7, ntiles);
    } else if (nb == 6) {
      RUN(16, 1, 6, (ntiles + 15) / 16 < 256 ? (ntiles + 15) / 16 : 256);
      RUN(16, 2, 3, (ntiles + 7) / 8 < 256 ? (ntiles + 7) / 8 : 256);
      RUN(4, 1, 6, (ntiles + 3) / 4);
      RUN(4, 2, 3, (ntiles + 1) / 2);
    } else if (nb == 64) {
      RUN(16, 8, 8, (ntiles + 1) / 2 < 256 ? (ntiles + 1) / 2 : 256);
      RUN(16, 16, 4, ntiles < 256 ? ntiles : 256);
      RUN(16, 16, 4, ntiles);
      RUN(8, 8, 8, ntiles);
    } else if (nb == 8) {
      RUN(16, 1, 8, (ntiles + 15) / 16 < 256 ? (ntiles + 15) / 16 : 256);
      RUN(16, 2, 4, (ntiles + 7) / 8 < 256 ? (ntiles + 7) / 8 : 256);
      RUN(16, 2, 4, (ntiles + 7) / 8);
      RUN(4, 2, 4, (ntiles + 1) / 2);
    }
    CK(hipFree(dW)); CK(hipFree(daq)); CK(hipFree(dad)); CK(hipFree(dab)); CK(hipFree(dout));
  }
  printf(failures ? "# FAILED: %d mismatches\n" : "# all checks passed\n", failures);
  return failures ? 1 : 0;
}
