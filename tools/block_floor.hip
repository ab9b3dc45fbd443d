// block_floor.hip -- the floor of a five-launch MoE block on this chip: five DEPENDENT streaming kernels with the byte profile of
// one DeepSeek-V3 Q2_K decode block (first-stage projections 5.0 MB, per-head attention 18.1 MB, wo 38.6 MB, router + shared
// expert 17.1 MB, routed experts 120.9 MB = 199.7 MB), trivial arithmetic, a 28 KB all-to-all vector between them (every
// workgroup of a kernel reads the whole vector its predecessor wrote, and writes its share of the next), non-temporal 16-byte
// loads with 8 in flight per lane over per-wave contiguous shares (the form dsk_measure_read_bw measures the roofline with),
// hipGraph-captured.  It is what the engine's 78.5 us per block would be if every launch were a pure stream: the guide's
// "launches-baseline" for THIS byte profile (MI355X_MICROARCH.md: 121.6 MB bf16 in 30.6 us).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/block_floor tools/block_floor.hip && tools/_build/block_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CK(x)                                                                                                      \
  do {                                                                                                             \
    hipError_t e_ = (x);                                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define VEC_N 7168  // floats of the vector handed from launch to launch

// NW waves per workgroup; wave w of the launch streams the contiguous share [w, w + 1) * bytes_per_wave of `src`, 1 KiB per step, 8
// steps in flight and re-issued as they are consumed (kernels_misc.hip read_bw_kernel: the form that streams 6.9 TB/s)
template <int NW>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const uint8_t* __restrict__ src, size_t bytes_per_wave, const float* __restrict__ vin, float* __restrict__ vout) {
  constexpr int D = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * NW + wave;
  // the predecessor's vector, whole (28 KB from L2 / memory: the dependency edge of a real block); requested first, used last
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 xv[(VEC_N / 4 + NW * 64 - 1) / (NW * 64)];
#pragma unroll
  for (int k = 0; k < (int)(sizeof(xv) / sizeof(xv[0])); ++k) {
    const int i = threadIdx.x + k * NW * 64;
    xv[k] = i < VEC_N / 4 ? reinterpret_cast<const f32x4*>(vin)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const unsigned long long base = (unsigned long long)(src + wid * bytes_per_wave);
  const u32 blo = __builtin_amdgcn_readfirstlane((u32)base), bhi = __builtin_amdgcn_readfirstlane((u32)(base >> 32));
  const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)bhi << 32) | blo), 0, -1, 0x00020000);
  const int steps = (int)(bytes_per_wave >> 10);
  u32x4 v[D];
  u32 acc = 0;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < steps) v[d] = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, d << 10, 2);
  for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s0 + d < steps) {
        const u32x4 w = v[d];
        acc += w.x + w.y + w.z + w.w;
        if (s0 + D + d < steps) v[d] = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, (s0 + D + d) << 10, 2);
      }
    }
  }
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < (int)(sizeof(xv) / sizeof(xv[0])); ++k) a += xv[k].x + xv[k].y + xv[k].z + xv[k].w;
  // this workgroup's share of the next vector
  const int per = (VEC_N + gridDim.x - 1) / gridDim.x;
  const int o = blockIdx.x * per + threadIdx.x;
  if ((int)threadIdx.x < per && o < VEC_N) vout[o] = a * 1e-9f + (float)(acc & 1);
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 58;
  const double mb[5] = {5.0, 18.1, 38.6, 17.1, 120.9};
  size_t bytes[5], off[5], per_layer = 0;
  for (int k = 0; k < 5; ++k) {
    bytes[k] = (size_t)(mb[k] * 1e6) / (1 << 20) * (1 << 20);
    off[k] = per_layer;
    per_layer += bytes[k] + (8 << 20);  // (a wave's share is rounded up to whole KiB: slack behind every region)
  }
  uint8_t* W;
  CK(hipMalloc(&W, per_layer * layers));
  CK(hipMemset(W, 1, per_layer * layers));
  float *va, *vb;
  CK(hipMalloc(&va, VEC_N * 4));
  CK(hipMalloc(&vb, VEC_N * 4));
  CK(hipMemset(va, 0, VEC_N * 4));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // geometry g: 0 = 256 sixteen-wave workgroups per launch (one per CU: the engine's), 1 = 1024 four-wave workgroups, 2 = 2048 four-wave
  auto enqueue = [&](int g, int only) {
    float *in = va, *out = vb;
    for (int l = 0; l < layers; ++l)
      for (int k = 0; k < 5; ++k) {
        if (only >= 0 && k != only) continue;
        const uint8_t* src = W + (size_t)l * per_layer + off[k];
        if (g == 0) {
          const size_t pw = (bytes[k] / (256 * 16) + 1023) / 1024 * 1024;
          hipLaunchKernelGGL(stream_kernel<16>, dim3(256), dim3(1024), 0, st, src, pw, in, out);
        } else {
          const int grid = g == 1 ? 1024 : 2048;
          const size_t pw = (bytes[k] / ((size_t)grid * 4) + 1023) / 1024 * 1024;
          hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(256), 0, st, src, pw, in, out);
        }
        std::swap(in, out);
      }
  };
  printf("# block_floor: %d blocks x 5 dependent streaming launches (5.0 / 18.1 / 38.6 / 17.1 / 120.9 MB), 28 KB vector between them, hipGraph\n", layers);
  const char* gname[3] = {"256 x 16 waves", "1024 x 4 waves", "2048 x 4 waves"};
  for (int g = 0; g < 3; ++g)
    for (int only = -1; only < 5; ++only) {
      hipGraph_t gr;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      enqueue(g, only);
      CK(hipStreamEndCapture(st, &gr));
      CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      std::vector<float> ms;
      for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m);
      }
      std::sort(ms.begin(), ms.end());
      const double us = ms[ms.size() / 2] * 1e3 / layers;
      if (only < 0) printf("%s: %.2f us per block (median of 10), %.0f GB/s over the block's 199.7 MB\n", gname[g], us, 199.7 / us * 1e3);
      else printf("    launch %d alone (%.1f MB, back to back x %d): %.2f us, %.0f GB/s\n", only, mb[only], layers, us, mb[only] / us * 1e3);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(gr));
    }
  return 0;
}
