// ldsdma_probe.hip -- does LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without a VGPR in between) lift what ONE CU can stream?
// The engine's GEMV kernels stage weights through VGPRs: a 16-wave workgroup at 128 VGPRs holds ~8 sixteen-byte loads per lane
// in flight (~130 KB requested per CU) and streams ~20 GB/s per CU with 256 workgroups, ~26 GB/s with 128;
// the launches that occupy 128 CUs (per-head attention, the router's two halves) therefore run at half the chip's rate.
// LDS-DMA keeps the bytes in flight in LDS instead (up to 160 KB per CU) at no register cost.
//
// Every wave streams its own contiguous share of a large buffer and "consumes" it with four integer adds per 16 bytes:
//   mode 0  VGPR-staged: D sixteen-byte loads per lane in flight (buffer_load_dwordx4 nt), consumed in order, re-issued
//   mode 1  LDS-DMA: a wave-private ring of D slots of 1 KiB in LDS; global_load_lds_dwordx4 fills a slot, s_waitcnt vmcnt(D - 1)
//           says the oldest has landed, one ds_read_b128 per lane consumes it, the slot is re-filled (no cross-wave traffic)
//   mode 2  as 1 with the nt policy on the fills (aux = 2)
// swept over waves per workgroup (4 / 8 / 16), depth D and the grid (256 = every CU, 128 = half the chip).
// Prints GB/s of the chip and per occupied CU.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ldsdma_probe tools/ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x)                                                                                                      \
  do {                                                                                                             \
    hipError_t e_ = (x);                                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define DEV __device__ __forceinline__
DEV rsrc_t make_rsrc(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
}

// mode 0: D loads per lane in flight through VGPRs
template <int D>
__global__ __launch_bounds__(1024) void vgpr_kernel(const uint8_t* __restrict__ buf, size_t bytes_per_wave, u32* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t wid = (size_t)blockIdx.x * nw + wave;
  const rsrc_t R = make_rsrc(buf + wid * bytes_per_wave);
  const int steps = (int)(bytes_per_wave >> 10);  // 1 KiB per wave and step
  u32x4 v[D];
  u32 acc = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, d << 10, 2);
  for (int s = 0; s < steps; s += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const u32x4 w = v[d];
      acc += w.x + w.y + w.z + w.w;
      if (s + D + d < steps) v[d] = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, (s + D + d) << 10, 2);
    }
  }
  if (acc == 0x12345678u) out[wid] = acc;
}

// modes 1 / 2: a wave-private ring of D slots of 1 KiB in LDS, filled by LDS-DMA
template <int D, int AUX>
__global__ __launch_bounds__(1024) void ldsdma_kernel(const uint8_t* __restrict__ buf, size_t bytes_per_wave, u32* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const size_t wid = (size_t)blockIdx.x * nw + wave;
  const uint8_t* src = buf + wid * bytes_per_wave + lane * 16;
  uint8_t* ring = lds + (size_t)wave * D * 1024;
  const int steps = (int)(bytes_per_wave >> 10);
  u32 acc = 0;
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
  for (int d = 0; d < D; ++d) __builtin_amdgcn_global_load_lds((gptr)(src + ((size_t)d << 10)), (lptr)(ring + d * 1024), 16, 0, AUX);
  for (int s = 0; s < steps; s += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // the oldest fill has landed when at most D - 1 are outstanding (fills return in order)
      if (s + d + D <= steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // (the read goes through inline asm: hipcc orders every LDS read it can see behind ALL outstanding LDS-DMA fills with
      // s_waitcnt vmcnt(0), which would drain the ring at every step)
      u32x4 w;
      const u32 la = (u32)(size_t)(ring + d * 1024 + lane * 16);
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(la) : "memory");  // read before the slot is re-filled
      acc += w.x + w.y + w.z + w.w;
      if (s + D + d < steps) __builtin_amdgcn_global_load_lds((gptr)(src + ((size_t)(s + D + d) << 10)), (lptr)(ring + d * 1024), 16, 0, AUX);
    }
  }
  if (acc == 0x12345678u) out[wid] = acc;
}

template <typename K>
static double time_kernel(K k, int grid, int nw, size_t lds, const uint8_t* buf, size_t bpw, u32* out, int iters) {
  if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, 0, buf, bpw, out);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, 0, buf, bpw, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return (double)ms * 1e3 / iters;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; ldsdma_probe: GB/s streamed from HBM (chip | per occupied CU), every wave its own contiguous share\n", prop.gcnArchName, prop.multiProcessorCount);
  const size_t total = 3ull << 30;  // 3 GiB: no cache holds it
  uint8_t* buf;
  u32* out;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 1, total));
  CK(hipMalloc(&out, 1 << 20));
  for (int grid : {256, 128}) {
    for (int nw : {16, 8, 4}) {
      const int waves = grid * nw;
      // ~96 KiB per wave and launch at 16 waves (a per-head launch's share), more for smaller workgroups; multiple of 16 KiB
      size_t bpw = ((total / 8) / waves) & ~(size_t)16383;
      if (bpw > (1u << 20)) bpw = 1u << 20;
      const double gb = (double)bpw * waves / 1e9;
#define RUN(name, K, lds)                                                                                                    \
  do {                                                                                                                       \
    const double us = time_kernel(K, grid, nw, lds, buf, bpw, out, 8);                                                       \
    printf("grid %3d x %2d waves  %-22s %8.1f us  %7.1f GB/s  %6.2f GB/s per CU   (%zu KiB per wave)\n", grid, nw, name, us, \
           gb / us * 1e6, gb / us * 1e6 / (grid < 256 ? grid : 256), bpw >> 10);                                             \
  } while (0)
      RUN("vgpr D=4", vgpr_kernel<4>, 0);
      RUN("vgpr D=8", vgpr_kernel<8>, 0);
      RUN("vgpr D=12", vgpr_kernel<12>, 0);
      RUN("lds-dma D=4", (ldsdma_kernel<4, 0>), (size_t)nw * 4 * 1024);
      RUN("lds-dma D=8", (ldsdma_kernel<8, 0>), (size_t)nw * 8 * 1024);
      RUN("lds-dma D=8 nt", (ldsdma_kernel<8, 2>), (size_t)nw * 8 * 1024);
      if (nw <= 8) RUN("lds-dma D=16", (ldsdma_kernel<16, 0>), (size_t)nw * 16 * 1024);
      if (nw <= 8) RUN("lds-dma D=16 nt", (ldsdma_kernel<16, 2>), (size_t)nw * 16 * 1024);
      if (nw <= 4) RUN("lds-dma D=32 nt", (ldsdma_kernel<32, 2>), (size_t)nw * 32 * 1024);
    }
  }
  return 0;
}
