// order_probe.hip -- do a CU's memory reads return in order ACROSS waves, and what does one dependent round trip cost while the
// chip streams?  (EXPERIMENTS.md 4.2 / 4.4: the fused expert launch's hand-off is a chain of dependent round trips issued while
// every CU has ~200 KB of weight reads outstanding; whether a poll or a small L2-resident read issued by ANOTHER wave of the same
// workgroup overtakes those reads decides what a "dedicated hand-off wave" could buy.)
//
// 256 workgroups (one per CU) x 5 waves.  Waves 0..3 are STREAMERS: each requests 48 KiB of cold HBM data (nt loads, all in
// flight, consumed at the end: ~190 KB outstanding per CU, what a CU of the engine's expert launch has).  Wave 4 is the PROBE:
// after `delay` it performs ONE small operation and times it with s_memtime (shader clocks; calibrated against the 100 MHz s_memrealtime):
//   op 0  a vector load of a 64-byte line that sits in L2 (warmed by an earlier launch)        buffer_load sc0 sc1 off: plain
//   op 1  the same through the scalar cache                                                    s_load_dword
//   op 2  a returning device-scope atomic add on a per-workgroup counter                       global_atomic_add_rtn
//   op 3  a device-scope (sc1) vector load of a line another launch wrote                      the hand-off's poll
//   op 4  a device-scope (write-through) store and the wait for its acknowledgement           publishing before an arrival
//   op 5  a vector load of a cold line (HBM)                                                    what the chip's streaming does to one miss
// in three settings: streamers idle (nobody streams), streamers on OTHER CUs only (7 of 8 workgroups stream 6 batches each, every
// eighth probes, mid-stream), streamers on the SAME CU (every workgroup streams and probes).  Same-CU minus other-CU latency = what in-order return costs.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/order_probe tools/order_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

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

constexpr int DEPTH = 48;   // KiB a streamer wave keeps in flight (48 loads of 16 B per lane = 192 VGPRs)
constexpr int NSTREAM = 4;  // streamer waves per workgroup

// setting: 0 nobody streams, 1 half of the workgroups stream / the other half probe, 2 every workgroup streams and probes
template <int OP>
__global__ __launch_bounds__(64 * (NSTREAM + 1)) void probe_kernel(const uint8_t* __restrict__ cold, size_t cold_stride, const u32* __restrict__ warm,
                                                    u32* __restrict__ ctr, int setting, int delay, int rounds, u32* __restrict__ out_ticks,
                                                    u32* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), bid = blockIdx.x;
  const bool odd = ((bid >> 3) & 7) == 7;  // (workgroup b runs on XCD b % 8: 4 probing and 28 streaming workgroups on every XCD)
  const bool streams = setting == 2 || (setting == 1 && !odd);
  const bool probes = setting != 1 || odd;
  if (wave < NSTREAM) {
    if (!streams) return;
    const rsrc_t R = make_rsrc(cold + ((size_t)bid * NSTREAM + wave) * cold_stride);
    u32 acc = 0;
    for (int r = 0; r < rounds; ++r) {  // sustained: `rounds` batches of DEPTH KiB, each requested at once and then consumed
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, (r * DEPTH + d) << 10, 2);
      __builtin_amdgcn_sched_barrier(0);  // every request of the batch leaves before its first value is consumed
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    }
    if (acc == 0x12345678u) sink[bid * NSTREAM + wave] = acc;
    return;
  }
  if (!probes) return;
  for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(8);  // let the streamers' requests queue up first (64 clocks each)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  u32 got = 0;
  if (OP == 0) {
    got = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(warm), (bid & 63) * 64 + (lane & 15) * 4, 0, 0);
  } else if (OP == 1) {
    const u32* p = warm + (bid & 63) * 16;
    u32 r;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    got = r;
  } else if (OP == 2) {
    if (lane == 0) got = __hip_atomic_fetch_add(ctr + bid * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (OP == 3) {
    got = __hip_atomic_load(warm + (bid & 63) * 16 + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (OP == 4) {
    __hip_atomic_store(ctr + bid * 16 + 1 + (lane & 7), (u32)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {  // one cold line per lane group from HBM: the far end of the stride region nobody streams
    got = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(cold + ((size_t)bid * NSTREAM + NSTREAM - 1) * cold_stride + cold_stride - 4096), (lane & 15) * 4, 0, 2);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(got) : "memory");  // the value has arrived
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out_ticks[bid] = (u32)(t1 - t0);
}

__global__ void clock_kernel(unsigned long long* out) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(16);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[0] = c1 - c0; out[1] = r1 - r0;
}
__global__ void warm_kernel(u32* warm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) warm[i] = i;
}

static double CLK_MHZ = 2400.0;  // s_memtime counts shader clocks: calibrated against s_memrealtime (100 MHz) in main
template <int OP>
static void run(const char* name, const uint8_t* cold, size_t stride, u32* warm, u32* ctr, u32* ticks, u32* sink, size_t cold_bytes) {
  std::vector<u32> h(256);
  for (int setting = 0; setting < 3; ++setting) {
    std::vector<double> med;
    for (int rep = 0; rep < 6; ++rep) {
      // every repetition streams a different cold region (no cache holds 3 GiB); the warm lines are re-written each time so that
      // they sit in every XCD's view of the memory system the way a producer's output does
      const uint8_t* c = cold + (size_t)(rep % 4) * (cold_bytes / 4);
      hipLaunchKernelGGL(warm_kernel, dim3(4), dim3(256), 0, 0, warm, 1024);
      hipLaunchKernelGGL(probe_kernel<OP>, dim3(256), dim3(64 * (NSTREAM + 1)), 0, 0, c, stride, (const u32*)warm, ctr, setting, 12, 6, ticks, sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), ticks, 256 * 4, hipMemcpyDeviceToHost));
      if (rep == 0) continue;
      std::vector<u32> v;
      for (int b = 0; b < 256; ++b)
        if (setting != 1 || ((b >> 3) & 7) == 7) v.push_back(h[b]);
      std::sort(v.begin(), v.end());
      med.push_back(v[v.size() / 2] / CLK_MHZ);
      if (rep == 5) printf("%-34s setting %d: median %5.2f us  (min %5.2f, p90 %5.2f, max %5.2f; medians of the repetitions:", name, setting,
                           v[v.size() / 2] / CLK_MHZ, v.front() / CLK_MHZ, v[v.size() * 9 / 10] / CLK_MHZ, v.back() / CLK_MHZ);
    }
    for (double m : med) printf(" %.2f", m);
    printf(")\n");
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; order_probe: latency of ONE small operation by the probe wave while %d streamer waves (setting 2: of the same\n"
         "# workgroup; setting 1: of the other workgroups only, 7 of 8; setting 0: nobody) have %d KiB of cold HBM reads each in flight.\n",
         prop.gcnArchName, prop.multiProcessorCount, NSTREAM, DEPTH);
  const size_t cold_bytes = 3ull << 30;
  uint8_t* cold;
  u32 *warm, *ctr, *ticks, *sink;
  CK(hipMalloc(&cold, cold_bytes));
  CK(hipMemset(cold, 1, cold_bytes));
  CK(hipMalloc(&warm, 4096));
  CK(hipMalloc(&ctr, 256 * 64));
  CK(hipMemset(ctr, 0, 256 * 64));
  CK(hipMalloc(&ticks, 1024));
  CK(hipMalloc(&sink, 8192));
  {
    unsigned long long* d;
    unsigned long long h[2];
    CK(hipMalloc(&d, 16));
    hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, 0, d);
    CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    CLK_MHZ = (double)h[0] / ((double)h[1] / 100.0);
    printf("# s_memtime runs at %.0f MHz here\n", CLK_MHZ);
  }
  const size_t stride = (cold_bytes / 4 / (256 * NSTREAM)) & ~(size_t)4095;  // 768 KiB per streamer wave and repetition (6 rounds of 48 KiB are streamed)
  run<0>("vector load, line in L2", cold, stride, warm, ctr, ticks, sink, cold_bytes);
  run<1>("scalar load (s_load_dword)", cold, stride, warm, ctr, ticks, sink, cold_bytes);
  run<2>("returning atomic add (agent scope)", cold, stride, warm, ctr, ticks, sink, cold_bytes);
  run<3>("agent-scope load (the poll)", cold, stride, warm, ctr, ticks, sink, cold_bytes);
  run<4>("agent-scope store + vmcnt(0)", cold, stride, warm, ctr, ticks, sink, cold_bytes);
  run<5>("vector load, cold line (HBM)", cold, stride, warm, ctr, ticks, sink, cold_bytes);
  return 0;
}
