// ll_probe.hip -- what does one in-kernel hand-off between workgroups on different XCDs cost on an otherwise idle chip?
//   mode 0  what the engine does: write-through stores -> s_waitcnt vmcnt(0) -> barrier -> one arrival on a counter ->
//           consumers poll the counter -> sc1 loads of the data                                     (MI355X: 4.3 us)
//   mode 1  8-byte tagged granules {payload, tag}: one write-through store each, consumers poll the granules themselves
//           (sc1 loads) until every tag is this round's                                              (MI355X: 2.1 us)
//   mode 2  as 1 with sc0 sc1 loads                                                                  (MI355X: 2.25 us)
// 256 workgroups x 1024 threads, each publishes 19 values after an uneven stretch of streaming, everybody consumes all 4864.
// Round 3 built the granule hand-over into the fused expert launch on the strength of these numbers:
// bit-identical, and NOT faster in situ - a CU serves its waves' requests in order, so the polls queue up behind the 128 KB
// of W2 rows every workgroup requests before the hand-off, and those rows have to stream through the same CU anyway.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ll_probe tools/ll_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

constexpr int NG = 4864;  // granules (8 B) or dwords of payload: 19 per workgroup

__device__ inline u32x4 ld_sc1_b128(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline u32x4 ld_sc01_b128(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(1024) void probe(int mode, u32 tag, u32* data, u32* ctr, u64* stamps, u32* err, int work, const u32x4* w) {
  const int tid = threadIdx.x, bid = blockIdx.x;
  __shared__ u32 sink[1024];
  // unequal "phase A" work so arrivals are spread like the real kernel
  u32 acc = 0;
  for (int i = 0; i < work + (bid & 7); ++i) { u32x4 v = w[(size_t)bid * 4096 + i * 1024 + tid]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  sink[tid] = acc;
  __syncthreads();
  if (tid == 0) stamps[bid * 4 + 0] = wall_clock64();
  if (mode == 0) {
    if (tid < 19) __hip_atomic_store(data + bid * 19 + tid, tag + (acc & 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stamps[bid * 4 + 1] = wall_clock64();
      u32 spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 256u * (tag)) { if (++spins > (1u << 22)) { *err = 1; break; } }
      stamps[bid * 4 + 2] = wall_clock64();
    }
    __syncthreads();
    u32 bad = 0;
    for (int i = tid; i < NG / 4; i += 1024) { u32x4 v = ld_sc1_b128(data + i * 4); bad |= (v.x != tag) | (v.y != tag) | (v.z != tag) | (v.w != tag); }
    if (bad) *err = 2;
    __syncthreads();
    if (tid == 0) stamps[bid * 4 + 3] = wall_clock64();
  } else {
    if (tid < 19) {
      u64 g = ((u64)tag << 32) | (u32)(bid * 19 + tid);
      __hip_atomic_store((u64*)data + bid * 19 + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) stamps[bid * 4 + 1] = wall_clock64();
    u32 spins = 0;
    for (int i = tid; i < NG / 2; i += 1024) {
      for (;;) {
        u32x4 v = mode == 1 ? ld_sc1_b128((u64*)data + i * 2) : ld_sc01_b128((u64*)data + i * 2);
        if (v.y == tag && v.w == tag) { if (v.x != (u32)(i * 2) || v.z != (u32)(i * 2 + 1)) *err = 2; break; }
        if (++spins > (1u << 20)) { *err = 1; break; }
      }
    }
    if (tid == 0) stamps[bid * 4 + 2] = wall_clock64();
    __syncthreads();
    if (tid == 0) stamps[bid * 4 + 3] = wall_clock64();
  }
}

int main() {
  u32 *data, *ctr, *err; u64* stamps; u32x4* w;
  CK(hipMalloc(&data, NG * 8)); CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&stamps, 256 * 4 * 8));
  CK(hipMalloc(&w, (size_t)256 * 4096 * 16 * 4));
  CK(hipMemset(w, 1, (size_t)256 * 4096 * 16 * 4));
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(data, 0, NG * 8)); CK(hipMemset(ctr, 0, 256)); CK(hipMemset(err, 0, 4));
    std::vector<double> hop, stage, tot;
    for (int it = 1; it <= 40; ++it) {
      probe<<<256, 1024>>>(mode, (u32)it, data, ctr, stamps, err, 3, w);
      CK(hipDeviceSynchronize());
      std::vector<u64> s(256 * 4);
      CK(hipMemcpy(s.data(), stamps, 256 * 4 * 8, hipMemcpyDeviceToHost));
      u64 last0 = 0, first2 = ~0ull, last3 = 0, last2 = 0;
      for (int b = 0; b < 256; ++b) { last0 = std::max(last0, s[b * 4]); first2 = std::min(first2, s[b * 4 + 2]); last2 = std::max(last2, s[b * 4 + 2]); last3 = std::max(last3, s[b * 4 + 3]); }
      if (it > 8) { hop.push_back(((double)first2 - (double)last0) / 100.0); stage.push_back(((double)last2 - (double)last0) / 100.0); tot.push_back(((double)last3 - (double)last0) / 100.0); }
    }
    u32 e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("mode %d err %u: after the LAST workgroup's data was ready: first pass %.2f us, last pass %.2f us, last done (data in hand) %.2f us\n", mode, e, med(hop), med(stage), med(tot));
  }
  return 0;
}
