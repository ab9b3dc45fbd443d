// scalar_store_probe.hip -- can a workgroup PUBLISH through the scalar memory path (s_store_dword, s_dcache_wb, s_atomic_add), i.e.
// outside its CU's in-order vector-memory queue, and does a consumer on ANOTHER XCD see the data?
// tools/order_probe.hip: any vector memory operation of a CU - a store's acknowledgement, a returning atomic - waits for every read
// the CU has in flight (4.7 us behind 190 KB); the hand-over chain of the fused expert launch is four such round trips.
// tools/scalar_coherence_probe.hip: a scalar LOAD with glc is outside that queue and sees other XCDs' agent-scope stores.
// This probe: the publishing side.  256 workgroups of 5 waves; waves 0-3 of EVERY workgroup keep 48 KiB of cold reads each in flight
// (the queue), wave 4 of an even workgroup publishes 16 dwords + a flag ~3 us into the launch, wave 4 of the odd workgroup b + 1 (the
// next XCD) polls the flag with s_load_dword glc and then reads the 16 dwords twice - s_load_dwordx16 glc and sc1 vector loads.
//   mode 0  vector: 16 lanes store (agent scope) ; s_waitcnt vmcnt(0) ; returning atomic add on the flag         (the engine today)
//   mode 1  scalar: 4 x s_store_dwordx4 glc ; s_dcache_wb ; s_waitcnt lgkmcnt(0) ; s_atomic_add glc on the flag
//   mode 2  scalar, stores without glc (s_dcache_wb alone pushes them out)
// Prints per mode: publish duration on the producer (first store -> atomic returned), flag sighting latency, and how many consumers
// read the right 16 dwords through each read path.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/scalar_store_probe tools/scalar_store_probe.hip
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
typedef u32 u32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define DEV __device__ __forceinline__
DEV rsrc_t make_rsrc(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
}
constexpr int DEPTH = 48, NSTREAM = 4, LIMIT = 40000;

// data: 64 dwords (256 B) per producer; flags: one 256-byte line per producer; out[b][8]
template <int MODE>
__global__ __launch_bounds__(64 * (NSTREAM + 1)) void probe_kernel(const uint8_t* __restrict__ cold, size_t cold_stride, u32* data, u32* flags, u32 seq,
                                                                 int streams, unsigned long long* out, u32* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), bid = blockIdx.x;
  if (wave < NSTREAM) {
    if (!streams) return;
    const rsrc_t R = make_rsrc(cold + ((size_t)bid * NSTREAM + wave) * cold_stride);
    u32 acc = 0;
    for (int r = 0; r < 8; ++r) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, (r * DEPTH + d) << 10, 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    }
    if (acc == 0x12345678u) sink[bid * NSTREAM + wave] = acc;
    return;
  }
  unsigned long long* o = out + (size_t)bid * 8;
  if ((bid & 1) == 0) {  // ---- producer ----
    for (int i = 0; i < 14; ++i) __builtin_amdgcn_s_sleep(8);  // ~3 us: the streamers' requests are queued
    u32* d = data + bid * 64;
    u32* f = flags + bid * 64;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    u32 old = 0;
    if (MODE == 0) {
      if (lane < 16) __hip_atomic_store(d + lane, seq * 1000u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) old = __hip_atomic_fetch_add(f, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
    } else {
      const u32 b = seq * 1000u;
      const u32x4 v0 = {b + 0, b + 1, b + 2, b + 3}, v1 = {b + 4, b + 5, b + 6, b + 7}, v2 = {b + 8, b + 9, b + 10, b + 11}, v3 = {b + 12, b + 13, b + 14, b + 15};
      u32 add = seq;
      if (MODE == 1)
        asm volatile(
            "s_store_dwordx4 %1, %5, 0x0 glc\n\ts_store_dwordx4 %2, %5, 0x10 glc\n\ts_store_dwordx4 %3, %5, 0x20 glc\n\ts_store_dwordx4 %4, %5, 0x30 glc\n\t"
            "s_dcache_wb\n\ts_waitcnt lgkmcnt(0)\n\t"
            "s_atomic_add %0, %6, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
            : "+s"(add)
            : "s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(d), "s"(f)
            : "memory");
      else
        asm volatile(
            "s_store_dwordx4 %1, %5, 0x0\n\ts_store_dwordx4 %2, %5, 0x10\n\ts_store_dwordx4 %3, %5, 0x20\n\ts_store_dwordx4 %4, %5, 0x30\n\t"
            "s_dcache_wb\n\ts_waitcnt lgkmcnt(0)\n\t"
            "s_atomic_add %0, %6, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
            : "+s"(add)
            : "s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(d), "s"(f)
            : "memory");
      old = add;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { o[0] = t0; o[1] = t1; o[2] = old; }
    return;
  }
  // ---- consumer of producer bid - 1 (another XCD) ----
  const u32* f = flags + (bid - 1) * 64;
  const u32* d = data + (bid - 1) * 64;
  u32 seen = 0;
  int polls = 0;
  for (;;) {
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(seen) : "s"(f) : "memory");
    if (seen == seq || ++polls >= LIMIT) break;
    __builtin_amdgcn_s_sleep(1);
  }
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  u32x16 sv;
  asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(sv) : "s"(d) : "memory");
  const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
  int ok_s = 1;
#pragma unroll
  for (int i = 0; i < 16; ++i) ok_s &= sv[i] == seq * 1000u + i;
  const u32 vv = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(d), (lane & 15) * 4, 0, 16);  // sc1: what the engine's consumers do
  const int ok_v = __all(vv == seq * 1000u + (lane & 15));
  const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { o[0] = t2; o[1] = t3; o[2] = (unsigned long long)polls; o[3] = ok_s; o[4] = ok_v; o[5] = seen; o[6] = t4; }
}

// do scalar atomics of DIFFERENT XCDs combine?  Every workgroup adds 1 to ONE shared counter `reps` times (returning form, and the
// non-returning form on a second counter): the final values must be 256 * reps and the returned values all different
__global__ __launch_bounds__(64) void atomic_kernel(u32* ctr, int reps, u32* ret) {
  const int bid = blockIdx.x;
  for (int r = 0; r < reps; ++r) {
    u32 add = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(add) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) ret[bid * reps + r] = add;
    u32 one = 1;
    asm volatile("s_atomic_add %0, %1, 0x100\n\ts_waitcnt lgkmcnt(0)" : : "s"(one), "s"(ctr) : "memory");
    __builtin_amdgcn_s_sleep(3);
  }
}
static void run_atomics() {
  const int reps = 16;
  u32 *ctr, *ret;
  CK(hipMalloc(&ctr, 1024));
  CK(hipMemset(ctr, 0, 1024));
  CK(hipMalloc(&ret, 256 * reps * 4));
  hipLaunchKernelGGL(atomic_kernel, dim3(256), dim3(64), 0, 0, ctr, reps, ret);
  CK(hipDeviceSynchronize());
  std::vector<u32> h(256), r(256 * reps);
  CK(hipMemcpy(h.data(), ctr, 1024, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), ret, r.size() * 4, hipMemcpyDeviceToHost));
  std::sort(r.begin(), r.end());
  int distinct = (int)(std::unique(r.begin(), r.end()) - r.begin());
  printf("s_atomic_add from 256 workgroups on 8 XCDs x %d: returning form sums to %u (expected %d), %d distinct return values (max %u); non-returning form sums to %u\n",
         reps, h[0], 256 * reps, distinct, r[distinct - 1], h[64]);
}

template <int MODE>
static void run(const char* name, const uint8_t* cold, size_t stride, size_t cold_bytes, u32* data, u32* flags, unsigned long long* out, u32* sink) {
  for (int streams = 0; streams < 2; ++streams) {
    std::vector<unsigned long long> h(256 * 8);
    std::vector<double> pub, sight, sread, vread;
    int ok_flag = 0, ok_s = 0, ok_v = 0, n = 0, old_ok = 0;
    for (u32 seq = 1; seq <= 6; ++seq) {
      CK(hipMemset(flags, 0, 256 * 256));
      CK(hipMemset(out, 0, 256 * 64));
      const uint8_t* c = cold + (size_t)(seq % 4) * (cold_bytes / 4);
      hipLaunchKernelGGL(probe_kernel<MODE>, dim3(256), dim3(64 * (NSTREAM + 1)), 0, 0, c, stride, data, flags, seq + 10 * MODE, streams, out, sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      if (seq == 1) continue;
      for (int b = 0; b < 256; b += 2) {
        const unsigned long long *P = &h[b * 8], *C = &h[(b + 1) * 8];
        ++n;
        pub.push_back((double)(P[1] - P[0]) / 100.0);
        old_ok += P[2] == 0;
        if (C[5] == seq + 10 * MODE && C[2] < (unsigned long long)LIMIT) {
          ++ok_flag;
          sight.push_back(((double)C[0] - (double)P[0]) / 100.0);
          sread.push_back((double)(C[1] - C[0]) / 100.0);
          vread.push_back((double)(C[6] - C[1]) / 100.0);
          ok_s += C[3] != 0;
          ok_v += C[4] != 0;
        }
      }
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double>& v) { return v.empty() ? -1.0 : *std::max_element(v.begin(), v.end()); };
    printf("%-30s %s: publish %5.2f us median (max %5.2f); atomic returned 0: %d / %d; flag seen %d / %d, first store -> sighting %5.2f us (max %5.2f);\n"
           "%-30s      16 dwords right by s_load_dwordx16 glc: %d, by sc1 vector load: %d; scalar read %4.2f us, vector read %4.2f us\n",
           name, streams ? "the CU streams " : "nobody streams", med(pub), mx(pub), old_ok, n, ok_flag, n, med(sight), mx(sight), "", ok_s, ok_v, med(sread), med(vread));
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; scalar_store_probe: a producer publishes 16 dwords + a flag, the consumer sits on the next XCD\n", prop.gcnArchName, prop.multiProcessorCount);
  const size_t cold_bytes = 3ull << 30;
  uint8_t* cold;
  u32 *data, *flags, *sink;
  unsigned long long* out;
  CK(hipMalloc(&cold, cold_bytes));
  CK(hipMemset(cold, 1, cold_bytes));
  CK(hipMalloc(&data, 256 * 256));
  CK(hipMemset(data, 0, 256 * 256));
  CK(hipMalloc(&flags, 256 * 256));
  CK(hipMalloc(&out, 256 * 64));
  CK(hipMalloc(&sink, 8192));
  const size_t stride = (cold_bytes / 4 / (256 * NSTREAM)) & ~(size_t)4095;
  run_atomics();
  run<0>("vector stores + vector atomic", cold, stride, cold_bytes, data, flags, out, sink);
  run<1>("s_store glc + wb + s_atomic", cold, stride, cold_bytes, data, flags, out, sink);
  run<2>("s_store + wb + s_atomic", cold, stride, cold_bytes, data, flags, out, sink);
  return 0;
}
