// scalar_coherence_probe.hip -- can the SCALAR memory path see what another XCD writes during the same launch?
// tools/order_probe.hip shows that scalar loads are not in a CU's in-order vector memory queue (0.45 us while the CU has 190 KB of
// reads outstanding, against 4.7 us for any vector operation).  A hand-off wait that polls through the scalar path - and a copy of a
// few KB through it - would not queue behind the consumer's own weight requests.  But MI355X has eight XCDs with an L2 each, the
// scalar cache sits in front of its XCD's L2, and the flags / hidden vectors of the fused expert launch are written by OTHER XCDs
// (agent-scope write-through stores, read back with sc1 vector loads today).  This probe measures whether a scalar load gets there.
//
// 256 workgroups of one wave.  Even workgroups are PRODUCERS: after ~3 us they store the launch's sequence number to their flag
// (agent-scope store, the engine's publishing form).  Workgroup b + 1 - a different XCD: workgroup b runs on XCD b % 8 - is the
// CONSUMER of flag b: it first reads the flag (pulling the line, with the PREVIOUS launch's value, into whatever caches its path
// has), then polls until it sees the new value or gives up after LIMIT polls:
//   mode 0  vector load, agent scope (sc1): what the engine does today
//   mode 1  s_load_dword ... glc
//   mode 2  s_dcache_inv ; s_load_dword
//   mode 3  s_load_dword (plain)
// Prints, per mode: how many consumers saw the value, and the time from the producer's store to the consumer's sighting.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/scalar_coherence_probe tools/scalar_coherence_probe.hip
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
constexpr int LIMIT = 20000;  // polls before a consumer gives up (bounded: a stale path must not hang the box)

template <int MODE>
__device__ __forceinline__ u32 poll_once(const u32* p) {
  u32 r;
  if (MODE == 0) {
    r = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (MODE == 1) {
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  } else if (MODE == 2) {
    asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  } else {
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  }
  return r;
}

// flags: one 256-byte line group per producer (no false sharing); out[b] = {polls, ticks from store to sighting, value seen, store time lo}
template <int MODE>
__global__ __launch_bounds__(64) void probe_kernel(u32* flags, unsigned long long* stamp, u32 seq, u32* out) {
  const int bid = blockIdx.x, lane = threadIdx.x;
  if ((bid & 1) == 0) {  // producer
    for (int i = 0; i < 14; ++i) __builtin_amdgcn_s_sleep(8);  // ~3 us: the consumers have read the old value by then
    if (lane == 0) {
      stamp[bid] = __builtin_amdgcn_s_memrealtime();
      __hip_atomic_store(flags + bid * 64, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const u32* p = flags + (bid - 1) * 64;
  u32 seen = poll_once<MODE>(p);  // the stale value, on purpose
  int polls = 0;
  while (seen != seq && polls < LIMIT) {
    seen = poll_once<MODE>(p);
    ++polls;
    __builtin_amdgcn_s_sleep(1);
  }
  const unsigned long long t = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    out[bid * 4 + 0] = (u32)polls;
    out[bid * 4 + 1] = seen;
    *reinterpret_cast<unsigned long long*>(out + bid * 4 + 2) = t;
  }
}

template <int MODE>
static void run(const char* name, u32* flags, unsigned long long* stamp, u32* out) {
  std::vector<u32> h(256 * 4);
  std::vector<unsigned long long> hs(256);
  int ok_total = 0, n_total = 0;
  std::vector<double> lat;
  for (u32 seq = 1; seq <= 6; ++seq) {
    hipLaunchKernelGGL(probe_kernel<MODE>, dim3(256), dim3(64), 0, 0, flags, stamp, seq + 100 * (MODE + 1), out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), stamp, hs.size() * 8, hipMemcpyDeviceToHost));
    if (seq == 1) continue;  // the first launch has no "previous value" in anybody's cache
    for (int b = 1; b < 256; b += 2) {
      ++n_total;
      if (h[b * 4 + 1] == seq + 100 * (MODE + 1) && h[b * 4 + 0] < (u32)LIMIT) {
        ++ok_total;
        const unsigned long long t = *reinterpret_cast<unsigned long long*>(&h[b * 4 + 2]);
        lat.push_back(((double)t - (double)hs[b - 1]) / 100.0);
      }
    }
  }
  std::sort(lat.begin(), lat.end());
  if (lat.empty()) printf("%-28s saw the new value: %4d of %4d consumers\n", name, ok_total, n_total);
  else
    printf("%-28s saw the new value: %4d of %4d consumers; store -> sighting: median %.2f us, p90 %.2f, max %.2f\n", name, ok_total, n_total,
           lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back());
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; scalar_coherence_probe: a consumer on another XCD polls a flag its neighbour stores (agent scope) ~3 us into the\n"
         "# launch, having read the previous launch's value first; gives up after %d polls\n", prop.gcnArchName, prop.multiProcessorCount, LIMIT);
  u32 *flags, *out;
  unsigned long long* stamp;
  CK(hipMalloc(&flags, 256 * 256));
  CK(hipMemset(flags, 0, 256 * 256));
  CK(hipMalloc(&stamp, 256 * 8));
  CK(hipMemset(stamp, 0, 256 * 8));
  CK(hipMalloc(&out, 256 * 16));
  run<0>("vector load, agent scope", flags, stamp, out);
  run<1>("s_load_dword glc", flags, stamp, out);
  run<2>("s_dcache_inv + s_load_dword", flags, stamp, out);
  run<3>("s_load_dword", flags, stamp, out);
  return 0;
}
