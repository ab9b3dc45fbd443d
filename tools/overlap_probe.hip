// overlap_probe.hip -- does a chain of DEPENDENT weight-streaming kernels run faster when consecutive kernels sit on
// two alternating streams and order themselves with in-kernel done/wait counters instead of same-stream kernel
// boundaries?  (Design question behind the engine's "overlapped token" mode)
//
// A "layer" is six launches shaped like the DeepSeek-V3 Q2_K decode layer (bytes streamed per launch, workgroup
// counts and sizes): qkv_a 5 MB, attention 21 MB on 128 workgroups, wo 39 MB, router + shared 17 MB on 192, experts
// w1/w3 77 MB, experts w2 44 MB.  Every kernel: [prefetch its first weight chunk] -> wait until the predecessor's
// done counter reaches its workgroup count (bounded spin) -> read the predecessor's 7168-float output vector (sc1
// loads), reduce it (stands for rmsnorm + Q8_K) -> stream its share of the weights -> write its slice of the output
// vector (sc1 stores) -> drain -> one arrival on its own done counter.
//
//   mode 0  one stream, no counters: the dependency is the kernel boundary               (what the engine does today)
//   mode 1  one stream, counters armed (shows the cost of the protocol alone)
//   mode 2  two alternating streams + counters, eager launches
//   mode 3  two alternating streams + counters, each stream's launches captured in its own hipGraph
//   mode 4  one stream, hipExtAnyOrderLaunch + counters (documented as unsupported on gfx9: probe)
//   mode 5  as 1, arrival counters sharded over 8 words 256 B apart (workgroup index & 7)
//   mode 6  as 4 with the sharded counters
// Each mode with prefetch = 0 / 1 (1: a kernel requests its first weight chunk BEFORE it waits for its predecessor).
// MI355X, round 3, us per layer at prefetch 0 / 1: mode 0 76.8 / 74.1, 1 94.5 / 92.7, 2 159 / 137, 3 160 / 136, 4 88.2 / 81.9,
// 5 80.8 / 77.6, 6 77.9 / 73.7: any-order launches do overlap on gfx950, and the best counter scheme equals plain boundaries.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/overlap_probe tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32 gu32;

struct KArgs {
  const u32x4* w;        // this launch's weights
  size_t n16;            // 16-byte items
  const float* act_in;   // predecessor's output (7168 floats)
  float* act_out;        // this launch's output (7168 floats)
  u32* wait_ctr;         // predecessor's done counter (null: no wait)
  u32 wait_target;
  u32* done_ctr;         // this launch's done counter (null: none)
  u32* err;              // set to 1 when a spin gives up
  int prefetch;          // 1: request the first chunk before waiting
  int shards;            // 1: one counter word; 8: eight words 256 B apart, arrivals dealt by workgroup index & 7
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NT>
__global__ __launch_bounds__(NT) void stream_kernel(const KArgs a) {
  __shared__ float red[16];
  __shared__ float lact[7168];
  const int tid = threadIdx.x, bid = blockIdx.x, nb = gridDim.x;
  const size_t per = (a.n16 + nb - 1) / nb;
  const size_t lo = per * bid, hi = lo + per < a.n16 ? lo + per : a.n16;
  u32x4 c[4];
  size_t i = lo + tid;
  if (a.prefetch) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + (size_t)u * NT < hi) c[u] = __builtin_nontemporal_load(a.w + i + (size_t)u * NT);
  }
  if (a.wait_ctr) {
    if (tid < 64) {  // lane k < shards watches word k (fan-in of 256 arrivals on ONE word is ~3 us: MI355X_MICROARCH fanin row)
      unsigned spins = 0;
      for (;;) {
        // shard k receives the arrivals of the predecessor's workgroups b with b % shards == k
        const u32 want = a.shards == 1 ? a.wait_target : (a.wait_target + a.shards - 1 - tid) / a.shards;
        const bool ok = tid >= a.shards || __hip_atomic_load(a.wait_ctr + tid * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 16)) { if (tid == 0) *a.err = 1; break; }
      }
    }
    __syncthreads();
  }
  // "staging": read the predecessor's vector with sc1 loads, reduce, park it in LDS
  float ss = 0.f;
  for (int k = tid; k < 7168; k += NT) {
    const float v = __hip_atomic_load(a.act_in + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lact[k] = v;
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
  for (int k = 0; k < NT / 64; ++k) tot += red[k];
  const float scale = 1.0f / sqrtf(tot / 7168.f + 1e-6f);
  __syncthreads();
  // stream
  u32 acc = 0;
  if (!a.prefetch) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + (size_t)u * NT < hi) c[u] = __builtin_nontemporal_load(a.w + i + (size_t)u * NT);
  }
  while (i < hi) {
    u32x4 n[4];
    const size_t j = i + (size_t)4 * NT;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j + (size_t)u * NT < hi) n[u] = __builtin_nontemporal_load(a.w + j + (size_t)u * NT);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + (size_t)u * NT < hi) {
        const u32 l = __builtin_bit_cast(u32, lact[(tid * 4 + u) % 7168]);
        acc += __builtin_amdgcn_sdot4((int)c[u].x, (int)l, 0, false) + __builtin_amdgcn_sdot4((int)c[u].y, (int)l, 0, false) +
               __builtin_amdgcn_sdot4((int)c[u].z, (int)l, 0, false) + __builtin_amdgcn_sdot4((int)c[u].w, (int)l, 0, false);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = n[u];
    i = j;
  }
  float r = wave_sum((float)(acc & 0xff) * scale * 1e-3f);
  if ((tid & 63) == 0) red[tid >> 6] = r;
  __syncthreads();
  // this workgroup's slice of the output vector: 7168 / nb floats (sc1 stores)
  const int o_lo = (int)((size_t)7168 * bid / nb), o_hi = (int)((size_t)7168 * (bid + 1) / nb);
  for (int k = o_lo + tid; k < o_hi; k += NT)
    __hip_atomic_store(a.act_out + k, red[k % (NT / 64)] * 1e-3f + lact[k] * 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.done_ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.done_ctr + (a.shards == 1 ? 0 : (bid % a.shards) * 64), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct Shape { const char* name; double mb; int wgs; int nt; };
static const Shape LAYER[6] = {{"qkv_a", 5.0, 256, 256},   {"attn", 20.8, 128, 1024}, {"wo", 38.6, 256, 1024},
                               {"router", 17.1, 192, 1024}, {"w13", 77.2, 256, 1024},  {"w2", 43.7, 1008, 256}};

static void launch(const Shape& s, const KArgs& a, hipStream_t st, int any_order) {
  void* args[] = {(void*)&a};
  const void* f = s.nt == 1024 ? (const void*)stream_kernel<1024> : (const void*)stream_kernel<256>;
  if (any_order) CK(hipExtLaunchKernel(f, dim3(s.wgs), dim3(s.nt), args, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch));
  else CK(hipLaunchKernel(f, dim3(s.wgs), dim3(s.nt), args, 0, st));
}

int main(int argc, char** argv) {
  const int n_layers = argc > 1 ? atoi(argv[1]) : 61;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int copies = 5;  // rotate the weights through > 256 MB (Infinity Cache) -- 5 x 203 MB
  CK(hipSetDevice(0));
  std::vector<u32x4*> w(copies * 6);
  for (int cidx = 0; cidx < copies; ++cidx)
    for (int k = 0; k < 6; ++k) {
      const size_t bytes = (size_t)(LAYER[k].mb * 1e6) / 16 * 16;
      CK(hipMalloc((void**)&w[cidx * 6 + k], bytes));
      CK(hipMemset(w[cidx * 6 + k], 0x5a, bytes));
    }
  float* act;
  CK(hipMalloc((void**)&act, 2 * 7168 * 4));
  std::vector<float> h(2 * 7168, 1.0f);
  CK(hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const int NL = n_layers * 6;
  u32 *ctr, *err;
  CK(hipMalloc((void**)&ctr, (size_t)(NL + 1) * 512 * 4));
  CK(hipMalloc((void**)&err, 4));
  CK(hipMemset(err, 0, 4));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, ev_fork, ev_join;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));

  auto kargs = [&](int idx, bool counters, int prefetch, int shards = 1) {
    const int k = idx % 6, layer = idx / 6;
    KArgs a;
    a.w = w[(layer % copies) * 6 + k];
    a.n16 = (size_t)(LAYER[k].mb * 1e6) / 16;
    a.act_in = act + (idx & 1) * 7168;
    a.act_out = act + ((idx + 1) & 1) * 7168;
    a.wait_ctr = counters && idx > 0 ? ctr + (size_t)(idx - 1) * 512 : nullptr;
    a.wait_target = idx > 0 ? (u32)LAYER[(idx - 1) % 6].wgs : 0;
    a.done_ctr = counters ? ctr + (size_t)idx * 512 : nullptr;
    a.shards = shards;
    a.err = err;
    a.prefetch = prefetch;
    return a;
  };

  for (int prefetch = 0; prefetch <= 1; ++prefetch)
    for (int mode = 0; mode <= 6; ++mode) {
      hipGraphExec_t ga = nullptr, gb = nullptr;
      if (mode == 3) {
        for (int par = 0; par < 2; ++par) {
          hipStream_t st = par ? sb : sa;
          hipGraph_t g;
          CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
          for (int idx = par; idx < NL; idx += 2) launch(LAYER[idx % 6], kargs(idx, true, prefetch), st, 0);
          CK(hipStreamEndCapture(st, &g));
          CK(hipGraphInstantiate(par ? &gb : &ga, g, nullptr, nullptr, 0));
          CK(hipGraphDestroy(g));
        }
      }
      double best = 1e30;
      for (int r = 0; r < reps + 1; ++r) {
        CK(hipMemsetAsync(ctr, 0, (size_t)(NL + 1) * 512 * 4, sa));
        CK(hipEventRecord(e0, sa));
        if (mode == 0 || mode == 1 || mode >= 4) {
          // 5: same stream + counters on 8 shards; 6: any-order launches + 8 shards
          for (int idx = 0; idx < NL; ++idx) launch(LAYER[idx % 6], kargs(idx, mode != 0, prefetch, mode >= 5 ? 8 : 1), sa, mode == 4 || mode == 6);
        } else {
          CK(hipEventRecord(ev_fork, sa));
          CK(hipStreamWaitEvent(sb, ev_fork, 0));
          if (mode == 2) {
            for (int idx = 0; idx < NL; ++idx) launch(LAYER[idx % 6], kargs(idx, true, prefetch), (idx & 1) ? sb : sa, 0);
          } else {
            CK(hipGraphLaunch(ga, sa));
            CK(hipGraphLaunch(gb, sb));
          }
          CK(hipEventRecord(ev_join, sb));
          CK(hipStreamWaitEvent(sa, ev_join, 0));
        }
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
      }
      u32 herr = 0;
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("prefetch=%d mode=%d  %8.3f ms / %d layers = %7.2f us per layer%s\n", prefetch, mode, best, n_layers,
             best * 1e3 / n_layers, herr ? "  [SPIN TIMEOUT]" : "");
      fflush(stdout);
      CK(hipMemset(err, 0, 4));
      if (ga) CK(hipGraphExecDestroy(ga));
      if (gb) CK(hipGraphExecDestroy(gb));
    }
  return 0;
}
