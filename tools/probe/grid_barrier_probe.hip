// Cost of a grid-wide barrier + cross-XCD data hand-over on gfx950, the number that decides whether fusing two dependent phases of a
// round into one launch can beat the ~4-5 us kernel boundary.   hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o gbp && ./gbp
//   mode 0: barrier only (agent-scope atomic counter, relaxed spin with sc1 loads)
//   mode 1: + data: every workgroup publishes 1 KiB with __threadfence() (release = buffer_wbl2) before arriving, acquire fence after
//   mode 2: + data: published with write-through stores (sc0 sc1) and NO wbl2; consumer side buffer_inv sc1 only
// All spins are bounded (a lost barrier prints an error instead of hanging the GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void store_wt(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* counter, unsigned* data, int rounds, unsigned* err, unsigned long long* cycles) {
  const int G = gridDim.x, wg = blockIdx.x, t = threadIdx.x;
  unsigned long long t0 = 0;
  if (wg == 0 && t == 0) t0 = __builtin_readcyclecounter();
  for (int r = 1; r <= rounds; ++r) {
    if (MODE >= 1) {
      unsigned* mine = data + (size_t)wg * 256;
      if (MODE == 1) mine[t] = (unsigned)(r * 1000003 + wg * 256 + t);
      else store_wt(mine + t, (unsigned)(r * 1000003 + wg * 256 + t));
    }
    if (MODE == 1) __threadfence();
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * G;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 22)) { atomicAdd(err, 1u); break; }
      }
    }
    __syncthreads();
    if (MODE >= 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const int src = (wg + 37) % G;
      const unsigned got = data[(size_t)src * 256 + t];
      if (got != (unsigned)(r * 1000003 + src * 256 + t)) atomicAdd(err + 1, 1u);
    }
    __syncthreads();  // nobody overwrites its slot before every reader of the previous round is past the next barrier (next iteration's)
  }
  if (wg == 0 && t == 0) *cycles = __builtin_readcyclecounter() - t0;
}

int main() {
  unsigned *counter, *data, *err;
  unsigned long long* cyc;
  CHECK(hipMalloc(&counter, 4));
  CHECK(hipMalloc(&data, 1024 * 1024 * 4));
  CHECK(hipMalloc(&err, 8));
  CHECK(hipMalloc(&cyc, 8));
  const int rounds = 200;
  for (int mode = 0; mode < 3; ++mode)
    for (int G : {256, 512}) {
      CHECK(hipMemset(counter, 0, 4));
      CHECK(hipMemset(err, 0, 8));
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, counter, data, rounds, err, cyc);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, counter, data, rounds, err, cyc);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(G), dim3(256), 0, 0, counter, data, rounds, err, cyc);
      hipEventRecord(e1);
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned h[2];
      CHECK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
      printf("mode %d  G=%d: %.2f us per barrier round  (lost barriers %u, stale reads %u)\n", mode, G, ms * 1e3 / rounds, h[0], h[1]);
    }
  return 0;
}
