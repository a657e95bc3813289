// Probe: does a kernel start faster when the PREVIOUS kernel of the chain has touched its first bytes?
//     hipcc --offload-arch=gfx950 -O3 tools/probe/next_kernel_prefetch_probe.hip -o /tmp/pf && /tmp/pf
// A single request's round is ~230 dependent skinny-GEMM launches whose per-launch constant (4.3-4.9 us, tools/fp8_k_sweep.py) is a fifth of the
// round.  Part of that constant is the first weight tiles' trip from HBM.  Here: a chain of streaming kernels shaped like the single-request
// GEMM (G workgroups x 4 waves, each wave walks its contiguous CH / 4 bytes with 4 KiB in flight, a different buffer per launch so that every
// launch is HBM-cold), where kernel i can also touch the first PF bytes of every wave's range of kernel i + 1 (workgroup b for workgroup b:
// the same XCD under round-robin dispatch, so the lines land in the L2 that will want them).  Reported: us per launch without / with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ld_nt(u32x4_t& w, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(w) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void ld_plain(u32x4_t& w, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w) : "v"(voff), "s"(sbase) : "memory");
}

// mode bit 0: prefetch the next buffer's heads at the START of this kernel (overlaps this kernel's own stream); bit 1: at the END
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* __restrict__ W, const unsigned char* __restrict__ next, unsigned* out, int ch,
                                                     int pf, int mode) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t off = (size_t)blockIdx.x * ch + (size_t)wave * (ch / 4);
  const unsigned char* base = W + off;
  const unsigned voff = lane * 16;
  unsigned acc = 0;
  u32x4_t pfv[4] = {};
  const int npf = pf / 1024;
  if (next && (mode & 1)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < npf) ld_plain(pfv[i], voff, next + off + i * 1024);
  }
  u32x4_t r[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) ld_nt(r[d], voff, base + d * 1024);
  const int steps = ch / 4 / 1024;
  for (int s = 4; s < steps; s += 4) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(r[d])::"memory");
      acc ^= r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
      ld_nt(r[d], voff, base + (size_t)(s + d) * 1024);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
#pragma unroll
  for (int d = 0; d < 4; ++d) acc ^= r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
  if (next && (mode & 2)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < npf) ld_plain(pfv[i], voff, next + off + i * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pfv[0]), "+v"(pfv[1]), "+v"(pfv[2]), "+v"(pfv[3])::"memory");
  if (next && mode) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= pfv[i].x & 1u;  // (keeps the prefetch loads alive; their value is irrelevant)
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main() {
  const int NB = 12;
  for (int G : {128, 344, 688}) {
    for (int ch_kib : {256, 512}) {
      const int ch = ch_kib * 1024;
      const size_t bytes = (size_t)G * ch;
      std::vector<unsigned char*> bufs(NB);
      for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
      unsigned* out; CK(hipMalloc(&out, 4096 * 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto run = [&](int mode, int pf) -> float {
        const int iters = 240;
        for (int i = 0; i < 24; ++i) stream_kernel<<<G, 256>>>(bufs[i % NB], mode ? bufs[(i + 1) % NB] : nullptr, out, ch, pf, mode);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) stream_kernel<<<G, 256>>>(bufs[i % NB], mode ? bufs[(i + 1) % NB] : nullptr, out, ch, pf, mode);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / iters;
      };
      const float base = run(0, 0);
      printf("G %4d  %3d KiB per workgroup (%6.1f MB per launch): no prefetch %6.2f us (%5.2f TB/s)", G, ch_kib, bytes / 1e6, base, bytes / base / 1e6);
      for (int pf : {1024, 4096}) printf(" | next's first %d B per wave at START %6.2f, at END %6.2f", pf, run(1, pf), run(2, pf));
      printf("\n");
      fflush(stdout);
      for (auto& b : bufs) CK(hipFree(b));
      CK(hipFree(out));
    }
  }
  return 0;
}
