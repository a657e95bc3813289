// Probe: what ONE CU can ingest, by how many bytes it keeps in flight.   hipcc --offload-arch=gfx950 -O3 tools/probe/cu_ingest_probe.hip -o /tmp/cuprobe
// One workgroup per CU (dynamic LDS of 100 KiB forbids a second one), C workgroups, WAVES waves each; every wave keeps DEPTH 1 KiB
// `global_load_dwordx4 nt` instructions in flight (register ring: a load is re-issued the moment its predecessor was consumed, the
// queue is never drained: hand-counted vmcnt) and xor-folds what lands.  Each workgroup walks its own 1 MiB (the weight share of a
// wide-cohort GEMM workgroup) — mode 0: a different MiB per workgroup and per launch (HBM), mode 1: the SAME MiB for every
// workgroup (L2 / MALL resident: the activation side of those kernels).
// Question (DESIGN §4 "Four requests per weight pass"): is the ~36-38 GB/s a CU sustains a cap of the CU's load path, or Little's law
// on 16 waves x 4 KiB?  Read the table by rows of equal WAVES x DEPTH.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ void ld(u32x4_t& w, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(w) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait(u32x4_t& w) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w) : "i"(N) : "memory"); }

template <int DEPTH>
__global__ __launch_bounds__(1024) void probe(const unsigned char* __restrict__ W, unsigned* out, long wg_stride, int kib_per_wave) {
  extern __shared__ unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned char* base = W + (size_t)blockIdx.x * wg_stride + (size_t)wave * kib_per_wave * 1024;
  const unsigned voff = lane * 16;
  u32x4_t r[DEPTH];
  unsigned acc = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {  // prologue: DEPTH loads in flight (offsets d KiB)
    if (d == 0) ld<0>(r[0], voff, base);
    if (d == 1) ld<1024>(r[d < DEPTH ? d : 0], voff, base);
    if (d == 2) ld<2048>(r[d < DEPTH ? d : 0], voff, base);
    if (d == 3) ld<3072>(r[d < DEPTH ? d : 0], voff, base);
    if (d >= 4) { const unsigned char* b2 = base + d * 1024; ld<0>(r[d < DEPTH ? d : 0], voff, b2); }
  }
  const int groups = kib_per_wave / DEPTH;
  for (int g = 1; g < groups; ++g) {
    const unsigned char* nb = base + (size_t)g * DEPTH * 1024;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      wait<DEPTH - 1>(r[d]);  // the oldest of the DEPTH outstanding loads
      acc ^= r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
      const unsigned char* b2 = nb + d * 1024;
      ld<0>(r[d], voff, b2);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    wait<0>(r[d]);
    acc ^= r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
  }
  if (acc == 0x12345u) out[threadIdx.x] = acc + smem[0];
}

template <int DEPTH>
static int run(const unsigned char* W, size_t bytes, unsigned* out, int C, int waves, int mode) {
  const int kib_per_wave = 1024 / waves;  // 1 MiB per workgroup
  const long wg_stride = mode == 0 ? (1l << 20) : 0;
  const size_t per_launch = mode == 0 ? (size_t)C << 20 : (size_t)1 << 20;
  const int slots = (int)(bytes / per_launch);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 40;
  for (int i = 0; i < 3; ++i) probe<DEPTH><<<C, waves * 64, 100 * 1024>>>(W + (size_t)(i % slots) * per_launch, out, wg_stride, kib_per_wave);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) probe<DEPTH><<<C, waves * 64, 100 * 1024>>>(W + (size_t)((i + 3) % slots) * per_launch, out, wg_stride, kib_per_wave);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("mode %d  C %3d  waves %2d  depth %2d  in flight/CU %4d KiB : %6.1f us/launch  %6.1f GB/s per CU  %5.2f TB/s chip\n", mode, C, waves, DEPTH,
         waves * DEPTH, us, (1 << 20) / us / 1e3, (double)C * (1 << 20) / us / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  const size_t bytes = 4ull << 30;
  unsigned char* W; unsigned* out;
  CK(hipMalloc(&W, bytes)); CK(hipMemset(W, 1, bytes)); CK(hipMalloc(&out, 4096));
  for (int mode = 0; mode < 2; ++mode)
    for (int C : {32, 96, 128, 172, 256})
      for (int waves : {4, 8, 16}) {
        if (run<2>(W, bytes, out, C, waves, mode)) return 1;
        if (run<4>(W, bytes, out, C, waves, mode)) return 1;
        if (run<8>(W, bytes, out, C, waves, mode)) return 1;
        if (run<16>(W, bytes, out, C, waves, mode)) return 1;
      }
  return 0;
}
