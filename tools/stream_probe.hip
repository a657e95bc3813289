// Probe: what limits a weight-streaming skinny GEMM on MI355X?  hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// MODE 0: read only (xor-reduce)  1: + MFMA with constant B   2: + X loads (B from global, fragment-shaped, L2-resident)
// MODE 3: + X loads fully coalesced (1 KiB contiguous per instruction, L2-resident)  4: mode 3 through LDS (write + read back)
// MODE 5: X staged in LDS once per 4 waves (shared), all waves read it
template <int MODE, int UN>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ W, const unsigned short* __restrict__ X, int ldx, float* out,
                                             long blocks_per_wave /*1KB blocks*/, int ksteps_total) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + wave;
  const uint4* p = W + gw * blocks_per_wave * 64 + lane;
  const unsigned short* px = X + (size_t)(lane & 31) * ldx + (lane >> 5) * 8 + ((gw * blocks_per_wave) % ksteps_total) * 16;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  uint4 x0 = make_uint4(0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80);
  uint4 xr = make_uint4(0,0,0,0);
  __shared__ uint4 lds[4][UN][66];
  const uint4* pxc = reinterpret_cast<const uint4*>(X) + lane + ((gw * blocks_per_wave) % ksteps_total) * 64;  // contiguous 1 KiB blocks
  for (long b = 0; b < blocks_per_wave; b += UN) {
    uint4 a[UN], bb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      a[u] = p[(b + u) * 64];
      if (MODE == 2) bb[u] = *reinterpret_cast<const uint4*>(px + (b + u) * 16);
      if (MODE == 3 || MODE == 4) bb[u] = pxc[((b + u) % 64) * 64];
      if (MODE == 5 && wave == (u & 3)) bb[u] = pxc[((b + u) % 64) * 64];
    }
    if (MODE == 4) {
#pragma unroll
      for (int u = 0; u < UN; ++u) lds[wave][u][lane] = bb[u];
#pragma unroll
      for (int u = 0; u < UN; ++u) bb[u] = lds[wave][u][lane ^ 1];
    }
    if (MODE == 5) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < UN; ++u) if (wave == (u & 3)) lds[0][u][lane] = bb[u];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < UN; ++u) bb[u] = lds[0][u][lane ^ 1];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (MODE == 0) { xr.x ^= a[u].x; xr.y ^= a[u].y; xr.z ^= a[u].z; xr.w ^= a[u].w; }
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[u]), *reinterpret_cast<bf16x8*>(MODE >= 2 ? &bb[u] : &x0), acc, 0, 0, 0);
    }
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 12345.f || (xr.x ^ xr.y ^ xr.z ^ xr.w) == 0x12345u) out[threadIdx.x] = s;
}

int main() {
  const size_t bytes = 4ull << 30;  // 4 GiB of "weights", streamed in slices so nothing is cache resident
  uint4* W; unsigned short* X; float* out;
  CK(hipMalloc(&W, bytes)); CK(hipMemset(W, 1, bytes));
  const int K = 4096, ldx = K;
  CK(hipMalloc(&X, 32 * K * 2)); CK(hipMemset(X, 0, 32 * K * 2)); CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t slice = 256ull << 20;  // one launch = 256 MiB (lm_head-sized)
  struct Cfg { int wgs; };
  for (int mode = 1; mode < 6; ++mode)
    for (int un : {4, 8})
      for (int wgs : {512, 1024, 2048}) {
        long blocks = slice / 1024 / ((long)wgs * 4);
        if (blocks % un) continue;
        auto launch = [&](int i) {
          const uint4* base = W + (size_t)(i % 16) * (slice / 16);
#define L(M, U) hipLaunchKernelGGL((probe<M, U>), dim3(wgs), dim3(256), 0, 0, base, X, ldx, out, blocks, K / 16)
          if (mode == 1) { if (un == 4) L(1, 4); else L(1, 8); }
          if (mode == 2) { if (un == 4) L(2, 4); else L(2, 8); }
          if (mode == 3) { if (un == 4) L(3, 4); else L(3, 8); }
          if (mode == 4) { if (un == 4) L(4, 4); else L(4, 8); }
          if (mode == 5) { if (un == 4) L(5, 4); else L(5, 8); }
        };
        for (int i = 0; i < 3; ++i) launch(i);
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        const int iters = 32;
        for (int i = 0; i < iters; ++i) launch(i);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d un %2d wgs %5d : %7.1f us/launch  %5.2f TB/s\n", mode, un, wgs, ms * 1e3 / iters, slice / (ms / iters * 1e-3) / 1e12);
      }
  return 0;
}
