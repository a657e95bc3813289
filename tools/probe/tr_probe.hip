// Semantics probe of ds_read_b64_tr_b16 on gfx950: LDS holds element index values, lane l supplies byte address addr[l];
// prints, for every lane and result element j, the index of the LDS element returned.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
  unsigned r0, r1;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(unsigned long long*)&r0) : "v"(a) : "memory");
  (void)r1;
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__global__ void probe2(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff;
  out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff;
  out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
  for (int variant = 0; variant < 3; ++variant) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) addr[l] = l * 8;                       // contiguous 8 B per lane
      if (variant == 1) addr[l] = (l & 15) * 256 + (l >> 4) * 8;  // lane&15 = row (256-B stride), lane>>4 = 8-B column group
      if (variant == 2) addr[l] = 1000 * 0 + 64;               // uniform
    }
    hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe2, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    std::vector<unsigned short> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("variant %d (addr bytes: lane0=%d lane1=%d lane16=%d)\n", variant, addr[0], addr[1], addr[16]);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" %5d", out[l * 4 + j]);
      if (l % 4 == 3) printf("\n"); 
    }
  }
  return 0;
}
