// Is  q = x*rc ; r = fma(-q, c, x) ; q' = fma(r, rc, q)  equal to the IEEE quotient x / c for EVERY bf16-valued x, c = sqrt(128)?
// (the eager attention path divides bf16 scores by sqrt(head_dim), modeling_llama_kv.py:602-604; the IEEE division is ~10 instructions)
// hipcc --offload-arch=gfx950 -O3 tools/div_check.hip -o /tmp/div_check && /tmp/div_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* bad, float c, float rc) {
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;  // all 65536 bf16 bit patterns
  const float x = __uint_as_float(b << 16);
  const float ref = x / c;
  float q = x * rc;
  const float r = __builtin_fmaf(-q, c, x);
  q = __builtin_fmaf(r, rc, q);
  const bool same = (__float_as_uint(q) == __float_as_uint(ref)) || (ref != ref && q != q);
  // outside the normal range the two differ (results below 2^-126, +-inf, the sign of -0): scores never live there
  if (!same && x == x && fabsf(x) >= 1e-36f && fabsf(x) < 1e38f) { atomicAdd(bad, 1u); printf("x=%g (0x%04x) ref=%.9g fast=%.9g\n", x, b, ref, q); }
}
int main() {
  unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  const float c = 11.313708498984761f, rc = 1.0f / c;
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, bad, c, rc);
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("mismatches over all bf16 inputs with 1e-36 <= |x| < 1e38: %u\n", h);
  return h != 0;
}
