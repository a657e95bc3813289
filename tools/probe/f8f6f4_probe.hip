#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// A: [32 rows][64 k] fp8 row-major ; B: [32 cols][64 k] fp8 (i.e. B^T row-major) ; D = A * B^T [32][32]
// lane l supplies for A: row l&31, bytes k = 32*(l>>5) + [0,32)  (hypothesis); same for B with col l&31.
__global__ void probe(const unsigned char* A, const unsigned char* B, float* D) {
  const int l = threadIdx.x, r = l & 31, hi = l >> 5;
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(A + r * 64 + 32 * hi + 4 * i);
    b[i] = *reinterpret_cast<const int*>(B + r * 64 + 32 * hi + 4 * i);
  }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);  // cbsz = 0 (fp8 e4m3), blgp = 0 (fp8 e4m3), scales 2^0
  for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}
static float e4m3(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main() {
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  for (int i = 0; i < 32 * 64; ++i) { A[i] = (unsigned char)((i * 37 + 11) % 120); B[i] = (unsigned char)(((i * 53 + 7) % 120) | ((i & 5) == 1 ? 0x80 : 0)); }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 64 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  std::vector<float> D(64 * 16);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  // expected with the 32x32x16-bf16 output layout used by the library: lane (j = l & 31, hi), register 4q + r -> D[i = 8q + 4hi + r][j]
  double worst = 0;
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) {
    int i = 8 * q + 4 * (l >> 5) + r, j = l & 31;
    double want = 0, mag = 0;
    for (int k = 0; k < 64; ++k) { const double t = (double)e4m3(A[i * 64 + k]) * e4m3(B[j * 64 + k]); want += t; mag += fabs(t); }
    // (error against the sum of |products|: a wrong operand layout leaves O(1); the right one leaves what the pipe's internal alignment of 64
    //  products of very different magnitude drops — the first run of this probe saw 7e-4 of the RESULT, i.e. not an exact fp32 sum of products)
    double err = fabs(want - D[l * 16 + 4 * q + r]) / (mag + 1e-30);
    if (err > worst) worst = err;
  }
  printf("f8f6f4 32x32x64: worst error / sum |products| vs the assumed layouts = %.3g (%s)\n", worst, worst < 1e-2 ? "layout confirmed" : "LAYOUT MISMATCH");
  return 0;
}
