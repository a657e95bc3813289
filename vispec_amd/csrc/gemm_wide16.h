// PROTOTYPE (measurement + bit-identity check through vispec_gemm_skinny_tune variants 9SS05 / 9SS06 only; no product path launches it).
//
// Why: a CU ingests at most ~55 GB/s whatever the source (tools/probe/cu_ingest_probe.hip), and every workgroup of gemm_w32_wide_kernel
// reads one byte of X (L2) per byte of W (HBM): four row blocks x four K-quarters per workgroup, because the four quarter sums of a row
// block need their own accumulators (bit-identity with the single-request kernel: ((q0 + q1) + q2) + q3).  The accumulators, not the
// arithmetic, cap the row blocks per staged X byte.  This form moves the K-quarters ACROSS workgroups: a workgroup = 16 waves = SIXTEEN
// row blocks of ONE (split, K-quarter), the staged X group (128 rows x 64 k, 16 KiB, double-buffered) is shared by all sixteen — one byte of
// X per FOUR bytes of W — and each wave writes its quarter sum as fp32 (`part[(4 split + kq)][128][N]`); wide16_reduce_kernel adds the
// quarters in the single-request kernel's order and the splits in splitk_reduce_kernel's, so the result is bit-identical to today's.
// Price: 4 x 128 x N x 4 B of partials written and read once (gate|up: 45 MB next to 180 MB of weights).
#pragma once
#include "gemm_wide.h"

#define WIDE16_LDS_BYTES (2 * WIDE_BUFBYTES)

__global__ __launch_bounds__(1024) void gemm_w32_wide16_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                               float* __restrict__ part, int m_tile, int N, int K, int S, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  constexpr int NL = 4, LOADS = 4, KSTEP = 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, hi = lane >> 5;
  const int split = blockIdx.y >> 2, kq = blockIdx.y & 3;
  const int tile_raw = blockIdx.x * 16 + wave;
  const bool tile_ok = tile_raw < tiles;
  const int tile = tile_ok ? tile_raw : 0;
  const int KS = K / KSTEP, Sa = S < 0 ? -S : S;
  const int ks_lo = (int)((long)KS * split / Sa), ks_hi = (int)((long)KS * (split + 1) / Sa);
  const int len = ks_hi - ks_lo;
  const int w_lo = ks_lo + (int)((long)len * kq / 4), w_hi = ks_lo + (int)((long)len * (kq + 1) / 4);
  const int n_steps = w_hi - w_lo;
  const int G = n_steps / LOADS;  // >= 1 (checked by the host)
  const int rem = n_steps - G * LOADS;
  const int Gq = G + (rem ? 1 : 0), skip = rem ? LOADS - rem : 0;  // tail group: the last 64 k again, already-used steps x zero fragment
  f32x16 acc[NL];
#pragma unroll
  for (int mt = 0; mt < NL; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  const unsigned lds_q = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_w;
  // wave w moves piece w of the 16 (activation tile mt = w / 4, rows 8 pc .. 8 pc + 7) of every group
  const int mt_d = wave >> 2, pc_d = wave & 3;
  const int row_d = 32 * mt_d + min(8 * pc_d + (lane >> 3), m_tile - 1), g_d = (lane & 7) ^ ((4 * pc_d + (lane >> 4)) & 7);
  const unsigned xoff = ((unsigned)row_d * (unsigned)ldx + (unsigned)g_d * 8u) * 2u;
  const unsigned xdst = lds_q + (unsigned)(mt_d * 4 + pc_d) * 1024u;
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(X + (size_t)w_lo * KSTEP);
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(P) + ((size_t)tile * KS + w_lo) * 1024;
  const unsigned wvo = lane * 16;
  const unsigned rrow = (unsigned)(j >> 3) * 1024u + (unsigned)(j & 7) * 128u, fsw = (unsigned)(j >> 1) & 7u;
  unsigned ro[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) ro[t] = rrow + ((((unsigned)(2 * t + hi)) ^ fsw) << 4);
  u32x4_t w[LOADS];
#define W16_DMA(grp, buf) wide_dma16(xoff, xsrc + (size_t)min((grp) * LOADS, n_steps - LOADS) * (KSTEP * 2), xdst + (buf) * WIDE_BUFBYTES);
#define W16_MFMA(u, xb, SK)                                                                                     \
  _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                           \
    uint4 bv = *reinterpret_cast<const uint4*>((xb) + mt * 4096 + ro[u]);                                       \
    if ((u) < (SK)) bv = make_uint4(0, 0, 0, 0);                                                                \
    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&w[u]), as_bf16x8(bv), acc[mt], 0, 0, 0); \
  }
#define W16_STEP_PREF(u)                    \
  wide_wait_vm<LOADS - 1 + 1>(w[u]);        \
  W16_MFMA(u, xb, 0)                        \
  wide_load_w<(u) * 1024>(w[u], wvo, wn);
#define W16_STEP_LAST(u)                    \
  wide_wait_vm<LOADS - 1 - (u)>(w[u]);      \
  W16_MFMA(u, xb, skip)
  W16_DMA(0, 0)
  wide_load_w<0>(w[0], wvo, wsrc);
  wide_load_w<1024>(w[1], wvo, wsrc);
  wide_load_w<2048>(w[2], wvo, wsrc);
  wide_load_w<3072>(w[3], wvo, wsrc);
  wide_wait_barrier<LOADS>();
  for (int g = 0; g + 1 < Gq; ++g) {
    const unsigned char* xb = smem_w + (g & 1) * WIDE_BUFBYTES;
    const unsigned char* wn = wsrc + (size_t)min((g + 1) * LOADS, n_steps - LOADS) * 1024;
    W16_DMA(g + 1, (g + 1) & 1)
    W16_STEP_PREF(0) W16_STEP_PREF(1) W16_STEP_PREF(2) W16_STEP_PREF(3)
    wide_wait_barrier<LOADS>();
  }
  {
    const unsigned char* xb = smem_w + ((Gq - 1) & 1) * WIDE_BUFBYTES;
    W16_STEP_LAST(0) W16_STEP_LAST(1) W16_STEP_LAST(2) W16_STEP_LAST(3)
  }
#undef W16_STEP_PREF
#undef W16_STEP_LAST
#undef W16_MFMA
#undef W16_DMA
  if (!tile_ok) return;
  if (S < 0) {  // measurement only (variant 9SS07): no partial stores — what the quarter-sum write-out costs
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < NL; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[mt][r];
    if (t == 12345.678f) part[lane] = t;
    return;
  }
  // D[i = n][j = m]: register 4q + r of a lane is column 8q + 4hi + r of the tile for row j of the activation tile
#pragma unroll
  for (int mt = 0; mt < NL; ++mt) {
    if (j >= m_tile) continue;
    const int m = 32 * mt + j;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = tile * 32 + 8 * q + 4 * hi;
      if (n >= N) continue;
      float* dst = part + ((size_t)(4 * split + kq) * WIDE_MPAD + m) * N + n;
      *reinterpret_cast<float4*>(dst) = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
    }
  }
}

// quarters in the single-request kernel's order, splits in splitk_reduce_kernel's; one bf16 rounding (the NONE epilogue without bias)
__global__ __launch_bounds__(256) void wide16_reduce_kernel(const float* __restrict__ part, bf16_t* __restrict__ Y, int ldy, int m_tile, int N, int S) {
  const int m = blockIdx.y, n = (blockIdx.x * 256 + threadIdx.x) * 4;
  if ((m & 31) >= m_tile || n >= N) return;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int s = 0; s < S; ++s) {
    const float* p = part + ((size_t)(4 * s) * WIDE_MPAD + m) * N + n;
    const size_t st = (size_t)WIDE_MPAD * N;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + st),
                 c = *reinterpret_cast<const float4*>(p + 2 * st), d = *reinterpret_cast<const float4*>(p + 3 * st);
    const float4 q = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w);
    if (s == 0) acc = q;
    else { acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
  }
  *reinterpret_cast<uint2*>(Y + (size_t)m * ldy + n) = make_uint2(pack2(acc.x, acc.y), pack2(acc.z, acc.w));
}
