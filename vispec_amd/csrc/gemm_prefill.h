// GEMM for the draft PREFILL (cnets_ours.py:603-661 ImgAdaptor K/V projection over N_img image rows; :879-975 fc(cat(emb, img_fc(cat(h, g))))
// and q|k|v over the L_c compressed rows): M = hundreds to thousands of rows against the same W32-packed weights the decode rounds
// stream.  This half of the path is MFMA-bound (2·2·N_img·D² + 2·(2·2D² + 3D²)·L_c flops ≈ 0.3 TFLOP at L = 2704), not
// HBM-bound: walking it in 32-row passes of the skinny kernel re-streamed 67–100 MB of weights ≈ 85 times per request.
//
// Tiling: workgroup = 4 waves = 128 weight rows x 128 activation rows.  Each wave owns one 32-row W32 block (its 1 KiB tiles go
// straight from HBM/L2 into registers as MFMA A operands, no LDS) and four 32-row activation tiles (B operands) that all four waves share
// through one LDS image per 64-k group (same padded layout as the skinny kernel, double-buffered, next group in registers while the
// current one is on the matrix cores): 16 MFMA 32x32x16 per wave per group against 4 KiB of weights and 16 KiB of shared activations.
// The activation rows are GATHERED while they are staged — row m of the operand is src[idx[m]] — and the operand may be the
// concatenation of two row sources along K (k < k_split from the first, the rest from the second, which may be one broadcast row):
// that is cat(emb, img_fc(...)) / cat(h, g) and the image-row selection of the reference without materialising either.
// Epilogues: PLAIN (bias, bf16 store), KV (adaptor: natural column order scattered into a [2][H][cap][128] cache), ROPE_KV (the draft's
// k|v columns of the rope-ordered q|k|v weight: rotary at pos[m] + append at cache row kv_row0 + m; see RopeEpi in kernels.h).
#pragma once
#include "kernels.h"

struct BigA {
  const bf16_t* src0 = nullptr; const int* idx0 = nullptr; int ld0 = 0;  // row m, k < k_split : src0[(idx0 ? idx0[m] : m) * ld0 + k]
  const bf16_t* src1 = nullptr; const int* idx1 = nullptr; int ld1 = 0;  //        k >= k_split: src1[(bcast1 ? 0 : idx1 ? idx1[m] : m) * ld1 + k - k_split]
  int k_split = 0;   // multiple of 64; >= K when there is no second source
  int bcast1 = 0;
};
struct BigEpi {
  const bf16_t* bias = nullptr;
  bf16_t* Y = nullptr; int ldy = 0;                       // PLAIN
  bf16_t* kc = nullptr; bf16_t* vc = nullptr;             // KV / ROPE_KV: caches [H][cap][128]
  int cap = 0, H = 0, kv_row0 = 0, D = 0;                 // D = H * 128 (columns of k resp. v)
  const bf16_t* cosT = nullptr; const bf16_t* sinT = nullptr; const int* pos = nullptr;  // ROPE_KV
};
enum { BIG_PLAIN = 0, BIG_KV = 1, BIG_ROPE_KV = 2 };

#define BIG_XTILE (4 * XS_STEP)                  // one 32-row activation tile of a 64-k group
#define BIG_LDS_BYTES (2 * 4 * BIG_XTILE)        // 2 buffers x 4 tiles

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w32_big_kernel(BigA a, int M, const bf16_t* __restrict__ P, int K, int n_tile0, int n_tiles,
                                                           BigEpi e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, hi = lane >> 5;
  const int m_blk = blockIdx.y * 128;
  const int tile_rel = blockIdx.x * 4 + wave;              // this wave's weight row block (relative to n_tile0)
  const bool tile_ok = tile_rel < n_tiles;
  const int tile = n_tile0 + (tile_ok ? tile_rel : 0);
  const int KS = K >> 4, G = K >> 6;
  const uint4* pa = reinterpret_cast<const uint4*>(P) + (size_t)tile * KS * 64 + lane;
  f32x16 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  // staging map: thread -> 16-B segment seg of the 128-B row segment of rows srow + 32 i (one row per activation tile i)
  const int seg = threadIdx.x & 7, srow = threadIdx.x >> 3;
  const bf16_t* p0[4];
  const bf16_t* p1[4];
  int woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_blk + 32 * i + srow;
    const int mc = m < M ? m : 0;  // rows past the end read row 0, never stored
    p0[i] = a.src0 + (size_t)(a.idx0 ? a.idx0[mc] : mc) * a.ld0 + seg * 8;
    p1[i] = a.src1 ? a.src1 + (size_t)(a.bcast1 ? 0 : (a.idx1 ? a.idx1[mc] : mc)) * a.ld1 + seg * 8 : p0[i];
    woff[i] = i * BIG_XTILE + (seg >> 1) * XS_STEP + (seg & 1) * XS_HALF + srow * 16;
  }
  const int roff = hi * XS_HALF + j * 16;
  uint4 x0, x1, x2, x3, w0, w1, w2, w3, n0, n1, n2, n3;
#define BIG_LOADX(g)                                                                                          \
  {                                                                                                           \
    const int k_ = (g) * 64;                                                                                  \
    if (k_ < a.k_split) {                                                                                     \
      x0 = *reinterpret_cast<const uint4*>(p0[0] + k_); x1 = *reinterpret_cast<const uint4*>(p0[1] + k_);     \
      x2 = *reinterpret_cast<const uint4*>(p0[2] + k_); x3 = *reinterpret_cast<const uint4*>(p0[3] + k_);     \
    } else {                                                                                                  \
      const int k2_ = k_ - a.k_split;                                                                         \
      x0 = *reinterpret_cast<const uint4*>(p1[0] + k2_); x1 = *reinterpret_cast<const uint4*>(p1[1] + k2_);   \
      x2 = *reinterpret_cast<const uint4*>(p1[2] + k2_); x3 = *reinterpret_cast<const uint4*>(p1[3] + k2_);   \
    }                                                                                                         \
  }
#define BIG_LOADW(g, A0, A1, A2, A3)                                                                          \
  {                                                                                                           \
    const uint4* q_ = pa + (size_t)(g) * 4 * 64;                                                              \
    A0 = q_[0]; A1 = q_[64]; A2 = q_[128]; A3 = q_[192];                                                      \
  }
#define BIG_WRITEX(buf)                                                                                       \
  {                                                                                                           \
    unsigned char* xb_ = smem_b + (buf) * (4 * BIG_XTILE);                                                    \
    *reinterpret_cast<uint4*>(xb_ + woff[0]) = x0; *reinterpret_cast<uint4*>(xb_ + woff[1]) = x1;             \
    *reinterpret_cast<uint4*>(xb_ + woff[2]) = x2; *reinterpret_cast<uint4*>(xb_ + woff[3]) = x3;             \
  }
  BIG_LOADX(0)
  BIG_LOADW(0, w0, w1, w2, w3)
  BIG_WRITEX(0)
  __syncthreads();
  for (int g = 0; g < G; ++g) {
    const int gn = g + 1 < G ? g + 1 : g;  // the last iteration re-reads its own group: unconditional loads keep the pipeline simple
    BIG_LOADX(gn)
    BIG_LOADW(gn, n0, n1, n2, n3)
    const unsigned char* xb = smem_b + (g & 1) * (4 * BIG_XTILE);
#define BIG_STEP(u, WV)                                                                                       \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                          \
    const uint4 bv = *reinterpret_cast<const uint4*>(xb + mt * BIG_XTILE + (u) * XS_STEP + roff);             \
    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(WV), as_bf16x8(bv), acc[mt], 0, 0, 0);        \
  }
    BIG_STEP(0, w0) BIG_STEP(1, w1) BIG_STEP(2, w2) BIG_STEP(3, w3)
#undef BIG_STEP
    BIG_WRITEX((g + 1) & 1)  // the other buffer was last read before the previous barrier
    __syncthreads();
    w0 = n0; w1 = n1; w2 = n2; w3 = n3;
  }
#undef BIG_LOADX
#undef BIG_LOADW
#undef BIG_WRITEX
  if (!tile_ok) return;
  // epilogue straight from the accumulators: D[i = n][j = m], a lane holds 4 groups of 4 consecutive n for its m
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m_blk + 32 * mt + j;
    if (m >= M) continue;
    if (EPI == BIG_ROPE_KV) {
      const int ncol0 = tile * 32;  // packed column of the tile (rope order inside q / k heads, natural in v)
      const int h = ncol0 >> 7, t4 = (ncol0 & 127) >> 5;
      const int kvrow = e.kv_row0 + m;
      if (h < 2 * e.H) {  // a k head (the q tiles are not handed to this kernel): rotate_half pairs (d, d + 64) sit in one lane
        const int pos = e.pos[m];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int d = 16 * t4 + 8 * qq + 4 * hi, c1 = h * 128 + d, c2 = c1 + 64;
          float o1[4], o2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v1 = acc[mt][4 * qq + r], v2 = acc[mt][4 * (qq + 2) + r];
            if (e.bias) { v1 += bf2f(e.bias[c1 + r]); v2 += bf2f(e.bias[c2 + r]); }
            v1 = rdbf(v1);
            v2 = rdbf(v2);
            const float cs = bf2f(e.cosT[(size_t)pos * 128 + d + r]), sn = bf2f(e.sinT[(size_t)pos * 128 + d + r]);
            o1[r] = rdbf(rdbf(v1 * cs) + rdbf(-v2 * sn));
            o2[r] = rdbf(rdbf(v2 * cs) + rdbf(v1 * sn));
          }
          bf16_t* dst = e.kc + ((size_t)(h - e.H) * e.cap + kvrow) * 128 + d;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
          *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]));
        }
      } else {  // v head: natural order
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int n = ncol0 + 8 * qq + 4 * hi;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[mt][4 * qq + r];
            if (e.bias) v += bf2f(e.bias[n + r]);
            o[r] = rdbf(v);
          }
          bf16_t* dst = e.vc + ((size_t)(h - 2 * e.H) * e.cap + kvrow) * 128 + (n & 127);
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
        }
      }
    } else {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int n = tile * 32 + 8 * qq + 4 * hi;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mt][4 * qq + r];
          if (e.bias) v += bf2f(e.bias[n + r]);
          o[r] = rdbf(v);
        }
        const uint2 st = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
        if (EPI == BIG_PLAIN) {
          *reinterpret_cast<uint2*>(e.Y + (size_t)m * e.ldy + n) = st;
        } else {  // BIG_KV: columns [0, D) are K, [D, 2D) are V, 128 per head
          const int isv = n >= e.D, nn = isv ? n - e.D : n;
          bf16_t* dst = (isv ? e.vc : e.kc) + ((size_t)(nn >> 7) * e.cap + e.kv_row0 + m) * 128 + (nn & 127);
          *reinterpret_cast<uint2*>(dst) = st;
        }
      }
    }
  }
}
