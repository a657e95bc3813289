// Wide-cohort GEMM: up to FOUR requests (activation tiles of 32 rows each, `m_tile` live rows per tile) share ONE pass over a
// W32-packed weight.   Y[128, N] = X[128, K] · W[N, K]^T, every weight byte streamed from HBM exactly once per launch.
//
// Why a new kernel (and not gemm_w32_kernel<…, MT = 4>): with four activation tiles per weight tile the skinny kernel's per-wave
// activation staging moves 4 bytes of X through ds_write/L1 per weight byte (measured 2.3 TB/s on three streams, tools/
// gemm_mt2_concurrency.py M=120).  Here a workgroup covers 4 weight row blocks x 4 K-quarters and the waves of a K-quarter SHARE one
// staged X image per 32-k sub-group (LDS, filled by LDS-DMA), so the workgroup moves one byte of X per byte of W — the ratio of the
// single-request kernel — while each weight tile held in registers feeds four MFMAs.
//
// Round-3 form (the first form had 16 waves, one row block each, 64-k groups double-buffered, one group of lookahead on both streams —
// and tools/wide_bench.py showed it LATENCY-bound: 2.6 us per 64-k group where the weight loads alone or the activation DMAs alone
// take 1.7 us and the MFMAs 0.85 us; a wave's weight tile was re-requested only one group before its use): 8 waves, each owning TWO
// row blocks of one K-quarter (two weight streams per wave, each activation fragment read from LDS feeds both), so that the register
// file (2 waves per SIMD: 256 VGPRs) holds a weight ring of THREE sub-groups per stream, and the X image is a ring of four 32-k
// sub-group buffers per K-quarter (three sub-groups of DMA lookahead in the same 128 KiB of LDS).
//
// Bit-identity with the single-request kernel (what keeps "a cohort request == the same request alone" exact): a (row block, K-quarter)
// chain accumulates exactly the k-steps wave kq of gemm_w32_kernel<1, …, NW = 4> would own for the same split (same ks_lo/ks_hi and
// w_lo/w_hi formulas), in ascending k order, one v_mfma_f32_32x32x16_bf16 per step with the same operands; the four quarter sums are
// then added as ((q0 + q1) + q2) + q3 — the order of the skinny kernel's LDS reduction — and the epilogues apply the same
// fp32 -> bf16 rounding points.
//
// Epilogues: as gemm_w32_kernel (NONE / RESIDUAL / SWIGLU / PARTIAL / ROPE).  Tile t of X/Y belongs to request t (NL requests).
#pragma once
#include "kernels.h"

// LDS image of the staged activations (no padding: LDS-DMA writes lane-linear 1 KiB pieces):
//   K-quarter kq: + kq * 32 KiB ; ring buffer b (sub-group h -> b = h & 3): + b * 8 KiB ; activation tile mt: + mt * 2 KiB ;
//   row j of the tile: + j * 64 B ; 16-byte slot c of the row holds k-segment  g = c ^ ((j >> 2) & 3)  of the sub-group's 32 k.
// A DMA piece is 16 rows x 64 B (lane l: row l >> 2, slot l & 3).  The XOR is applied on the SOURCE address of the DMA (the
// destination of a global_load_lds is fixed: base + lane x 16) and again by the fragment reads: the 16 lanes of every ds_read_b128
// service group then hit 16 different 16-byte bank slots.
#define WIDE_THREADS 512
#define WIDE_QBYTES (32 * 1024)
#define WIDE_BUFBYTES (8 * 1024)
#define WIDE_LDS_BYTES (4 * WIDE_QBYTES)               // 128 KiB: one workgroup per CU
#define WIDE_MPAD 128

// Hand-counted memory pipeline.  hipcc's own s_waitcnt placement cannot express it: with an LDS-DMA in flight it waits vmcnt(0) at every
// use of an ordinary load (MI355X guide, "three .s-level traps"), and around a loop back-edge it falls back to vmcnt(0) for the weight
// registers as well.  So every VMEM operation of the main loop is an asm statement the compiler does not see, and the waits are counted
// by hand (the queue retires in order).  With n = DMAs and L = weight loads a wave issues per sub-group, the queue at the top of a
// steady-state sub-group h, after its DMAs for h+3 went out, is
//     w(h) x L | D(h+1) x n, w(h+1) x L | D(h+2) x n, w(h+2) x L | D(h+3) x n            (+ the w(h+3) re-issued slot by slot)
//   slot s waits vmcnt(3n + 3L - 1): tile s of sub-group h has landed; its MFMAs run; its register is re-issued for sub-group h+3;
//   the sub-group ends with vmcnt(3L + 2n): this wave's pieces of X(h+1) have landed -> barrier -> every wave's have.
// The last three sub-groups of a K-quarter have nothing left to fetch and fetch stand-ins (L2 hits) so that the counts never change.
template <int N>
__device__ __forceinline__ void wide_wait_vm(u32x4_t& w) {  // "+v": nothing that consumes w may be scheduled above the wait
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w) : "i"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void wide_load_w(u32x4_t& w, unsigned voff, const unsigned char* sbase) {
  // ("+v": the load lands in the register the tile it replaces was in — a fresh "=v" value may be given a second register while the old
  //  one is still an MFMA operand, and the ring does not have registers to spare)
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "+v"(w) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
__device__ __forceinline__ void wide_dma16(unsigned voff, const unsigned char* sbase, unsigned lds_dst) {
  unsigned keep;  // M0 (the DMA's LDS base) is compiler-reserved: save, set, use and restore it inside ONE statement
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait_barrier() {  // own DMAs landed (N younger loads may stay in flight), own LDS reads drained, rendezvous
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "i"(N) : "memory");
}

struct WideArgs {  // what the epilogue needs besides the sums
  const bf16_t* bias; void* Yv; int ldy; const bf16_t* R; int ldr; int m_tile, N; const float* wscale;
};

// Epilogue of one (row block, activation tile) sum held in registers — the arithmetic of gemm_w32_kernel's.  D[i = n][j = m]: register
// 4q + r of a lane is column 8q + 4hi + r of the tile for row j of the activation tile.
template <int EPI, bool W8>
__device__ __forceinline__ void wide_store_tile(const f32x16& a, int mt, int tile, int split, int j, int hi, const WideArgs& g, const RopeEpi& re) {
  const bf16_t* bias = g.bias;
  const float* wscale = g.wscale;
  const int N = g.N, m = 32 * mt + j;
  if (EPI == EPI_ROPE) {
    const PosSpec& ps_ = re.ps[mt];
    const int kvrow = (ps_.kv_base ? *ps_.kv_base : 0) + ps_.kv_add + j;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int ncol = tile * 32 + 8 * qq + 4 * hi;  // packed column of a[0]
      const int h = ncol >> 7, t4 = (ncol & 127) >> 5, c = ncol & 31;
      if (h < re.H + re.H_kv) {
        const int d = 16 * t4 + c, c1 = h * 128 + d, c2 = c1 + 64;  // natural columns of the pair
        const int pos = (ps_.base ? *ps_.base : 0) + (ps_.base2 ? *ps_.base2 : 0) + ps_.add + (ps_.off ? ps_.off[j] : (ps_.row ? j : 0));
        float o1[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x1 = a[4 * qq + r], x2 = a[4 * (qq + 2) + r];
          if (W8) { x1 *= wscale[c1 + r]; x2 *= wscale[c2 + r]; }
          if (bias) { x1 += bf2f(bias[c1 + r]); x2 += bf2f(bias[c2 + r]); }
          x1 = rdbf(x1);
          x2 = rdbf(x2);
          const float cs = bf2f(re.cosT[(size_t)pos * 128 + d + r]), sn = bf2f(re.sinT[(size_t)pos * 128 + d + r]);
          o1[r] = rdbf(rdbf(x1 * cs) + rdbf(-x2 * sn));
          o2[r] = rdbf(rdbf(x2 * cs) + rdbf(x1 * sn));
        }
        bf16_t* dst = (h < re.H) ? reinterpret_cast<bf16_t*>(g.Yv) + (size_t)m * g.ldy + c1
                                 : re.kc[mt] + ((size_t)(h - re.H) * re.s_max + kvrow) * 128 + d;
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
        *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]));
      } else {  // v head: natural order, columns ncol + r and ncol + 16 + r
        float o1[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x1 = a[4 * qq + r], x2 = a[4 * (qq + 2) + r];
          if (W8) { x1 *= wscale[ncol + r]; x2 *= wscale[ncol + 16 + r]; }
          if (bias) { x1 += bf2f(bias[ncol + r]); x2 += bf2f(bias[ncol + 16 + r]); }
          o1[r] = rdbf(x1);
          o2[r] = rdbf(x2);
        }
        bf16_t* dst = re.vc[mt] + ((size_t)(h - re.H - re.H_kv) * re.s_max + kvrow) * 128 + (ncol & 127);
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
        *reinterpret_cast<uint2*>(dst + 16) = make_uint2(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]));
      }
    }
  } else if (EPI == EPI_SWIGLU) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int n = tile * 16 + 8 * qq + 4 * hi;  // output column; gate row n, up row N + n of the natural weight
      if (n >= N) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = a[4 * qq + r], u = a[4 * (qq + 2) + r];
        if (W8) { y *= wscale[n + r]; u *= wscale[N + n + r]; }
        if (bias) { y += bf2f(bias[n + r]); u += bf2f(bias[N + n + r]); }
        y = rdbf(y);
        u = rdbf(u);
        const float act = rdbf(y / (1.0f + __expf(-y)));
        o[r] = rdbf(act * u);
      }
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.Yv) + (size_t)m * g.ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = tile * 32 + 8 * q + 4 * hi;
      if (n >= N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = a[4 * q + r];
        if (W8) v[r] *= wscale[n + r];  // per-output-channel dequantisation scale on the fp32 accumulator
      }
      if (EPI == EPI_PARTIAL) {
        float* part = reinterpret_cast<float*>(g.Yv) + ((size_t)split * WIDE_MPAD + m) * N + n;
        *reinterpret_cast<float4*>(part) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = v[r];
          if (bias) y += bf2f(bias[n + r]);
          y = rdbf(y);
          if (EPI == EPI_RESIDUAL) y = rdbf(bf2f(g.R[(size_t)m * g.ldr + n + r]) + y);
          o[r] = y;
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.Yv) + (size_t)m * g.ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
      }
    }
  }
}

// NL = live activation tiles (requests) — a template parameter: runtime guards around the MFMAs made hipcc branch around every
// one of them and keep the accumulators in scratch memory.
// DBG (measurement only, wrong results): 1 = no activation DMAs, 2 = no weight loads
template <int EPI, bool W8, int NL, int DBG = 0>
__global__ __launch_bounds__(WIDE_THREADS) void gemm_w32_wide_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                                     const bf16_t* __restrict__ bias, void* __restrict__ Yv, int ldy,
                                                                     const bf16_t* __restrict__ R, int ldr, int m_tile, int N, int K, int S,
                                                                     const float* __restrict__ wscale, RopeEpi re, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rbp = wave & 1, kq = wave >> 1;  // row-block pair of the workgroup / K-quarter
  const int j = lane & 31, hi = lane >> 5;
  const int split = blockIdx.y;
  const int tile_raw0 = blockIdx.x * 4 + 2 * rbp, tile_raw1 = tile_raw0 + 1;
  const bool ok0 = tile_raw0 < tiles, ok1 = tile_raw1 < tiles;
  const int tile0 = ok0 ? tile_raw0 : 0, tile1 = ok1 ? tile_raw1 : 0;  // a ragged last workgroup streams tile 0 again and stores nothing
  constexpr int KSTEP = W8 ? 32 : 16;       // k per 1 KiB weight tile
  constexpr int LS = W8 ? 1 : 2;            // weight tiles (k-steps) per 32-k sub-group and row block
  constexpr int L = 2 * LS;                 // weight loads a wave issues per sub-group (two row blocks)
  constexpr int QW = DBG == 2 ? 0 : L, QX = DBG == 1 ? 0 : NL;  // loads / DMAs per sub-group and wave that really enter the queue
  const int KS = K / KSTEP;
  const int ks_lo = (int)((long)KS * split / S), ks_hi = (int)((long)KS * (split + 1) / S);
  const int len = ks_hi - ks_lo;
  const int w_lo = ks_lo + (int)((long)len * kq / 4), w_hi = ks_lo + (int)((long)len * (kq + 1) / 4);
  const int n_steps = w_hi - w_lo;
  const int G = n_steps / LS;       // 32-k sub-groups of this K-quarter (quarter-uniform)
  const bool piped = G >= 3;        // the pipeline needs three sub-groups to prime; shorter quarters take the plain path below
  int NB = 0;                       // s_barrier every wave of the workgroup executes, whatever its quarter's sub-group count
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = ks_lo + (int)((long)len * q / 4), b = ks_lo + (int)((long)len * (q + 1) / 4);
    const int gq = (b - a) / LS;
    NB = max(NB, gq >= 3 ? (gq + 2) / 3 * 3 + 2 : 0);
  }
  f32x16 acc0[NL], acc1[NL];
#pragma unroll
  for (int mt = 0; mt < NL; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[mt][r] = 0.f; acc1[mt][r] = 0.f; }

  // ---- staging of this K-quarter's activations: its 2 waves move [32 NL rows] x [32 k] per sub-group as 2 NL pieces of 16 rows x 64 B,
  // NL pieces per wave, by LDS-DMA (no staging registers, no ds_write pass).  Rows past m_tile of a tile are fetched too (the
  // workspaces hold 32 rows per tile) and only ever reach output rows that are not stored.
  unsigned char* xq = smem_w + kq * WIDE_QBYTES;
  const unsigned lds_q = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xq;
  unsigned xoff[NL], xdst[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int q = rbp * NL + i, mt = q >> 1, half = q & 1;
    const int r = lane >> 2, row = 32 * mt + 16 * half + r, g = (lane & 3) ^ ((r >> 2) & 3);
    xoff[i] = ((unsigned)row * (unsigned)ldx + (unsigned)g * 8u) * 2u;
    xdst[i] = lds_q + (unsigned)q * 1024u;
  }
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(X + (size_t)w_lo * KSTEP);  // + 64 B per sub-group
  const unsigned char* wsrc0 = reinterpret_cast<const unsigned char*>(P) + ((size_t)tile0 * KS + w_lo) * 1024;  // + LS KiB per sub-group
  const unsigned char* wsrc1 = reinterpret_cast<const unsigned char*>(P) + ((size_t)tile1 * KS + w_lo) * 1024;
  const unsigned wvo = lane * 16;
  // fragment reads: lane (j, hi) takes k-segment s of row j: bf16 step u -> s = 2u + hi ; fp8 (one 32-k tile) -> s = 2hi and 2hi + 1
  const unsigned rrow = (unsigned)j * 64u, fsw = (unsigned)(j >> 2) & 3u;
  unsigned ro[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const unsigned sseg = W8 ? (unsigned)(2 * hi + t) : (unsigned)(2 * t + hi);
    ro[t] = rrow + ((sseg ^ fsw) << 4);
  }
  // weight ring: slot (sub-group mod 3) x L tiles; tile index inside a sub-group: 2u + r (k-step u, row block r of the pair).
  // ONE loop body serves every sub-group (an if / else-if chain of peeled tail bodies made hipcc keep the accumulators in scratch):
  // the loop runs over G3 = G rounded up to a multiple of three sub-groups, always with the steady-state wait counts, and what has
  // nothing left to fetch fetches something harmless instead — the DMA re-reads the quarter's last sub-group into the ring buffer
  // nobody will read any more, the weight register re-loads 1 KiB of X (L2-resident) — while a sub-group past G skips its MFMAs.
  u32x4_t w0[L], w1[L], w2[L];
  const int G3 = (G + 2) / 3 * 3;
  const unsigned char* wdummy = reinterpret_cast<const unsigned char*>(X);
#define WIDE_DMA(sg)                                                                                            \
  if constexpr (DBG != 1) {                                                                                     \
    const int sgc_ = min((sg), G - 1);                                                                          \
    _Pragma("unroll") for (int i = 0; i < NL; ++i)                                                              \
        wide_dma16(xoff[i], xsrc + (size_t)sgc_ * 64, xdst[i] + (unsigned)((sg) & 3) * WIDE_BUFBYTES);          \
  }
  // all L weight tiles of sub-group `sg` into ring slot WR (prologue; G >= 3 there)
#define WIDE_LOAD_ALL(WR, sg)                                                                                   \
  if constexpr (DBG != 2) {                                                                                     \
    const unsigned char* a0_ = wsrc0 + (size_t)(sg) * (LS * 1024);                                              \
    const unsigned char* a1_ = wsrc1 + (size_t)(sg) * (LS * 1024);                                              \
    wide_load_w<0>(WR[0], wvo, a0_);                                                                            \
    wide_load_w<0>(WR[1], wvo, a1_);                                                                            \
    if constexpr (LS == 2) {                                                                                    \
      wide_load_w<1024>(WR[LS == 2 ? 2 : 0], wvo, a0_);                                                         \
      wide_load_w<1024>(WR[LS == 2 ? 3 : 0], wvo, a1_);                                                         \
    }                                                                                                           \
  }
  // one (k-step u, row block r) slot of a sub-group: wait for its tile, its NL MFMAs (fp8: 2 NL), re-issue the register for sub-group + 3
#define WIDE_SLOT(WR, u, r, ACC, wnext)                                                                         \
  {                                                                                                             \
    constexpr int si_ = 2 * (u) + (r);                                                                          \
    if constexpr (QW > 0) wide_wait_vm<3 * QX + 3 * QW - 1>(WR[si_]);                                           \
    if (live_) {                                                                                                \
      if constexpr (!W8) {                                                                                      \
        _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                       \
            ACC[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&WR[si_]), as_bf16x8(xf[mt]), ACC[mt], 0, 0, 0); \
      } else {                                                                                                  \
        uint4 a_lo, a_hi;                                                                                       \
        fp8x16_to_bf16(make_uint4(WR[si_].x, WR[si_].y, WR[si_].z, WR[si_].w), a_lo, a_hi);                     \
        _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                     \
          ACC[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(xf[mt]), ACC[mt], 0, 0, 0); \
          ACC[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(xg[mt]), ACC[mt], 0, 0, 0); \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
    if constexpr (QW > 0) wide_load_w<(u) * 1024>(WR[si_], wvo, wnext);                                         \
  }
  // the fragments of k-step u of the sub-group in ring buffer xb: read once, used by both row blocks
#define WIDE_FRAGS(u)                                                                                           \
  if (live_) {                                                                                                  \
    _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                         \
      xf[mt] = *reinterpret_cast<const uint4*>(xb + mt * 2048 + ro[W8 ? 0 : (u)]);                              \
      if constexpr (W8) xg[mt] = *reinterpret_cast<const uint4*>(xb + mt * 2048 + ro[1]);                       \
    }                                                                                                           \
  }
#define WIDE_SUB(h_, WR)                                                                                        \
  {                                                                                                             \
    const int hh_ = (h_);                                                                                       \
    const bool live_ = hh_ < G, more_ = hh_ + 3 < G;                                                            \
    const unsigned char* xb = xq + (hh_ & 3) * WIDE_BUFBYTES;                                                   \
    const unsigned char* wn0_ = more_ ? wsrc0 + (size_t)(hh_ + 3) * (LS * 1024) : wdummy;                       \
    const unsigned char* wn1_ = more_ ? wsrc1 + (size_t)(hh_ + 3) * (LS * 1024) : wdummy;                       \
    WIDE_DMA(hh_ + 3)  /* into the buffer that was last read before the previous barrier */                     \
    uint4 xf[NL], xg[W8 ? NL : 1];                                                                              \
    WIDE_FRAGS(0)                                                                                               \
    WIDE_SLOT(WR, 0, 0, acc0, wn0_)                                                                             \
    WIDE_SLOT(WR, 0, 1, acc1, wn1_)                                                                             \
    if constexpr (LS == 2) {                                                                                    \
      WIDE_FRAGS(1)                                                                                             \
      WIDE_SLOT(WR, (LS == 2 ? 1 : 0), 0, acc0, wn0_)                                                           \
      WIDE_SLOT(WR, (LS == 2 ? 1 : 0), 1, acc1, wn1_)                                                           \
    }                                                                                                           \
    wide_wait_barrier<3 * QW + 2 * QX>();                                                                       \
  }
  int nb_done = 0;
  if (piped) {
    WIDE_DMA(0) WIDE_LOAD_ALL(w0, 0)
    WIDE_DMA(1) WIDE_LOAD_ALL(w1, 1)
    WIDE_DMA(2) WIDE_LOAD_ALL(w2, 2)
    wide_wait_barrier<3 * QW + 2 * QX>();  // X(0) of every wave of the quarter is in its buffer
    for (int h = 0; h < G3; h += 3) {
      WIDE_SUB(h, w0) WIDE_SUB(h + 1, w1) WIDE_SUB(h + 2, w2)
    }
    wide_wait_barrier<0>();  // the stand-in fetches of the tail have landed before the reduction re-uses the staging area
    nb_done = G3 + 2;
  }
  for (int e = nb_done; e < NB; ++e) wide_wait_barrier<0>();  // the quarters share nothing but the rendezvous itself
#undef WIDE_SUB
#undef WIDE_FRAGS
#undef WIDE_SLOT
#undef WIDE_LOAD_ALL
#undef WIDE_DMA
  {  // leftover steps of this wave's K range (less than a sub-group; everything when the quarter is too short for the pipeline):
     // fragment-shaped X loads straight from global
    const int s0 = piped ? G * LS : 0;
    const uint4* pl0 = reinterpret_cast<const uint4*>(P) + ((size_t)tile0 * KS + w_lo + s0) * 64 + lane;
    const uint4* pl1 = reinterpret_cast<const uint4*>(P) + ((size_t)tile1 * KS + w_lo + s0) * 64 + lane;
    const size_t k0 = (size_t)(w_lo + s0) * KSTEP + (W8 ? hi * 16 : hi * 8);
    for (int s = s0; s < n_steps; ++s) {
      const uint4 av0 = *pl0, av1 = *pl1;
#pragma unroll
      for (int mt = 0; mt < NL; ++mt) {
        const bf16_t* px = X + (size_t)(j < m_tile ? 32 * mt + j : 0) * ldx + k0 + (size_t)(s - s0) * KSTEP;
        const uint4 bv = *reinterpret_cast<const uint4*>(px);
        if (!W8) {
          acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av0), as_bf16x8(bv), acc0[mt], 0, 0, 0);
          acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av1), as_bf16x8(bv), acc1[mt], 0, 0, 0);
        } else {
          const uint4 bv1 = *reinterpret_cast<const uint4*>(px + 8);
          uint4 a_lo, a_hi;
          fp8x16_to_bf16(av0, a_lo, a_hi);
          acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(bv), acc0[mt], 0, 0, 0);
          acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(bv1), acc0[mt], 0, 0, 0);
          fp8x16_to_bf16(av1, a_lo, a_hi);
          acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(bv), acc1[mt], 0, 0, 0);
          acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(bv1), acc1[mt], 0, 0, 0);
        }
      }
      pl0 += 64;
      pl1 += 64;
    }
  }
  // ---- K-quarter reduction through LDS (aliases the staging area): quarters 1..3 publish one row block's NL sums per pass, the
  // quarter-0 wave of each row-block pair adds them in the fixed order ((q0 + q1) + q2) + q3 and keeps the result in registers.
  // Layout [q-1][rbp][tile][4-register group][lane] x 16 B: conflict-free 16-byte LDS accesses (6 waves x NL x 4 KiB <= 96 KiB).
  float4* red = reinterpret_cast<float4*>(smem_w);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();  // pass 0: the main loop's / leftover reads of the staging area are over; pass 1: pass 0's reads are over
    if (kq > 0) {
#pragma unroll
      for (int t = 0; t < NL; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x16& a = pass == 0 ? acc0[t] : acc1[t];
          red[((((kq - 1) * 2 + rbp) * NL + t) * 4 + r4) * 64 + lane] = make_float4(a[4 * r4], a[4 * r4 + 1], a[4 * r4 + 2], a[4 * r4 + 3]);
        }
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int q = 1; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NL; ++t)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = red[((((q - 1) * 2 + rbp) * NL + t) * 4 + r4) * 64 + lane];
            f32x16& a = pass == 0 ? acc0[t] : acc1[t];
            a[4 * r4] += v.x; a[4 * r4 + 1] += v.y; a[4 * r4 + 2] += v.z; a[4 * r4 + 3] += v.w;
          }
    }
  }
  if (kq != 0 || j >= m_tile) return;
  const WideArgs ga{bias, Yv, ldy, R, ldr, m_tile, N, wscale};
#pragma unroll
  for (int mt = 0; mt < NL; ++mt) {
    if (ok0) wide_store_tile<EPI, W8>(acc0[mt], mt, tile0, split, j, hi, ga, re);
    if (ok1) wide_store_tile<EPI, W8>(acc1[mt], mt, tile1, split, j, hi, ga, re);
  }
}
