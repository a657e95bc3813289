// Cohort-8 GEMM (round 5): FIVE to EIGHT requests (activation tiles of 32 rows each, `m_tile` live rows per tile) share ONE pass over a
// W32-packed weight.   Y[256, N] = X[256, K] · W[N, K]^T, every weight byte streamed from HBM exactly once per launch.
//
// Why a third wide kernel.  The eight-row-block kernel of gemm_wide.h keeps a cohort row bit-identical to the single-request kernel by
// holding TWO accumulator sets per wave (`cur` of the running K-quarter, `tot` of the folded quarters): 2 x 16 x NL registers, which ends at
// NL = 4 activation tiles for 8 waves per CU.  A CU's ingest path (<= ~54 GB/s of W + X, tools/probe/cu_ingest_probe.hip) is what a cohort
// GEMM costs, in CU-time: per (row block x request tile) a workgroup of RB row blocks x NL tiles ingests (RB + NL) / (RB NL) units —
// 0.375 at 8 x 4 (and at every shape two accumulator sets allow for eight tiles), 0.25 at 8 x 8.  So the eight-request form gives the
// quarter fold up: ONE accumulator set per wave (16 x 8 = 128 registers), the split's K range walked in ascending k from the first
// k-step to the last.  Measured (profiles/r05_*): 1.35-1.39 x the time of the 8 x 4 kernel for twice the requests; the matrix pipe of
// the CUs it occupies is ~78 % busy (SQ_VALU_MFMA_BUSY_CYCLES = 32 x the MFMA count) next to ~50 of ~54 GB/s of ingest: M = 256 rows per
// weight pass is the CU's balance point — more requests per pass would be MFMA-bound.
//
// Arithmetic ("the c8 order"): an output element is ONE chain of v_mfma_f32_32x32x16_bf16 (fp8 activations: v_mfma_scale_f32_32x32x64_f8f6f4)
// accumulations over the split's k-steps in ascending order, starting from +0; split-K partials are added in ascending split order by
// splitk_reduce_kernel; the epilogues round where gemm_w32_kernel's do.  It differs from the single-request kernel's order (four K-quarters
// per split, folded ((q0 + q1) + q2) + q3) in fp32 rounding only.  What holds bit for bit: a request's rows do not depend on which tile it
// sits in, on how many tiles are live, or on m_tile (an MFMA output column depends on its own B column only) — a request of a cohort of
// 5..8 computes the same tokens whatever shares its weight pass, and the speculative and autoregressive forms of such a cohort agree
// (tests/test_c8_gpu.py).  Against the oracle the c8 order is held to the same float bar as every other kernel.
//
// Structure: a workgroup is 8 waves; wave w owns weight row block 8 blockIdx.x + w over the split's WHOLE K range; the eight waves walk k
// in lockstep through ONE staged X ring (LDS-DMA, LA = 2 groups of lookahead, 3 x 32 KiB + a zero buffer), wave w stages tile w's four 1 KiB pieces of
// every group.  Splits are cut on GROUP boundaries (64 k; fp8 activations 128 k), so every group is TL weight tiles + 4 activation pieces
// per wave and the memory pipeline is one steady state with hand-counted vmcnt from the first group to the last; the prefetches of the
// last two groups are stand-ins that stay on chip (the workgroup's first weight tile, the last activation group again): no weight byte
// comes from HBM twice.
#pragma once
#include "gemm_wide.h"

#define C8_NL 8
#define C8_MPAD (32 * C8_NL)
#define C8_BUFBYTES (C8_NL * 4096)
#ifndef C8_LA
#define C8_LA 2  // groups of lookahead (-DC8_LA=3: 12 KiB of W per wave + 96 KiB of X per workgroup in flight, all 160 KiB of LDS)
#endif
#ifndef C8_LW
#define C8_LW C8_LA  // groups of lookahead of the WEIGHT stream alone (registers, not LDS; round 6 experiment: -DC8_LW=3 / 4 at C8_LA = 2 keeps 12 / 16
#endif               // KiB of weights per wave in flight against the loaded HBM latency while the activation ring, which reads L2, stays at two groups)
#define C8_LDS_BYTES ((C8_LA + 2) * C8_BUFBYTES)  // 128 KiB (ring of LA + 1 buffers + the zero buffer): one workgroup per CU
static_assert(C8_LDS_BYTES <= 160 * 1024, "the cohort-8 ring does not fit a CU's 160 KiB of LDS");

// the fast kernel needs whole groups: K a multiple of the group, at least one group per split
static inline bool c8_fast_ok(int K, int S, int W8) {
  const int gk = W8 == 2 ? 128 : 64;
  return K % gk == 0 && K / gk >= S;
}

template <int EPI, int W8>
__global__ __launch_bounds__(512) void gemm_w32_c8_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                          const bf16_t* __restrict__ bias, void* __restrict__ Yv, int ldy,
                                                          const bf16_t* __restrict__ R, int ldr, int m_tile, int n_live, int N, int K, int S,
                                                          const float* __restrict__ wscale, RopeEpi re, int tiles,
                                                          const float* __restrict__ xscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  WGCLK_BEGIN();
  constexpr bool A8 = W8 == 2;
  constexpr int NL = C8_NL, LA = C8_LA, LW = C8_LW, NB = LA + 1;
  constexpr int KSTEP = A8 ? 64 : (W8 ? 32 : 16), LOADS = W8 ? 2 : 4, TPS = A8 ? 2 : 1, TL = LOADS * TPS;  // k per step, steps per group, tiles per step / group
  constexpr int GK = KSTEP * LOADS;        // k per group: 128 bytes of every activation row in all three forms
  constexpr int PPW = 4;                   // 1 KiB activation pieces a wave moves per group: the four pieces of tile `wave`
  // The wave's memory queue (in order).  Group h issues D(h + LA) at its top and, after step u, W(h + LW, u).  At step u of group g the ops
  // younger than W(g, u) are: the rest of the LW weight groups in flight (LW TL - TPS) and the LW activation groups issued since W(g, *) went
  // out (LW PPW) -> STEP_ALLOW; at the closing barrier D(g + 1) must have landed: younger than it are the weights of LA groups and LA - 1
  // activation groups -> BAR_ALLOW.  (LW = LA: QIN - TPS and QIN - PPW of the round-5 form, QIN = LA (TL + PPW).)
  constexpr int STEP_ALLOW = LW * TL - TPS + LW * PPW, BAR_ALLOW = LA * TL + (LA - 1) * PPW;
  static_assert((LA == 2 || LA == 3) && LW >= LA && LW <= 4, "the loop below is unrolled for LW = 2..4 weight groups in flight, LA = 2 or 3 ring groups");
  static_assert(STEP_ALLOW < 64, "vmcnt is a 6-bit counter");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, hi = lane >> 5;
  // Split-K launches: the workgroups of one split read the same [256, K / S] slice of X.  Workgroups go to the XCDs round-robin by linear id
  // (id % 8, MI355X_MICROARCH.md: observed, for speed only), so with the plain (blockIdx.x, blockIdx.y) mapping every XCD's L2 fetches the
  // WHOLE X (8 x X bytes over the fabric per launch: 1.47 x the algorithmic bytes of the o_proj / down launches at eight requests,
  // profiles/r05_pmc_fetch_size.json).  Remapped: a split's workgroups share 8 / S XCDs, each L2 holds only its splits' slices.
  int split = blockIdx.y, bx = blockIdx.x;
#ifndef VISPEC_C8_XCD_SWIZZLE
#define VISPEC_C8_XCD_SWIZZLE 1
#endif
  if (VISPEC_C8_XCD_SWIZZLE && (S == 2 || S == 4 || S == 8) && gridDim.x % (8 / S) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, r = lin >> 3, xps = 8 / S;
    split = xcd / xps;
    bx = r * xps + xcd % xps;
  }
  const int tile_raw = bx * 8 + wave;
  const bool tile_ok = tile_raw < tiles;
  const int tile = tile_ok ? tile_raw : 0;  // a ragged last workgroup streams tile 0 again and stores nothing
  const int KG = K / GK;
  const int g_lo = (int)((long)KG * split / S), g_hi = (int)((long)KG * (split + 1) / S);
  const int G = g_hi - g_lo;  // >= 1 (host: c8_fast_ok)
  f32x16 acc[NL];
#pragma unroll
  for (int mt = 0; mt < NL; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // ---- activation staging (LDS image and source-side swizzle of gemm_w32_wide_kernel): wave w moves tile w's [32 rows] x [128 B] per group
  // as 4 pieces of 8 rows x 128 B.  Rows past m_tile re-read the tile's last live row; tiles past n_live re-read tile 0 (never stored).
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_w;
  unsigned xoff[PPW], xdst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int mt = wave, pc = i;
    const int row = 32 * (mt < n_live ? mt : 0) + min(8 * pc + (lane >> 3), m_tile - 1), g = (lane & 7) ^ ((4 * pc + (lane >> 4)) & 7);
    xoff[i] = ((unsigned)row * (unsigned)ldx + (unsigned)g * 8u) * 2u;  // (bytes; A8: ldx is the row pitch in 2-byte units)
    xdst[i] = lds0 + (unsigned)(mt * 4 + pc) * 1024u;
  }
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(X) + (size_t)g_lo * 128;                                          // + 128 B per group
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(P) + ((size_t)tile * (K / KSTEP) * TPS + (size_t)g_lo * TL) * 1024;  // + TL KiB per group
  const unsigned wvo = lane * 16;
  // fragment reads: lane (j, hi) takes k-segment s of row j: bf16 step u -> s = 2u + hi ; fp8 tile c -> s = 4c + 2hi and s + 1
  const unsigned rrow = (unsigned)(j >> 3) * 1024u + (unsigned)(j & 7) * 128u, fsw = (unsigned)(j >> 1) & 7u;
  unsigned ro[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned sseg = A8 ? (unsigned)(4 * (t >> 1) + 2 * (t & 1) + hi) : (W8 ? (unsigned)(4 * (t >> 1) + 2 * hi + (t & 1)) : (unsigned)(2 * t + hi));
    ro[t] = rrow + ((sseg ^ fsw) << 4);
  }
  u32x4_t w[LW][TL];
  uint4 xf[LOADS][NL][W8 ? 2 : 1];
#define C8_DMA(grp, slot)                                                                                       \
  {                                                                                                             \
    const unsigned char* xs_ = xsrc + (size_t)(grp) * 128;                                                      \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) wide_dma16(xoff[i], xs_, xdst[i] + (unsigned)(slot) * C8_BUFBYTES); \
  }
#define C8_WLOAD_ALL(SL, wp)                                                                                    \
  {                                                                                                             \
    wide_load_w<0>(w[SL][0], wvo, wp);                                                                          \
    wide_load_w<1024>(w[SL][1], wvo, wp);                                                                       \
    if constexpr (TL > 2) {                                                                                     \
      wide_load_w<2048>(w[SL][TL > 2 ? 2 : 0], wvo, wp);                                                        \
      wide_load_w<3072>(w[SL][TL > 2 ? 3 : 0], wvo, wp);                                                        \
    }                                                                                                           \
  }
  // the activation fragments of step u (NL x 16 B per lane; fp8 weights: 2 x) from the group's ring slot
#define C8_XREAD(u, xb_)                                                                                        \
  {                                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                         \
      if constexpr (!W8) {                                                                                      \
        xf[u][mt][0] = *reinterpret_cast<const uint4*>((xb_) + mt * 4096 + ro[u]);                              \
      } else {                                                                                                  \
        xf[u][mt][0] = *reinterpret_cast<const uint4*>((xb_) + mt * 4096 + ro[(2 * (u)) & 3]);                  \
        xf[u][mt][W8 ? 1 : 0] = *reinterpret_cast<const uint4*>((xb_) + mt * 4096 + ro[(2 * (u) + 1) & 3]);     \
      }                                                                                                         \
    }                                                                                                           \
  }
  // one k-step (TPS weight tile registers) against the NL staged activation tiles
#define C8_MFMA(SL, u)                                                                                          \
  {                                                                                                             \
    if constexpr (A8) {                                                                                         \
      const u32x4_t &w0_ = w[SL][(2 * (u)) % TL], &w1_ = w[SL][(2 * (u) + 1) % TL];                             \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                         \
        acc[mt] = mfma_f8_64(make_uint4(w0_.x, w0_.y, w0_.z, w0_.w), make_uint4(w1_.x, w1_.y, w1_.z, w1_.w), xf[u][mt][0], xf[u][mt][W8 ? 1 : 0], acc[mt]); \
    } else if constexpr (!W8) {                                                                                 \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                         \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&w[SL][u]), as_bf16x8(xf[u][mt][0]), acc[mt], 0, 0, 0); \
    } else {                                                                                                    \
      uint4 a_lo, a_hi;                                                                                         \
      fp8x16_to_bf16(make_uint4(w[SL][u].x, w[SL][u].y, w[SL][u].z, w[SL][u].w), a_lo, a_hi);                   \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                       \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(xf[u][mt][0]), acc[mt], 0, 0, 0);           \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(xf[u][mt][W8 ? 1 : 0]), acc[mt], 0, 0, 0);  \
      }                                                                                                         \
    }                                                                                                           \
  }
  // step u of a group: wait until at most ALLOW younger operations are in flight (= this step's TPS tiles have landed), read the NEXT step's
  // fragments, run the step's MFMAs; REISSUE: the step's registers are loaded again from `wn` (the group LA ahead)
#define C8_STEP(SL, u, ALLOW, REISSUE)                                                                          \
  if constexpr ((u) < LOADS) {                                                                                  \
    if constexpr (TPS == 2) wide_wait_vm2<(ALLOW)>(w[SL][(2 * (u)) % TL], w[SL][(2 * (u) + 1) % TL]);           \
    else wide_wait_vm<(ALLOW)>(w[SL][(u) < LOADS ? (u) : 0]);                                                   \
    if constexpr ((u) + 1 < LOADS) C8_XREAD(((u) + 1 < LOADS ? (u) + 1 : 0), xb)                                \
    C8_MFMA(SL, (u) < LOADS ? (u) : 0)                                                                          \
    if constexpr (REISSUE) {                                                                                    \
      if constexpr (TPS == 2) {                                                                                 \
        wide_load_w<((2 * (u)) % TL) * 1024>(w[SL][(2 * (u)) % TL], wvo, wn);                                   \
        wide_load_w<((2 * (u) + 1) % TL) * 1024>(w[SL][(2 * (u) + 1) % TL], wvo, wn);                           \
      } else {                                                                                                  \
        wide_load_w<((u) < LOADS ? (u) : 0) * 1024>(w[SL][(u) < LOADS ? (u) : 0], wvo, wn);                     \
      }                                                                                                         \
    }                                                                                                           \
  }
  // Queue of a wave (oldest first) at the top of group g (LW = LA = 2):  W(g, *), D(g+1), W(g+1, *)  — then D(g+2) is issued; step u needs
  // W(g, TPS u ..): STEP_ALLOW younger ones may stay; after its MFMAs W(g+LW, TPS u ..) is issued; the closing barrier needs D(g+1) landed:
  // BAR_ALLOW younger ones may stay.  ONE code path from the first group to the last (branches around the unrolled bodies made hipcc move
  // the accumulators through scratch): the loop runs in pairs of groups; the prefetches of the last two groups are STAND-INS — the
  // activation pieces of the last group again (L2-resident) into the free ring slot, and the workgroup's first weight tile (one 1 KiB line
  // set for all eight waves: L2 hits) — and an odd split's extra group multiplies real (finite) weight bytes by the zero buffer.
#define C8_GROUP(SL)                                                                                            \
  {                                                                                                             \
    const bool live = gi < G;                                                                                   \
    const bool pf_real = gi + LW < G;                                                                           \
    const unsigned char* xb = live ? smem_w + rd * C8_BUFBYTES : smem_w + NB * C8_BUFBYTES;                     \
    const unsigned char* wn = wsrc + (pf_real ? (long)(gi + LW) * (TL * 1024) : wstand_off);                    \
    C8_XREAD(0, xb)                                                                                             \
    C8_DMA(min(gi + LA, G - 1), wr)                                                                             \
    C8_STEP(SL, 0, STEP_ALLOW, true) C8_STEP(SL, 1, STEP_ALLOW, true) C8_STEP(SL, 2, STEP_ALLOW, true) C8_STEP(SL, 3, STEP_ALLOW, true) \
    wide_wait_barrier<BAR_ALLOW>();                                                                             \
    ++gi;                                                                                                       \
    rd = rd + 1 == NB ? 0 : rd + 1;                                                                             \
    wr = wr + 1 == NB ? 0 : wr + 1;                                                                             \
  }
  {  // the zero buffer (slot NB): 32 KiB, 64 B per thread
    uint4* z = reinterpret_cast<uint4*>(smem_w + NB * C8_BUFBYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) z[threadIdx.x + 512 * i] = make_uint4(0, 0, 0, 0);
  }
  // stand-in weight source: tile 0 of the workgroup's first row block, every 1 KiB load at the same bytes (offsets cancelled)
  const long wstand_off = ((long)min(bx * 8, tiles - 1) - tile) * (K / KSTEP) * (TPS * 1024) - (long)g_lo * (TL * 1024);  // (relative to wsrc)
  __syncthreads();  // (the zero buffer is written; nothing else touches LDS before the first DMA lands)
  // ---- prologue: what the groups -LW .. -1 of the steady state would have issued, in their order: group h = i - LW issues D(h + LA) (if that
  // group exists) and then W(i, *):  LW = LA = 2: [D(0), W(0), D(1), W(1)];  LW = 3, LA = 2: [W(0), D(0), W(1), D(1), W(2)]
#define C8_PROLOGUE(i)                                                                                          \
  if constexpr ((i) < LW) {                                                                                     \
    if constexpr ((i) - LW + LA >= 0) C8_DMA(min(((i) - LW + LA >= 0 ? (i) - LW + LA : 0), G - 1), ((i) - LW + LA >= 0 ? (i) - LW + LA : 0)) \
    const unsigned char* wp_ = wsrc + (G > (i) ? (long)(i) * (TL * 1024) : wstand_off);                         \
    C8_WLOAD_ALL(((i) < LW ? (i) : 0), wp_)                                                                     \
  }
  C8_PROLOGUE(0) C8_PROLOGUE(1) C8_PROLOGUE(2) C8_PROLOGUE(3)
#undef C8_PROLOGUE
  wide_wait_barrier<BAR_ALLOW>();  // group 0's activations are staged
  int rd = 0, wr = LA, gi = 0;  // ring slot read by the current group / written by the prefetch (LA ahead, mod NB)
  while (gi < G) {  // (the weight registers cycle mod LW at compile time, the ring slots mod NB at run time)
    C8_GROUP(0)
    C8_GROUP(1)
    if constexpr (LW > 2) C8_GROUP((LW > 2 ? 2 : 0))
    if constexpr (LW > 3) C8_GROUP((LW > 3 ? 3 : 0))
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stand-in fetches of the last groups (their LDS-DMAs must not outlive the workgroup)
#undef C8_GROUP
#undef C8_STEP
#undef C8_MFMA
#undef C8_XREAD
#undef C8_WLOAD_ALL
#undef C8_DMA
  if (!tile_ok) return;
  wide_epilogue<EPI, W8, NL, C8_MPAD>(acc, tile, split, j, hi, m_tile, N, bias, Yv, ldy, R, ldr, wscale, re, xscale, n_live);
  WGCLK_END(10 + EPI, Yv);
}

// The same arithmetic for shapes the ring cannot walk (K not a multiple of a group, or fewer groups than splits: the tiny models of the test
// suites): one wave per (row block, split), operands fragment-shaped straight from global memory, the split cut on k-STEP boundaries.
template <int EPI, int W8>
__global__ __launch_bounds__(64) void gemm_w32_c8_small_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                               const bf16_t* __restrict__ bias, void* __restrict__ Yv, int ldy,
                                                               const bf16_t* __restrict__ R, int ldr, int m_tile, int n_live, int N, int K, int S,
                                                               const float* __restrict__ wscale, RopeEpi re, int tiles,
                                                               const float* __restrict__ xscale = nullptr) {
  constexpr bool A8 = W8 == 2;
  constexpr int NL = C8_NL, KSTEP = A8 ? 64 : (W8 ? 32 : 16), TPS = A8 ? 2 : 1;
  const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int KS = K / KSTEP;
  const int ks_lo = (int)((long)KS * split / S), ks_hi = (int)((long)KS * (split + 1) / S);
  f32x16 acc[NL];
#pragma unroll
  for (int mt = 0; mt < NL; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  const uint4* pl = reinterpret_cast<const uint4*>(P) + ((size_t)tile * KS + ks_lo) * TPS * 64 + lane;
  const size_t k0 = A8 ? (size_t)ks_lo * 32 + hi * 8 : (size_t)ks_lo * KSTEP + (W8 ? hi * 16 : hi * 8);  // (2-byte units)
  for (int s = ks_lo; s < ks_hi; ++s) {
    const uint4 av = *pl;
#pragma unroll
    for (int mt = 0; mt < NL; ++mt) {
      const bf16_t* px = X + (size_t)(32 * (mt < n_live ? mt : 0) + min(j, m_tile - 1)) * ldx + k0 + (size_t)(s - ks_lo) * (A8 ? 32 : KSTEP);
      const uint4 bv = *reinterpret_cast<const uint4*>(px);
      if (A8) {
        acc[mt] = mfma_f8_64(av, pl[64], bv, *reinterpret_cast<const uint4*>(px + 16), acc[mt]);
      } else if (!W8) {
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av), as_bf16x8(bv), acc[mt], 0, 0, 0);
      } else {
        const uint4 bv1 = *reinterpret_cast<const uint4*>(px + 8);
        uint4 a_lo, a_hi;
        fp8x16_to_bf16(av, a_lo, a_hi);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(bv), acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(bv1), acc[mt], 0, 0, 0);
      }
    }
    pl += 64 * TPS;
  }
  wide_epilogue<EPI, W8, NL, C8_MPAD>(acc, tile, split, j, hi, m_tile, N, bias, Yv, ldy, R, ldr, wscale, re, xscale, n_live);
}
