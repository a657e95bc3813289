// Wide-cohort GEMM: up to FOUR requests (activation tiles of 32 rows each, `m_tile` live rows per tile) share ONE pass over a
// W32-packed weight.   Y[128, N] = X[128, K] · W[N, K]^T, every weight byte streamed from HBM exactly once per launch.
//
// Why a new kernel (and not gemm_w32_kernel<…, MT = 4>): with four activation tiles per weight tile the skinny kernel's per-wave
// activation staging moves 4 bytes of X through ds_write/L1 per weight byte (measured 2.3 TB/s on three streams, tools/
// gemm_mt2_concurrency.py M=120) — the LDS store path (ds_write_b128 ≈ 79 B/clk/CU, MI355X_MICROARCH.md §LDS) and the TA, not HBM, set
// its pace.  Here a workgroup is 16 waves = 4 row blocks x 4 K-quarters: the four row-block waves of a K-quarter SHARE one staged
// X image per 64-k group (LDS, double-buffered, same padded layout as the skinny kernel), so the workgroup moves one byte of X per
// byte of W — the ratio of the single-request kernel — while each weight tile held in registers feeds four MFMAs.
//
// Bit-identity with the single-request kernel (what keeps "a cohort request == the same request alone" exact): wave (rb, kq)
// accumulates row block rb over exactly the k-steps wave kq of gemm_w32_kernel<1, …, NW = 4> would own for the same split (same
// ks_lo/ks_hi and w_lo/w_hi formulas), in ascending k order, one v_mfma_f32_32x32x16_bf16 per step with the same operands; the four
// quarter sums are then added as ((q0 + q1) + q2) + q3 — the order of the skinny kernel's LDS reduction — and the epilogues apply
// the same fp32 -> bf16 rounding points.
//
// Epilogues: as gemm_w32_kernel (NONE / RESIDUAL / SWIGLU / PARTIAL / ROPE).  Tile t of X/Y belongs to request t (`n_live` requests).
#pragma once
#include "kernels.h"

// LDS image of the staged activations (no padding: LDS-DMA writes lane-linear 1 KiB pieces):
//   K-quarter kq: + kq * 32 KiB ; buffer b: + b * 16 KiB ; activation tile mt: + mt * 4 KiB ; piece pc (8 rows x 64 k): + pc * 1 KiB ;
//   row r of the piece: + r * 128 B ; 16-byte slot c of the row holds k-segment  g = c ^ ((4 pc + (r >> 1)) & 7)  of that row.
// The XOR is applied on the SOURCE address of the DMA (the destination of a global_load_lds is fixed: base + lane x 16) and again by the
// fragment reads; with f(row j) = (j >> 1) & 7 the 16 lanes of every ds_read_b128 service group hit 16 different 16-byte bank slots.
#define WIDE_QBYTES (32 * 1024)
#define WIDE_BUFBYTES (16 * 1024)
#define WIDE_LDS_BYTES (4 * WIDE_QBYTES)               // 128 KiB: one workgroup per CU
#define WIDE_MPAD 128
#ifndef VISPEC_WIDE_DEADROW_REDIRECT  // A/B switch (VISPEC_HIPCC_FLAGS=-DVISPEC_WIDE_DEADROW_REDIRECT=0)
#define VISPEC_WIDE_DEADROW_REDIRECT 1
#endif

// Hand-counted memory pipeline.  hipcc's own s_waitcnt placement cannot express it: with an LDS-DMA in flight it waits vmcnt(0) at every use
// of an ordinary load (MI355X guide, "three .s-level traps"), and around a loop back-edge it falls back to vmcnt(0) for the weight
// registers as well — either way the whole HBM latency of the last weight tile lands on every 64-k group.  So every VMEM operation of the
// main loop is an asm statement the compiler does not see, and the waits are counted by hand (the queue is in order):
//   top of a steady-state group:  [w0 .. w(L-1)]                 (this group's tiles, issued during the previous group)
//   + NL activation pieces of the NEXT group by LDS-DMA:         [w0 .. w(L-1), D x NL]
//   step u waits vmcnt(L - 1 + NL) (tile u landed), runs its MFMAs, then re-issues its register for the next group: the queue
//   keeps that length; before the barrier vmcnt(L) retires the DMAs and leaves the L weight tiles in flight.
template <int N>
__device__ __forceinline__ void wide_wait_vm(u32x4_t& w) {  // "+v": nothing that consumes w may be scheduled above the wait
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait_vm2(u32x4_t& w0, u32x4_t& w1) {  // the same for a step that consumes two weight registers
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w0), "+v"(w1) : "i"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void wide_load_w(u32x4_t& w, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(w) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
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

// ---- epilogues (the arithmetic of gemm_w32_kernel's, on the sums held in registers) — shared by the two wide kernels.  D[i = n][j = m]:
// register 4q + r of a lane is column 8q + 4hi + r of the tile for row j of the activation tile.
template <int EPI, int W8, int NL, int MPAD = WIDE_MPAD>
__device__ __forceinline__ void wide_epilogue(f32x16 (&acc)[NL], int tile, int split, int j, int hi, int m_tile, int N, const bf16_t* __restrict__ bias,
                                              void* __restrict__ Yv, int ldy, const bf16_t* __restrict__ R, int ldr, const float* __restrict__ wscale,
                                              const RopeEpi& re, const float* __restrict__ xscale = nullptr, int n_live = NL) {
  constexpr bool A8 = W8 == 2;  // e4m3 activations: the accumulator also takes the row's activation scale xscale[m] (after the weight scale)
#pragma unroll
  for (int mt = 0; mt < NL; ++mt) {
    if (j >= m_tile || mt >= n_live) continue;  // (n_live < NL: the cohort-8 kernel always computes eight tiles)
    const int m = 32 * mt + j;
    if (EPI == EPI_ROPE) {
      if (re.rows[mt] && j >= re.rows[mt]) continue;  // (second tile of a 33..64-node tree: rows past the tree; 255 -> 0 live rows below)
      if (re.rows[mt] == 255) continue;
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
            float x1 = acc[mt][4 * qq + r], x2 = acc[mt][4 * (qq + 2) + r];
            if (W8) { x1 *= wscale[c1 + r]; x2 *= wscale[c2 + r]; }
            if (A8) { x1 *= xscale[m]; x2 *= xscale[m]; }
            if (bias) { x1 += bf2f(bias[c1 + r]); x2 += bf2f(bias[c2 + r]); }
            x1 = rdbf(x1);
            x2 = rdbf(x2);
            const float cs = bf2f(re.cosT[(size_t)pos * 128 + d + r]), sn = bf2f(re.sinT[(size_t)pos * 128 + d + r]);
            o1[r] = rdbf(rdbf(x1 * cs) + rdbf(-x2 * sn));
            o2[r] = rdbf(rdbf(x2 * cs) + rdbf(x1 * sn));
          }
          bf16_t* dst = (h < re.H) ? reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + c1
                                   : re.kc[mt] + ((size_t)(h - re.H) * re.s_max + kvrow) * 128 + d;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
          *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]));
        } else {  // v head: natural order, columns ncol + r and ncol + 16 + r
          float o1[4], o2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x1 = acc[mt][4 * qq + r], x2 = acc[mt][4 * (qq + 2) + r];
            if (W8) { x1 *= wscale[ncol + r]; x2 *= wscale[ncol + 16 + r]; }
            if (A8) { x1 *= xscale[m]; x2 *= xscale[m]; }
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
          float y = acc[mt][4 * qq + r], u = acc[mt][4 * (qq + 2) + r];
          if (W8) { y *= wscale[n + r]; u *= wscale[N + n + r]; }
          if (A8) { y *= xscale[m]; u *= xscale[m]; }
          if (bias) { y += bf2f(bias[n + r]); u += bf2f(bias[N + n + r]); }
          y = rdbf(y);
          u = rdbf(u);
          const float act = rdbf(y / (1.0f + __expf(-y)));
          o[r] = rdbf(act * u);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = tile * 32 + 8 * q + 4 * hi;
        if (n >= N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[mt][4 * q + r];
          if (W8) v[r] *= wscale[n + r];  // per-output-channel dequantisation scale on the fp32 accumulator
          if (A8) v[r] *= xscale[m];
        }
        if (EPI == EPI_PARTIAL) {
          float* part = reinterpret_cast<float*>(Yv) + ((size_t)split * MPAD + m) * N + n;
          *reinterpret_cast<float4*>(part) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float y = v[r];
            if (bias) y += bf2f(bias[n + r]);
            y = rdbf(y);
            if (EPI == EPI_RESIDUAL) y = rdbf(bf2f(R[(size_t)m * ldr + n + r]) + y);
            o[r] = y;
          }
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
        }
      }
    }
  }
}

// NL = live activation tiles (requests) — a template parameter: runtime guards around the MFMAs made hipcc branch around every
// one of them and keep the accumulators in scratch memory.
// DBG (measurement only, wrong results): 1 = no activation DMAs, 2 = no weight loads
// RB = weight row blocks per workgroup (4 K-quarters each: RB x 4 waves).  4: one byte of X per byte of W.  2: twice the workgroups for
// GEMMs whose row blocks cannot fill the chip otherwise (a CU sustains ~45-50 GB/s of X + W whatever the lookahead — tools/wide_bench.py —
// so an idle CU costs more than the doubled X traffic of the busy ones)
template <int EPI, int W8 /* 0 bf16, 1 e4m3 weights x bf16 activations, 2 e4m3 x e4m3 (X = codes, ldx in 2-byte units, xscale) */, int NL, int DBG = 0, int RB = 4>
__global__ __launch_bounds__(RB * 256) void gemm_w32_wide_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                             const bf16_t* __restrict__ bias, void* __restrict__ Yv, int ldy,
                                                             const bf16_t* __restrict__ R, int ldr, int m_tile, int N, int K, int S,
                                                             const float* __restrict__ wscale, RopeEpi re, int tiles,
                                                             const float* __restrict__ xscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  constexpr bool A8 = W8 == 2;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rb = wave % RB, kq = wave / RB;  // row block of the workgroup / K-quarter
  const int j = lane & 31, hi = lane >> 5;
  const int split = blockIdx.y;
  const int tile_raw = blockIdx.x * RB + rb;
  const bool tile_ok = tile_raw < tiles;
  const int tile = tile_ok ? tile_raw : 0;  // a ragged last workgroup streams tile 0 again and stores nothing
  constexpr int KSTEP = A8 ? 64 : (W8 ? 32 : 16);  // k per step: one 1 KiB weight tile — A8: one MFMA = 64 k = TWO weight tiles
  constexpr int TPS = A8 ? 2 : 1;                  // weight tiles per step
  constexpr int XB = A8 ? 64 : KSTEP * 2;          // bytes of an X row one step covers (a group is 128 bytes of every row in all three forms)
  constexpr int PPW = (4 * NL + RB - 1) / RB;  // 1 KiB activation pieces a wave moves per 64-k group (the quarter's RB waves share 4 NL;
                                               // RB = 3 with NL = 4: 18 slots for 16 pieces, the last piece is fetched three times)
  constexpr int LOADS = W8 ? 2 : 4;         // steps per group
  const int KS = K / KSTEP;
  const int ks_lo = (int)((long)KS * split / S), ks_hi = (int)((long)KS * (split + 1) / S);
  const int len = ks_hi - ks_lo;
  const int w_lo = ks_lo + (int)((long)len * kq / 4), w_hi = ks_lo + (int)((long)len * (kq + 1) / 4);
  const int n_steps = w_hi - w_lo;
  const int G = n_steps / LOADS;  // whole 64-k groups of this K-quarter (quarter-uniform)
  // A quarter whose step count is not a multiple of a group (down_proj: 43 steps) ends with a TAIL group: the LAST 64 k of the quarter,
  // fetched like any other group; its first `skip` steps were already accumulated by the group before, so they multiply by a ZERO
  // activation fragment (acc + W x 0 leaves every accumulator bit as it is: an accumulator that started at +0 is never -0).  The steps
  // therefore still enter the sums in ascending k, once each, and nothing waits on a fragment-shaped global load (the round-3a kernel
  // walked these steps one exposed load latency at a time: 31 -> see tools/wide_bench.py).
  const int rem = G > 0 ? n_steps - G * LOADS : 0;
  const int Gq = G + (rem ? 1 : 0), skip = rem ? LOADS - rem : 0;
  int Gmax = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = ks_lo + (int)((long)len * q / 4), b = ks_lo + (int)((long)len * (q + 1) / 4);
    const int gq = (b - a) / LOADS;
    Gmax = max(Gmax, gq + ((gq > 0 && (b - a) % LOADS) ? 1 : 0));
  }
  const uint4* pa = reinterpret_cast<const uint4*>(P) + (size_t)tile * KS * TPS * 64 + lane + (size_t)w_lo * TPS * 64;
  f32x16 acc[NL];
#pragma unroll
  for (int mt = 0; mt < NL; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // ---- staging of this K-quarter's activations: its 4 waves move [32 NL rows] x [64 k] per group as 4 NL pieces of 8 rows x 128 B (whole
  // lines), PPW = 4 NL / RB pieces per wave, by LDS-DMA (no staging registers, no ds_write pass).  Rows past m_tile of a tile hold a copy of
  // the last live row and only ever reach output rows that are not stored.
  unsigned char* xq = smem_w + kq * WIDE_QBYTES;
  const unsigned lds_q = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xq;
  unsigned xoff[PPW], xdst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = min(rb * PPW + i, 4 * NL - 1), mt = q >> 2, pc = q & 3;
    // rows past m_tile of a tile are dead (their output rows are never stored): their lanes re-read the tile's last live row — the same
    // 128-byte line the live lanes of that row fetch, so the piece costs the TA / L2 one line less per dead row (T = 30: 2 of 32)
    const int row = 32 * mt + (VISPEC_WIDE_DEADROW_REDIRECT ? min(8 * pc + (lane >> 3), m_tile - 1) : 8 * pc + (lane >> 3)),
              g = (lane & 7) ^ ((4 * pc + (lane >> 4)) & 7);
    xoff[i] = ((unsigned)row * (unsigned)ldx + (unsigned)g * 8u) * 2u;  // (bytes; A8: ldx is the row pitch in 2-byte units)
    xdst[i] = lds_q + (unsigned)(mt * 4 + pc) * 1024u;
  }
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(X) + (size_t)w_lo * XB;  // + 128 B per group
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(P) + ((size_t)tile * KS + w_lo) * TPS * 1024;  // + LOADS TPS KiB per group
  const unsigned wvo = lane * 16;
  // fragment reads: lane (j, hi) takes k-segment s of row j: bf16 step u -> s = 2u + hi ; fp8 tile c -> s = 4c + 2hi and s + 1
  const unsigned rrow = (unsigned)(j >> 3) * 1024u + (unsigned)(j & 7) * 128u, fsw = (unsigned)(j >> 1) & 7u;
  unsigned ro[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    // A8: step u = t >> 1 reads 16-byte slots 4 u + hi and 4 u + 2 + hi of the row chunk (its two weight tiles' k ranges for this lane)
    const unsigned sseg = A8 ? (unsigned)(4 * (t >> 1) + 2 * (t & 1) + hi) : (W8 ? (unsigned)(4 * (t >> 1) + 2 * hi + (t & 1)) : (unsigned)(2 * t + hi));
    ro[t] = rrow + ((sseg ^ fsw) << 4);
  }
  u32x4_t w[LOADS * TPS];
  constexpr int QW = DBG == 2 ? 0 : LOADS * TPS, QX = DBG == 1 ? 0 : PPW;  // weight loads / activation DMAs in flight per group and wave
#define WIDE_DMA(grp, buf)                                                                                      \
  if constexpr (DBG != 1) {                                                                                     \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i)                                                             \
        wide_dma16(xoff[i], xsrc + (size_t)min((grp) * LOADS, n_steps - LOADS) * XB, xdst[i] + (buf) * WIDE_BUFBYTES); \
  }
  // one k-step (one weight tile register) against the NL staged activation tiles
#define WIDE_MFMA(u, xb, SK)                                                                                    \
  if constexpr (A8) {                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                         \
      uint4 b0 = *reinterpret_cast<const uint4*>((xb) + mt * 4096 + ro[(2 * (u)) & 3]);                         \
      uint4 b1 = *reinterpret_cast<const uint4*>((xb) + mt * 4096 + ro[(2 * (u) + 1) & 3]);                     \
      {  /* (a mask, not a branch: the branchy form of the tail group spilled eight registers to scratch) */    \
        const unsigned km_ = (u) < (SK) ? 0u : 0xffffffffu;                                                     \
        b0.x &= km_; b0.y &= km_; b0.z &= km_; b0.w &= km_; b1.x &= km_; b1.y &= km_; b1.z &= km_; b1.w &= km_; \
      }                                                                                                         \
      acc[mt] = mfma_f8_64(make_uint4(w[(2 * (u)) % (LOADS * TPS)].x, w[(2 * (u)) % (LOADS * TPS)].y, w[(2 * (u)) % (LOADS * TPS)].z, w[(2 * (u)) % (LOADS * TPS)].w), \
                           make_uint4(w[(2 * (u) + 1) % (LOADS * TPS)].x, w[(2 * (u) + 1) % (LOADS * TPS)].y, w[(2 * (u) + 1) % (LOADS * TPS)].z, w[(2 * (u) + 1) % (LOADS * TPS)].w), \
                           b0, b1, acc[mt]);                                                                    \
    }                                                                                                           \
  } else if constexpr (!W8) {                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                         \
      uint4 bv = *reinterpret_cast<const uint4*>((xb) + mt * 4096 + ro[u]);                                     \
      if ((u) < (SK)) bv = make_uint4(0, 0, 0, 0);                                                              \
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&w[u]), as_bf16x8(bv), acc[mt], 0, 0, 0); \
    }                                                                                                           \
  } else {                                                                                                      \
    uint4 a_lo, a_hi;                                                                                           \
    fp8x16_to_bf16(make_uint4(w[u].x, w[u].y, w[u].z, w[u].w), a_lo, a_hi);                                     \
    _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                         \
      uint4 b0 = *reinterpret_cast<const uint4*>((xb) + mt * 4096 + ro[(2 * (u)) & 3]);                         \
      uint4 b1 = *reinterpret_cast<const uint4*>((xb) + mt * 4096 + ro[(2 * (u) + 1) & 3]);                     \
      if ((u) < (SK)) { b0 = make_uint4(0, 0, 0, 0); b1 = b0; }                                                 \
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(b0), acc[mt], 0, 0, 0);      \
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(b1), acc[mt], 0, 0, 0);      \
    }                                                                                                           \
  }
  // step u: its TPS weight tiles (registers TPS u ..) have landed -> its MFMAs -> the registers are re-issued for the next group.  In flight
  // before the wait: (LOADS - u) TPS tiles of this group + QX DMAs + u TPS re-issued ones = QW + QX; the oldest TPS must be done.
#define WIDE_STEP_PREF(u)                                                                                       \
  if constexpr ((u) < LOADS) {                                                                                            \
    if constexpr (QW > 0) {                                                                                     \
      if constexpr (TPS == 2) wide_wait_vm2<(QW > 0 ? QW - TPS : 0) + QX>(w[(TPS * (u)) % (LOADS * TPS)], w[(TPS * (u) + 1) % (LOADS * TPS)]); \
      else wide_wait_vm<(QW > 0 ? QW - 1 : 0) + QX>(w[(u) < LOADS ? (u) : 0]);                                  \
    }                                                                                                           \
    WIDE_MFMA((u) < LOADS ? (u) : 0, xb, 0)                                                                     \
    if constexpr (QW > 0) {                                                                                     \
      wide_load_w<((TPS * (u)) % (LOADS * TPS)) * 1024>(w[(TPS * (u)) % (LOADS * TPS)], wvo, wn);               \
      if constexpr (TPS == 2) wide_load_w<((TPS * (u) + 1) % (LOADS * TPS)) * 1024>(w[(TPS * (u) + 1) % (LOADS * TPS)], wvo, wn); \
    }                                                                                                           \
  }
#define WIDE_STEP_LAST(u)                                                                                       \
  if constexpr ((u) < LOADS) {                                                                                            \
    if constexpr (QW > 0) {                                                                                     \
      if constexpr (TPS == 2) wide_wait_vm2<((u) < LOADS ? (LOADS - 1 - (u)) * TPS : 0)>(w[(TPS * (u)) % (LOADS * TPS)], w[(TPS * (u) + 1) % (LOADS * TPS)]); \
      else wide_wait_vm<((u) < LOADS ? LOADS - 1 - (u) : 0)>(w[(u) < LOADS ? (u) : 0]);                         \
    }                                                                                                           \
    WIDE_MFMA((u) < LOADS ? (u) : 0, xb, skip)                                                                  \
  }
  if (G > 0) {
    WIDE_DMA(0, 0)
    wide_load_w<0>(w[0], wvo, wsrc);
    wide_load_w<1024>(w[1], wvo, wsrc);
    if constexpr (LOADS * TPS > 2) {
      wide_load_w<2048>(w[LOADS * TPS > 2 ? 2 : 0], wvo, wsrc);
      wide_load_w<3072>(w[LOADS * TPS > 2 ? 3 : 0], wvo, wsrc);
    }
    if constexpr (DBG == 2) wide_wait_vm<QX>(w[0]);  // (the first group's tiles stay in the registers for the whole run)
  }
  if (Gmax > 0) wide_wait_barrier<QW>();  // (a quarter without groups has nothing in flight: the count is harmless)
  // single-path loop body (an if / else-if / else around the two unrolled bodies made hipcc shuffle the accumulators through scratch):
  // steady-state groups, then the quarter's last group, then barrier-only rounds so that every wave of the workgroup executes the same
  // number of s_barrier whatever its quarter's group count (the quarters share nothing but the rendezvous itself)
  for (int g = 0; g + 1 < Gq; ++g) {  // group g + 1 is fetched while group g is on the matrix cores
    const unsigned char* xb = xq + (g & 1) * WIDE_BUFBYTES;
    const unsigned char* wn = wsrc + (size_t)min((g + 1) * LOADS, n_steps - LOADS) * TPS * 1024;
    WIDE_DMA(g + 1, (g + 1) & 1)  // that buffer was last read before the previous barrier
    WIDE_STEP_PREF(0) WIDE_STEP_PREF(1) WIDE_STEP_PREF(2) WIDE_STEP_PREF(3)
    wide_wait_barrier<QW>();
  }
  if (Gq > 0) {  // the quarter's last group (the tail group when there is one): nothing left to fetch
    const unsigned char* xb = xq + ((Gq - 1) & 1) * WIDE_BUFBYTES;
    WIDE_STEP_LAST(0) WIDE_STEP_LAST(1) WIDE_STEP_LAST(2) WIDE_STEP_LAST(3)
    wide_wait_barrier<0>();
  }
  for (int e = Gq; e < Gmax; ++e) wide_wait_barrier<0>();
#undef WIDE_STEP_PREF
#undef WIDE_STEP_LAST
#undef WIDE_MFMA
#undef WIDE_DMA
  if (G == 0) {  // a K range shorter than one group (tiny models): fragment-shaped X loads straight from global
    const uint4* pl = pa;
    const size_t k0 = A8 ? (size_t)w_lo * 32 + hi * 8 : (size_t)w_lo * KSTEP + (W8 ? hi * 16 : hi * 8);  // (2-byte units)
    for (int s = 0; s < n_steps; ++s) {
      const uint4 av = *pl;
#pragma unroll
      for (int mt = 0; mt < NL; ++mt) {
          const bf16_t* px = X + (size_t)(j < m_tile ? 32 * mt + j : 0) * ldx + k0 + (size_t)s * (A8 ? 32 : KSTEP);
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
  }
  // ---- K-quarter reduction through LDS (aliases the staging area): quarters 1..3 publish two activation tiles per pass, the
  // quarter-0 wave of each row block adds them in the fixed order ((q0 + q1) + q2) + q3 and keeps the result in registers.
  // Layout [q-1][rb][tile of the pass][4-register group][lane] x 16 B: conflict-free 16-byte LDS accesses.
  float4* red = reinterpret_cast<float4*>(smem_w);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();  // pass 0: the main loop's / leftover reads of the staging area are over; pass 1: pass 0's reads are over
    if (kq > 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          if (2 * pass + t >= NL) continue;
          const f32x16& a = acc[2 * pass + t < NL ? 2 * pass + t : 0];
          red[((((kq - 1) * RB + rb) * 2 + t) * 4 + r4) * 64 + lane] = make_float4(a[4 * r4], a[4 * r4 + 1], a[4 * r4 + 2], a[4 * r4 + 3]);
        }
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int q = 1; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            if (2 * pass + t >= NL) continue;
            const float4 v = red[((((q - 1) * RB + rb) * 2 + t) * 4 + r4) * 64 + lane];
            f32x16& a = acc[2 * pass + t < NL ? 2 * pass + t : 0];
            a[4 * r4] += v.x; a[4 * r4 + 1] += v.y; a[4 * r4 + 2] += v.z; a[4 * r4 + 3] += v.w;
          }
    }
  }
  if (kq != 0 || !tile_ok) return;
  wide_epilogue<EPI, W8, NL>(acc, tile, split, j, hi, m_tile, N, bias, Yv, ldy, R, ldr, wscale, re, xscale);
}

// =====================================================================================================================================
// Wide-8 (round 4): EIGHT weight row blocks per workgroup — the form for a GPU that several lanes keep busy, where a launch costs CU-time in
// proportion to the bytes its workgroups ingest (W + X per workgroup, <= ~55 GB/s per CU whatever the source: tools/probe/cu_ingest_probe.hip).
// The 4 x 4 kernel above re-reads the whole [32 NL, K] activation block once per FOUR row blocks (bf16: one byte of X per byte of W; fp8
// weights: two); here a workgroup is 8 waves, wave w owns row block 8 blockIdx.x + w over the split's WHOLE K range, and the eight waves
// walk k in lockstep through ONE staged X ring (LDS-DMA, LA groups of lookahead) — half the X traffic per weight byte.
//
// Bit-identity with the single-request kernel is kept by accumulating the K range quarter by quarter: `cur` runs over the k-steps of
// quarter q (the steps wave q of gemm_w32_kernel<1, ..., NW = 4> owns, ascending, one MFMA per step with the same operands) and is folded
// into `tot` at the quarter's end — tot = q0, then (q0 + q1), ((q0 + q1) + q2), (((q0 + q1) + q2) + q3): the order of the skinny kernel's
// LDS reduction.  Two accumulator sets (2 x 16 NL registers) are what limits a wave to one row block at 8 waves per CU.
//
// Groups of 64 k are aligned to the QUARTER starts (a quarter of n steps = n / LOADS whole groups + a tail group that re-reads the last
// 64 k with its already-used steps pointed at a zero buffer — acc + W x 0 changes no bit), so every group is LOADS weight tiles and
// 4 NL activation pieces and the memory pipeline is one steady state with hand-counted vmcnt from the first group to the last:
//   per group and wave: [PPW activation DMAs of group g + LA] then, after step u's MFMAs, [weight tile u of group g + LA]
//   in flight at the top of a group: LA x (LOADS + PPW) operations; step u waits vmcnt(LA (LOADS + PPW) - 1), the barrier that publishes
//   group g + 1's activations waits vmcnt(LA (LOADS + PPW) - PPW).  The last LA groups fetch stand-ins (the last group again).
// Needs every quarter of the split to hold at least one whole group (the host falls back to the 4 x 4 kernel otherwise: tiny models).
#define WIDE8_BUFBYTES (16 * 1024)
#ifndef VISPEC_WIDE8_A8_LA
#define VISPEC_WIDE8_A8_LA 2  // (experiments: -DVISPEC_WIDE8_A8_LA=3 = 12 KiB of W per wave in flight for the W8A8 form)
#endif
template <int W8> constexpr int wide8_la() { return W8 == 1 ? 4 : (W8 == 2 ? VISPEC_WIDE8_A8_LA : 2); }  // groups of lookahead (8 KiB of W per wave in flight)
template <int W8> constexpr int wide8_lds_bytes() { return (wide8_la<W8>() + 2) * WIDE8_BUFBYTES; }  // ring of LA + 1 buffers + the zero buffer

template <int EPI, int W8, int NL>
__global__ __launch_bounds__(512) void gemm_w32_wide8_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                             const bf16_t* __restrict__ bias, void* __restrict__ Yv, int ldy,
                                                             const bf16_t* __restrict__ R, int ldr, int m_tile, int N, int K, int S,
                                                             const float* __restrict__ wscale, RopeEpi re, int tiles,
                                                             const float* __restrict__ xscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  // W8 = 2 (fp8 activations too, `X` = e4m3 codes, ldx in 2-byte units, xscale = per-row scales): a k-step is 64 k = TWO weight tiles against
  // 64 activation bytes per row on the f8f6f4 MFMA; a group (one 16 KiB ring buffer) is 128 k = 2 steps = 4 tile loads per wave, like bf16
  constexpr bool A8 = W8 == 2;
  constexpr int LA = wide8_la<W8>(), NB = LA + 1;
  constexpr int KSTEP = A8 ? 64 : (W8 ? 32 : 16), LOADS = W8 ? 2 : 4, TPS = A8 ? 2 : 1, TL = LOADS * TPS;  // k per step, steps per group, tiles per step / group
  constexpr int XB = A8 ? 64 : KSTEP * 2;  // activation bytes per step and row
  constexpr int PPW = (4 * NL + 7) / 8;  // 1 KiB activation pieces a wave moves per group (NL = 3: 16 slots for 12 pieces, the last one fetched five times)
  constexpr int QIN = LA * (TL + PPW);   // memory operations in flight per wave in steady state
  constexpr int XAHEAD = W8 ? 1 : 2;       // k-steps whose activation fragments are read from LDS ahead of their MFMAs (8 NL... 2 NL x 16 B per lane either way)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, hi = lane >> 5;
  const int split = blockIdx.y;
  const int tile_raw = blockIdx.x * 8 + wave;
  const bool tile_ok = tile_raw < tiles;
  const int tile = tile_ok ? tile_raw : 0;  // a ragged last workgroup streams tile 0 again and stores nothing
  const int KS = K / KSTEP;
  const int ks_lo = (int)((long)KS * split / S), ks_hi = (int)((long)KS * (split + 1) / S);
  const int len = ks_hi - ks_lo;
  // total number of groups of the split (all quarters)
  int Gtot = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = ks_lo + (int)((long)len * q / 4), b = ks_lo + (int)((long)len * (q + 1) / 4);
    Gtot += (b - a + LOADS - 1) / LOADS;
  }
  f32x16 cur[NL], tot[NL];
#pragma unroll
  for (int mt = 0; mt < NL; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { cur[mt][r] = 0.f; tot[mt][r] = 0.f; }

  // ---- activation staging: the workgroup's 8 waves move [32 NL rows] x [64 k] per group as 4 NL pieces of 8 rows x 128 B (LDS image and
  // source-side swizzle of the 4 x 4 kernel above)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_w;
  unsigned xoff[PPW], xdst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = min(wave * PPW + i, 4 * NL - 1), mt = q >> 2, pc = q & 3;
    const int row = 32 * mt + min(8 * pc + (lane >> 3), m_tile - 1), g = (lane & 7) ^ ((4 * pc + (lane >> 4)) & 7);
    xoff[i] = ((unsigned)row * (unsigned)ldx + (unsigned)g * 8u) * 2u;
    xdst[i] = lds0 + (unsigned)(mt * 4 + pc) * 1024u;
  }
  {  // the zero buffer (slot NB): 16 KiB, 32 B per thread
    uint4* z = reinterpret_cast<uint4*>(smem_w + NB * WIDE8_BUFBYTES);
    z[threadIdx.x] = make_uint4(0, 0, 0, 0);
    z[threadIdx.x + 512] = make_uint4(0, 0, 0, 0);
  }
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(X);                                     // + step * XB bytes
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(P) + (size_t)tile * KS * (TPS * 1024);   // + step * TPS KiB
  const unsigned wvo = lane * 16;
  const unsigned rrow = (unsigned)(j >> 3) * 1024u + (unsigned)(j & 7) * 128u, fsw = (unsigned)(j >> 1) & 7u;
  unsigned ro[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned sseg = A8 ? (unsigned)(4 * (t >> 1) + 2 * (t & 1) + hi) : (W8 ? (unsigned)(4 * (t >> 1) + 2 * hi + (t & 1)) : (unsigned)(2 * t + hi));
    ro[t] = rrow + ((sseg ^ fsw) << 4);
  }
  // ---- group iterators (wave-uniform scalars): (quarter, index in quarter) -> first k-step, zeroed leading steps, last-of-quarter
  struct It { int q, i, a, n; };
  auto it_begin = [&]() { It t; t.q = 0; t.i = 0; t.a = ks_lo; t.n = ks_lo + (int)((long)len * 1 / 4) - ks_lo; return t; };
  auto it_step0 = [&](const It& t) { return t.i < t.n / LOADS ? t.a + t.i * LOADS : t.a + t.n - LOADS; };
  auto it_skip = [&](const It& t) { return t.i < t.n / LOADS ? 0 : LOADS - t.n % LOADS; };
  auto it_last = [&](const It& t) { return t.i == (t.n + LOADS - 1) / LOADS - 1; };
  auto it_next = [&](It& t) {  // never moves past the split's last group (stand-in fetches repeat it)
    if (t.i + 1 < (t.n + LOADS - 1) / LOADS) { ++t.i; return; }
    if (t.q == 3) return;
    ++t.q; t.i = 0; t.a += t.n;
    t.n = ks_lo + (int)((long)len * (t.q + 1) / 4) - t.a;
  };
  u32x4_t w[LA][TL];
#define W8_DMA(step0, slot)                                                                                   \
  {                                                                                                             \
    const unsigned char* xs_ = xsrc + (size_t)(step0) * XB;                                                     \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) wide_dma16(xoff[i], xs_, xdst[i] + (unsigned)(slot) * WIDE8_BUFBYTES); \
  }
  // The activation fragments of a group are read from LDS AHEAD of the MFMAs that consume them (two k-steps = 8 NL... 2 x NL x 16 B per lane in
  // flight): the whole group is staged before its barrier, so the reads need not wait for the weight tile — with two waves per SIMD an
  // LDS round trip in front of every MFMA quartet is what kept the matrix pipe at 44 % in the first form of this kernel (fp8: 29 GB/s
  // per CU instead of the ~55 the CU can ingest).  A step below `sk` reads the zero buffer instead of the ring slot.
  uint4 xf[LOADS][NL][W8 ? 2 : 1];
#define W8_XREAD(u, xb_live, sk)                                                                              \
  {                                                                                                             \
    const unsigned char* xb_ = (u) < (sk) ? smem_w + NB * WIDE8_BUFBYTES : (xb_live);                           \
    _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                         \
      if constexpr (!W8) {                                                                                      \
        xf[u][mt][0] = *reinterpret_cast<const uint4*>(xb_ + mt * 4096 + ro[u]);                                \
      } else {                                                                                                  \
        xf[u][mt][0] = *reinterpret_cast<const uint4*>(xb_ + mt * 4096 + ro[(2 * (u)) & 3]);                    \
        xf[u][mt][W8 ? 1 : 0] = *reinterpret_cast<const uint4*>(xb_ + mt * 4096 + ro[(2 * (u) + 1) & 3]);       \
      }                                                                                                         \
    }                                                                                                           \
  }
  // one k-step (one weight tile register) against the NL staged activation tiles
#define W8_MFMA(SL, u)                                                                                        \
  {                                                                                                             \
    if constexpr (A8) {                                                                                         \
      const u32x4_t &w0_ = w[SL][(2 * (u)) % TL], &w1_ = w[SL][(2 * (u) + 1) % TL];                             \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                         \
        cur[mt] = mfma_f8_64(make_uint4(w0_.x, w0_.y, w0_.z, w0_.w), make_uint4(w1_.x, w1_.y, w1_.z, w1_.w), xf[u][mt][0], xf[u][mt][W8 ? 1 : 0], cur[mt]); \
    } else if constexpr (!W8) {                                                                                        \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                         \
        cur[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&w[SL][u]), as_bf16x8(xf[u][mt][0]), cur[mt], 0, 0, 0); \
    } else {                                                                                                    \
      uint4 a_lo, a_hi;                                                                                         \
      fp8x16_to_bf16(make_uint4(w[SL][u].x, w[SL][u].y, w[SL][u].z, w[SL][u].w), a_lo, a_hi);                   \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) {                                                       \
        cur[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(xf[u][mt][0]), cur[mt], 0, 0, 0);           \
        cur[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(xf[u][mt][W8 ? 1 : 0]), cur[mt], 0, 0, 0);  \
      }                                                                                                         \
    }                                                                                                           \
  }
  // step u: its weight tile has landed -> (the fragment reads of the step after next go out) -> its MFMAs -> its register is re-issued
#define W8_STEP(SL, u)                                                                                        \
  if constexpr ((u) < LOADS) {                                                                                  \
    if constexpr (TPS == 2) wide_wait_vm2<QIN - TPS>(w[SL][(2 * (u)) % TL], w[SL][(2 * (u) + 1) % TL]);         \
    else wide_wait_vm<QIN - 1>(w[SL][(u) < LOADS ? (u) : 0]);                                                   \
    if constexpr ((u) + XAHEAD < LOADS) W8_XREAD(((u) + XAHEAD < LOADS ? (u) + XAHEAD : 0), xb, skip)           \
    W8_MFMA(SL, (u) < LOADS ? (u) : 0)                                                                          \
    if constexpr (TPS == 2) {                                                                                   \
      wide_load_w<((2 * (u)) % TL) * 1024>(w[SL][(2 * (u)) % TL], wvo, wn);                                     \
      wide_load_w<((2 * (u) + 1) % TL) * 1024>(w[SL][(2 * (u) + 1) % TL], wvo, wn);                             \
    } else {                                                                                                    \
      wide_load_w<((u) < LOADS ? (u) : 0) * 1024>(w[SL][(u) < LOADS ? (u) : 0], wvo, wn);                       \
    }                                                                                                           \
  }
  // one group: SL = its weight-register slot (g % LA, compile time), `c` = its iterator, `pf` = the iterator LA groups ahead
#define W8_GROUP(SL)                                                                                          \
  {                                                                                                             \
    const bool live = g + (SL) < Gtot;                                                                          \
    const int skip = live ? it_skip(c) : LOADS;                                                                 \
    const int s0p = it_step0(pf);                                                                               \
    const unsigned char* wn = wsrc + (size_t)s0p * (TPS * 1024);                                                \
    const unsigned char* xb = smem_w + rd * WIDE8_BUFBYTES;                                                     \
    W8_XREAD(0, xb, skip)                                                                                       \
    if constexpr (XAHEAD > 1) W8_XREAD((XAHEAD > 1 ? 1 : 0), xb, skip)                                          \
    W8_DMA(s0p, wr)                                                                                             \
    W8_STEP(SL, 0) W8_STEP(SL, 1) W8_STEP(SL, 2) W8_STEP(SL, 3)                                                 \
    if (live && it_last(c)) { /* the quarter is complete: fold it (uniform branch, three or four times per launch) */ \
      if (c.q == 0) {                                                                                           \
        _Pragma("unroll") for (int mt = 0; mt < NL; ++mt) tot[mt] = cur[mt];                                    \
      } else {                                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                       \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) tot[mt][r] += cur[mt][r];                              \
      }                                                                                                         \
      _Pragma("unroll") for (int mt = 0; mt < NL; ++mt)                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) cur[mt][r] = 0.f;                                        \
    }                                                                                                           \
    wide_wait_barrier<QIN - PPW>();                                                                             \
    it_next(c); it_next(pf);                                                                                    \
    rd = rd + 1 == NB ? 0 : rd + 1;                                                                             \
    wr = wr + 1 == NB ? 0 : wr + 1;                                                                             \
  }
  // ---- prologue: groups 0 .. LA-1 in the steady-state order [D(0), W(0,*), D(1), W(1,*), ...]
  It c = it_begin(), pf = it_begin();
  __syncthreads();  // (the zero buffer is written; nothing else touches LDS before the first DMA lands)
#pragma unroll
  for (int p = 0; p < LA; ++p) {
    const int s0 = it_step0(pf);
    const unsigned char* wp = wsrc + (size_t)s0 * (TPS * 1024);
    W8_DMA(s0, p)
    wide_load_w<0>(w[p][0], wvo, wp);
    wide_load_w<1024>(w[p][1], wvo, wp);
    if constexpr (TL > 2) {
      wide_load_w<2048>(w[p][TL > 2 ? 2 : 0], wvo, wp);
      wide_load_w<3072>(w[p][TL > 2 ? 3 : 0], wvo, wp);
    }
    it_next(pf);
  }
  wide_wait_barrier<QIN - PPW>();  // group 0's activations are staged
  int rd = 0, wr = LA;             // ring slot read by the current group / written by the prefetch (LA ahead, mod NB)
  for (int g = 0; g < Gtot; g += LA) {
    W8_GROUP(0)
    W8_GROUP(1)
    if constexpr (LA > 2) W8_GROUP(LA > 2 ? 2 : 0)
    if constexpr (LA > 3) W8_GROUP(LA > 3 ? 3 : 0)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stand-in fetches of the last groups (their LDS-DMAs must not outlive the workgroup)
#undef W8_GROUP
#undef W8_STEP
#undef W8_MFMA
#undef W8_XREAD
#undef W8_DMA
  if (!tile_ok) return;
  wide_epilogue<EPI, W8, NL>(tot, tile, split, j, hi, m_tile, N, bias, Yv, ldy, R, ldr, wscale, re, xscale);
}
