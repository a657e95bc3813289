// Device kernels of libvispec_hip (gfx950 / CDNA4 only: wave64, MFMA bf16, 160 KiB LDS).
// Numerics contract (matches oracle/vispec_oracle.py in bf16 mode): bf16 storage, fp32 accumulation,
// round-to-nearest-even to bf16 wherever the reference's torch-bf16 graph materialises a tensor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wgclock.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define NEG_INF (-__builtin_huge_valf())

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16 round-to-nearest-even: gfx950 has it in hardware (v_cvt_pk_bf16_f32, two values per instruction); the casts below
// compile to it.  (The bit-twiddling form costs 4-5 VALU instructions per value and sat in every epilogue and attention chunk.)
typedef __attribute__((ext_vector_type(2))) float f32x2_cvt;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cvt;
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return *reinterpret_cast<const bf16_t*>(&b);
}
__device__ __forceinline__ float rdbf(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const f32x2_cvt v = {lo, hi};
  const bf16x2_cvt r = __builtin_convertvector(v, bf16x2_cvt);
  return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return *reinterpret_cast<bf16x8*>(&v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ------------------------------------------------------------------------------------------------
// One launch for the same small kernel of every request of a cohort (top-k, tree bookkeeping, accept, compaction, gathers: single-
// or few-workgroup kernels whose cost is the launch itself — 72 of them per four-request round when issued request by request).
// Each such kernel is written as a __device__ body + a functor `<name>_fn`; batch4_kernel runs the body with the argument pack of
// request blockIdx.z.  The arithmetic is the body's, so a batched request computes exactly what its own launch would.
// ------------------------------------------------------------------------------------------------
template <class... Ts> struct ArgPack;
template <> struct ArgPack<> {};
template <class T, class... Ts> struct ArgPack<T, Ts...> { T v; ArgPack<Ts...> rest; };
static inline ArgPack<> make_pack() { return ArgPack<>{}; }
template <class T, class... Ts> static inline ArgPack<T, Ts...> make_pack(T a, Ts... r) {
  ArgPack<T, Ts...> p;
  p.v = a;
  p.rest = make_pack(r...);
  return p;
}
template <class F, class... Us> __device__ __forceinline__ void pack_apply(F f, const ArgPack<>&, Us... a) { f(a...); }
template <class F, class T, class... Ts, class... Us>
__device__ __forceinline__ void pack_apply(F f, const ArgPack<T, Ts...>& p, Us... a) { pack_apply(f, p.rest, a..., p.v); }
#define MAX_COHORT 8  // requests that can share one weight pass (round 5: eight; gemm_c8.h)
template <class P> struct Packs4 { P p[MAX_COHORT]; };  // (the name is round 2's: up to MAX_COHORT packs)
template <class Fn, int TPB, class P>
__global__ __launch_bounds__(TPB) void batch4_kernel(Packs4<P> a) {
  const int z = blockIdx.z;  // (selected with compares: a dynamically indexed kernel argument would be copied to scratch; ONE call
  P p = a.p[0];              //  site: a body's static __shared__ arrays must not be instantiated once per request)
  if (z == 1) p = a.p[1];
  else if (z == 2) p = a.p[2];
  else if (z == 3) p = a.p[3];
  else if (z == 4) p = a.p[4];
  else if (z == 5) p = a.p[5];
  else if (z == 6) p = a.p[6];
  else if (z == 7) p = a.p[7];
  pack_apply(Fn{}, p);
}

// ------------------------------------------------------------------------------------------------
// Device-resident round state: nothing here ever needs the host between rounds.
// ------------------------------------------------------------------------------------------------
struct DevState {
  int n_ctx;        // committed context length n (tokens[0..n) are final; target KV rows [0,n) valid)
  int n_prev;       // n before the last accept (base of the last verify's KV rows)
  int accept_len;   // a of the last round
  int best;         // best candidate row of the last round
  int next_token;   // token sampled by the target after the accepted prefix (root of the next tree)
  int new_token;    // generated tokens so far (spec_model_ours.py:476, utils.py:582)
  int rounds;
  int done;         // bit0: eos seen (spec_model_ours.py:544)  bit1: new_token > max_new_tokens (:546)
                    // bit2: the next round would not fit a KV cache (the reference's KVCache.cat raises there: kv_cache.py:40-58)
  int max_new_tokens;
  int eos_token_id;
  int draft_len;    // n_c : rows of the draft's stable KV (cnets_ours.py:1108)
  int draft_real_len; // position base of the next catch-up row (cnets_ours.py:416-418, 862-867)
  int n_leaf, max_depth; // shape of the current tree's retrieve table
  int tree_T;       // nodes in the current tree (total_token, or 1 for the AR baseline)
  int rope_delta;   // Qwen2.5-VL: cached rope_deltas added to every decode position (utils.py:397-402); 0 otherwise
  int kv_cap, draft_cap;  // rows of the target / draft KV caches
  int stop2;        // second stop token (llama-3 "<|eot_id|>", spec_model_ours.py:268-269,540-542); -1 = none
  int draft_rope_rows;  // rows of the draft's rotary tables: draft rows rotate at their UNCOMPRESSED position (cnets_ours.py:845-868)
  int draft_round_rows; // draft KV rows one round appends behind the stable KV: catch-up (<= depth+2) + top_k per tree level
  int frozen;       // the request had already finished (done != 0) when the last accept step ran: that step and everything after it
                    // in the round left the request's state alone (a cohort keeps launching rounds until its last request is done)
};
#define KV_GUARD_ROWS 64  // rows kept free beyond the next tree (the AR baseline polls `done` only every 16 steps)

// ------------------------------------------------------------------------------------------------
// Skinny GEMM  Y[M,N] = X[M,K] · W[N,K]^T   (M <= 32·MT, MT in {1,2}; every weight byte is streamed from HBM exactly once per call)
//
// Weight layout "W32" (built once at load by pack_w32_kernel): the matrix is cut into tiles of 32 rows x 16 k; a tile is
// stored as the 1 KiB image of the A operand of v_mfma_f32_32x32x16_bf16 — lane l holds W[32t + (l&31)][16c + 8(l>>5) .. +8] —
// and the tiles of one 32-row block follow each other along k.  A wave therefore streams its share of K with one fully
// coalesced 1 KiB global_load_dwordx4 per MFMA straight into registers (no LDS round trip: nothing is reused across waves).
//   grid  = (row blocks, S)   S = split-K over workgroups (small N would otherwise leave most of the 256 CUs idle)
//   block = NW waves, each owning a contiguous K range of the workgroup's split; register double-buffering keeps UNROLL
//           loads per operand in flight behind the MFMAs; waves are reduced through LDS in a fixed order (deterministic).
//   X (activations, L2-resident) is fetched in whole 128-B lines and re-shaped into the B operand through a per-wave LDS image
//   (MT images when M > 32: the weight tile in registers then feeds MT MFMAs).
//   NT = 2: the workgroup owns TWO row blocks (tile, tile + tile2_off) and every staged activation fragment feeds both — half the
//   L2 -> CU activation traffic per weight byte.  Used with MT = 2, where that traffic is what the second tile costs (see below).
//   D[i = n][j = m]: a lane ends with 4 groups of 4 consecutive n for its m -> 8-byte bf16 / 16-byte fp32 stores.
// Epilogues (bf16 results carry the reference's rounding points):
//   NONE / RESIDUAL   plain
//   SWIGLU            weight in "SwiGLU order" (16 gate rows + the 16 up rows of the same outputs per tile), act closes in-lane
//   PARTIAL           fp32 partial sums [S][32·MT][N] that splitk_reduce_kernel finishes (bias, residual, RMSNorm fused)
//   ROPE              q|k|v projection: rotary + KV append (weight in "rope order", see RopeEpi)
// ------------------------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_PARTIAL = 3, EPI_ROPE = 4 };

struct PosSpec {  // position = *base + *base2 + add + (off ? off[m] : (row ? m : 0)) ; kv row = *kv_base + kv_add + m
  const int* base = nullptr;
  const int* base2 = nullptr;
  int add = 0;
  const int* off = nullptr;
  int row = 1;
  const int* kv_base = nullptr;
  int kv_add = 0;
};
// EPI_ROPE: the fused q|k|v projection finishes with the rotary embedding and the KV append (modeling_llama_kv.py:575-594,
// cnets_ours.py:104-119,393-396) instead of a separate pass over the qkv rows.  rotate_half pairs (d, d+64) of a head must
// meet in one lane, so the q and k rows of the weight are packed in "rope order": 32-row tile t of a head holds
// d = 16t..16t+15 followed by d+64 (qkv_rope_perm(); v rows keep their order).  q goes to Y at its natural column, k / v
// go straight to cache row *kv_base + kv_add + m.
// Cohort mode (m_tile > 0, up to four requests sharing one weight pass): activation tile mt belongs to request mt, which has its own
// positions and its own KV cache -> index [mt]; otherwise index 0 serves every row.
struct RopeEpi {
  const bf16_t* cosT = nullptr;
  const bf16_t* sinT = nullptr;
  PosSpec ps[MAX_COHORT];
  bf16_t* kc[MAX_COHORT] = {};
  bf16_t* vc[MAX_COHORT] = {};
  int s_max = 0, H = 0, H_kv = 0;
  // live rows of activation tile mt when they differ from the launch's m_tile (0 = m_tile): a request whose tree has 33..64 nodes owns TWO
  // tiles of a cohort launch (round 6) — 32 live rows in the first, T - 32 in the second; rows past them must not reach the KV cache
  unsigned char rows[MAX_COHORT] = {};
};

__global__ __launch_bounds__(64) void pack_w32_kernel(const bf16_t* __restrict__ W, int N, int K, bf16_t* __restrict__ P) {
  const int ks = blockIdx.x, tile = blockIdx.y, l = threadIdx.x;
  const int row = tile * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < N) v = *reinterpret_cast<const uint4*>(W + (size_t)row * K + k);
  *reinterpret_cast<uint4*>(P + (((size_t)tile * (K >> 4) + ks) * 64 + l) * 8) = v;
}

// ---- fp8 (OCP e4m3fn) weights: the same tiling with 32 k per 1 KiB tile (lane l: W[32t + (l&31)][32c + 16(l>>5) .. +16]).
// The GEMM stays HBM-bound, so the win is the halved weight stream; activations stay bf16 and the tile is up-converted in
// registers (e4m3 is exactly representable in bf16) for the bf16 MFMA, the per-output-channel scale is applied to the fp32
// accumulator in the epilogue (W8A16).
__global__ __launch_bounds__(64) void pack_w32_fp8_kernel(const unsigned char* __restrict__ W, int N, int K, unsigned char* __restrict__ P) {
  const int ks = blockIdx.x, tile = blockIdx.y, l = threadIdx.x;
  const int row = tile * 32 + (l & 31), k = ks * 32 + (l >> 5) * 16;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < N) v = *reinterpret_cast<const uint4*>(W + (size_t)row * K + k);
  *reinterpret_cast<uint4*>(P + (((size_t)tile * (K >> 5) + ks) * 64 + l) * 16) = v;
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned fp8x2_to_bf16x2(unsigned v, bool hi_word) {
  const f32x2_t f = hi_word ? __builtin_amdgcn_cvt_pk_f32_fp8(v, true) : __builtin_amdgcn_cvt_pk_f32_fp8(v, false);
  return (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xFFFF0000u);  // exact: e4m3 fits bf16
}
__device__ __forceinline__ void fp8x16_to_bf16(uint4 raw, uint4& lo, uint4& hi) {
  lo = make_uint4(fp8x2_to_bf16x2(raw.x, false), fp8x2_to_bf16x2(raw.x, true), fp8x2_to_bf16x2(raw.y, false), fp8x2_to_bf16x2(raw.y, true));
  hi = make_uint4(fp8x2_to_bf16x2(raw.z, false), fp8x2_to_bf16x2(raw.z, true), fp8x2_to_bf16x2(raw.w, false), fp8x2_to_bf16x2(raw.w, true));
}

// Weight-stream load.  Every weight byte is read exactly once per launch by exactly one wave, so it is loaded non-temporal
// (`global_load_dwordx4 ... nt`): the stream does not displace the L2-resident activations / split-K partials.
#ifndef VISPEC_W_NT
#define VISPEC_W_NT 1
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_weight(const uint4* p) {
#if VISPEC_W_NT
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

// Read-once fp32 partials (split-K slabs, attention key-split partials) in the kernels that finish them: -DVISPEC_PARTIAL_NT=1 loads them
// non-temporal (after this read nobody needs the line again) — measured: 3379 / 3380 vs 3401 / 3396 tok/s with plain loads, same box
// (profiles/r05_ab_partials_nontemporal.txt): off
#ifndef VISPEC_PARTIAL_NT
#define VISPEC_PARTIAL_NT 0
#endif
__device__ __forceinline__ float4 ld_partial(const float4* p) {
#if VISPEC_PARTIAL_NT
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

// ---- fp8 ACTIVATIONS (round 4, W8A8: BASELINE config 5 "CDNA4 fp8 MFMA"): the target's q|k|v, gate|up and down GEMMs can take their activations in
// e4m3 as well, one dynamic scale per row (token): x ~ sx[m] * q[m, k].  The GEMM then runs on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the
// bf16 rate, 64 k per instruction = two 1 KiB weight tiles) with unit block scales, and the epilogue multiplies the fp32 accumulator by
// wscale[n] * sx[m].  A different arithmetic from W8A16 (tests/test_fp8_activation_study.py prices it): its own oracle mode, its own tests.
typedef int v8i_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f8_64(uint4 a0, uint4 a1, uint4 b0, uint4 b1, f32x16 c) {
  v8i_t a, b;
  a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
  b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  // cbsz = blgp = 0: both operands OCP e4m3; scale exponents 127 = 2^0.  Lane l supplies 32 bytes of row / column l & 31; WHICH 32 of the
  // 64 k they are does not matter as long as A and B agree (a dot product is order-free over k up to fp32 accumulation inside the
  // instruction, and every kernel of the library feeds it the same way): here bytes [16 hi, +16) and [32 + 16 hi, +16) of the 64.
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
}
// four values of a row -> four e4m3 codes (x / scale, rne, saturating): the arithmetic of quant_rows_e4m3_kernel for the fused form below
__device__ __forceinline__ unsigned quant4_e4m3(float a, float b, float c, float d, float scale) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(a, scale), __fdiv_rn(b, scale), v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(c, scale), __fdiv_rn(d, scale), v, true);
  return (unsigned)v;
}
// One workgroup per row: sx[m] = max(|x[m, :]|, 1e-12) / 448, q[m, k] = e4m3_rne(x[m, k] / sx[m])   (oracle: Ops.linear(..., a8=True))
__global__ __launch_bounds__(256) void quant_rows_e4m3_kernel(const bf16_t* __restrict__ X, int ldx, unsigned char* __restrict__ Q, int ldq,
                                                              float* __restrict__ sx, int K) {
  // The row stays in registers between the maximum and the conversion (10 x 16 B per thread: K <= 20480, every model here): one trip to
  // L2 per element instead of two, all of a thread's loads in flight at once.  Longer rows take the second pass from memory.
  constexpr int KEEP = 10;
  __shared__ float s_m[4];
  const int m = blockIdx.x;
  const bf16_t* x = X + (size_t)m * ldx;
  uint4 keep[KEEP];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < KEEP; ++c) {
    const int k = (c * 256 + threadIdx.x) * 8;
    keep[c] = k < K ? *reinterpret_cast<const uint4*>(x + k) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < KEEP; ++c) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&keep[c]);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(bf2f(e[i])));
  }
  for (int k = (KEEP * 256 + threadIdx.x) * 8; k < K; k += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + k);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(bf2f(e[i])));
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  const float scale = __fdiv_rn(fmaxf(amax, 1e-12f), 448.0f);
  if (threadIdx.x == 0) sx[m] = scale;
  auto put = [&](const uint4& v, int k) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    *reinterpret_cast<uint2*>(Q + (size_t)m * ldq + k) = make_uint2(quant4_e4m3(bf2f(e[0]), bf2f(e[1]), bf2f(e[2]), bf2f(e[3]), scale),
                                                                    quant4_e4m3(bf2f(e[4]), bf2f(e[5]), bf2f(e[6]), bf2f(e[7]), scale));
  };
#pragma unroll
  for (int c = 0; c < KEEP; ++c) {
    const int k = (c * 256 + threadIdx.x) * 8;
    if (k < K) put(keep[c], k);
  }
  for (int k = (KEEP * 256 + threadIdx.x) * 8; k < K; k += 256 * 8) put(*reinterpret_cast<const uint4*>(x + k), k);
}

// bytes of LDS one staged X group takes: UNROLL k-step images of 1 KiB, padded so that both the staging writes
// (8 lanes = one 128-B row segment -> 8 different 16-B bank groups) and the fragment reads are conflict-free
#define XS_STEP 1056
#define XS_HALF 528
// MT = 32-row activation tiles per launch (M <= 32*MT): every weight tile in registers feeds MT MFMAs
// Two-tile (MT = 2: cohorts, trees of 33..64 nodes) instantiation: ONE activation LDS buffer per wave (same-wave LDS traffic is
// processed in issue order, so the second buffer is not needed for correctness) and a register budget for 3 waves per SIMD:
// 33.8 KB of LDS and 152 VGPRs instead of 67.6 KB / 181 -> three workgroups per CU instead of two (1030 -> 1048 tok/s for one cohort
// lane).  What the second tile really costs is the activation traffic itself: tools/gemm_bench.py (MS=30,60, variants 10100 / 20100)
// — gate|up 37.4 us at M = 30, 49.3 us at M = 60, 38.4 us at M = 60 WITHOUT the activation loads: every one of the 688 workgroups
// re-reads the whole 64 x 4096 block from L2 (360 MB per launch, ~33 TB/s: the L2's own limit).  The library therefore launches the
// two-tile GEMMs in the PAIRED form (NT = 2: two row blocks per workgroup, each staged activation fragment feeds both — 238 VGPRs,
// 2 waves per SIMD); the single-block two-tile form above remains for odd tile counts and for the A/B switch VISPEC_MT2_SINGLE_BLOCK.
#ifndef VISPEC_MT2_LDSBUF
#define VISPEC_MT2_LDSBUF 1
#endif
#ifndef VISPEC_MT2_MINWAVES
#define VISPEC_MT2_MINWAVES 3
#endif
// W8A16 two-tile instantiations (fp8 cohorts of two, 33..64-node trees, and since round 5 the two-tile slab of the draft's fp8 lm_head in a cohort
// of 5..8): the up-conversion temporaries do not fit the register budget of the bf16 forms.  NT = 1: 16..28 B of scratch at 3 waves per SIMD ->
// 2 waves per SIMD, no scratch.  NT = 2 (the paired form): 80..92 B of scratch at 2 waves per SIMD; at ONE wave per SIMD it has none (222 VGPRs +
// 80 AGPRs) but the large grids lose their latency hiding — same box, Qwen2.5-VL-7B W8A16, 4 lanes x cohort 8 (the draft's 545 MB lm_head runs
// this kernel four times a round): 2429 tok/s with the scratch, 2281 without; cohorts of two 1264 / 1291 (profiles/r05_w8_two_tile_bounds_ab.txt).
// The scratch stays for NT = 2 (-DVISPEC_W8_MT2NT2_MINWAVES=1 builds the other form); converting one weight tile at a time instead of both moved
// it from 92 to 84 B: the pressure is the double-buffered staging registers, not the conversion.
#ifndef VISPEC_W8_MT2_MINWAVES
#define VISPEC_W8_MT2_MINWAVES 2
#endif
#ifndef VISPEC_W8_MT2NT2_MINWAVES
#define VISPEC_W8_MT2NT2_MINWAVES 2
#endif
#ifndef VISPEC_MT2NT2_LDSBUF
#define VISPEC_MT2NT2_LDSBUF 1
#endif
template <int MT, int NT = 1>
#ifndef VISPEC_MT1_LDSBUF
#define VISPEC_MT1_LDSBUF 2
#endif
constexpr int gemm_w32_xbufs() { return MT == 2 ? (NT == 2 ? VISPEC_MT2NT2_LDSBUF : VISPEC_MT2_LDSBUF) : (MT > 2 ? 1 : VISPEC_MT1_LDSBUF); }
template <int NT, int UNROLL, int NW, int MT = 1>
constexpr int gemm_w32_lds_bytes() {
  return (NW * MT * gemm_w32_xbufs<MT, NT>() * UNROLL * XS_STEP) > (NW * NT * MT * 4096) ? (NW * MT * gemm_w32_xbufs<MT, NT>() * UNROLL * XS_STEP)
                                                                                           : (NW * NT * MT * 4096);
}

// SLAB (m_tile < 0, MT = 1: up to four requests, MT = 2 (round 5): five to eight; the draft GEMMs of a cohort, whose requests have at most 8 live rows each — top_k rows of a tree level, depth + 2
// catch-up rows, one root row): the ONE activation tile holds up to four requests, tile row m = 8 t + i is row i of request t, which lives
// at row 32 t + i of X / Y / R (the cohort members' tile-aliased workspaces, unchanged).  The weight pass then costs what a single request's
// costs — one 32-row activation block per workgroup instead of the 128 rows of the wide form — and a row is the same dot products in the
// same order as in the single-request launch (split-K partials are indexed by the tile row m).
template <int NT, int EPI, int UNROLL, int NW, int DBG = 0, int W8 = 0 /* 0 bf16 weights, 1 e4m3 weights x bf16 activations, 2 e4m3 x e4m3 */, int MT = 1, bool SLAB = false>  // DBG (tools/gemm_bench.py only): 1 = no activation loads, 2 = no epilogue, 3 = only tile 0's activations loaded (wrong results: traffic upper bound)
__global__ __launch_bounds__(NW * 64, (MT == 2 ? (NT == 1 ? (W8 == 1 ? VISPEC_W8_MT2_MINWAVES : VISPEC_MT2_MINWAVES) : (W8 == 1 ? VISPEC_W8_MT2NT2_MINWAVES : 2)) : 1)) void gemm_w32_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ P,
                                                           int tile2_off, const bf16_t* __restrict__ bias, void* __restrict__ Yv,
                                                           int ldy, const bf16_t* __restrict__ R, int ldr, int M, int N, int K,
                                                           int S, const float* __restrict__ wscale, RopeEpi re, int m_tile,
                                                           const float* __restrict__ xscale = nullptr) {
  // W8 == 2 (A8): X holds e4m3 codes (pointer and ldx in 2-BYTE units, i.e. ldx = row bytes / 2), xscale[row of X] their per-row scales
  constexpr bool A8 = W8 == 2;
  // m_tile > 0 ("cohort"): the MT activation tiles belong to different requests, rows 32 mt .. 32 mt + m_tile - 1 of each are valid
  // (instead of the contiguous rows 0 .. M-1); the weights are still streamed once for all of them
  static_assert(!SLAB || MT <= 2, "slab mode packs up to four requests into each activation tile (MT = 2: five to eight requests)");
  auto row_ok = [&](int m) { return SLAB ? ((m & 7) < -m_tile && m < M) : (m_tile > 0 ? ((m & 31) < m_tile) : (m < M)); };
  auto grow = [&](int m) { return SLAB ? ((m >> 3) << 5) + (m & 7) : m; };  // row of X / Y / R that tile row m stands for
  static_assert(UNROLL == 4 || UNROLL == 8, "staging map is written for 4 or 8 k-steps per group");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
  WGCLK_SCOPE(40 + EPI + 8 * (SLAB ? 1 : 0), Yv);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index provably uniform
  const int j = lane & 31, hi = lane >> 5;
  const int tile = blockIdx.x, split = blockIdx.y;
  // a "step" is one 1 KiB weight tile: 16 k in bf16, 32 k in fp8 (two MFMAs); a group is 64 k either way.  A8: a step is one MFMA = 64 k =
  // TWO weight tiles, a group 128 k (the same 128 bytes of X per row)
  constexpr int KSTEP = A8 ? 64 : (W8 ? 32 : 16);
  constexpr int TPS = A8 ? 2 : 1;                  // weight tiles per step
  constexpr int LOADS = W8 ? UNROLL / 2 : UNROLL;  // steps per group
  constexpr int XU = A8 ? 32 : KSTEP;              // 2-byte units of an X row one step covers
  const int KS = K / KSTEP;
  const int ks_lo = (int)((long)KS * split / S), ks_hi = (int)((long)KS * (split + 1) / S);
  const int len = ks_hi - ks_lo;
  const int w_lo = ks_lo + (int)((long)len * wave / NW), w_hi = ks_lo + (int)((long)len * (wave + 1) / NW);
  const uint4* pa0 = reinterpret_cast<const uint4*>(P) + (size_t)tile * KS * TPS * 64 + lane + (size_t)w_lo * TPS * 64;
  const uint4* pa1 = (NT == 2) ? reinterpret_cast<const uint4*>(P) + (size_t)(tile + tile2_off) * KS * TPS * 64 + lane + (size_t)w_lo * TPS * 64 : pa0;
  f32x16 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.f;

  // ---- X staging map: one instruction = (64 / (2*UNROLL)) rows x (UNROLL*32) contiguous bytes -> whole 128/256-B lines.
  // The activations are re-read by every workgroup from L2; fetching them fragment-shaped (32 rows x 32 B per instruction)
  // costs twice the L1/TA requests of the weight stream and was measured to cap the kernel at ~3.7 TB/s (tools/stream_probe).
  constexpr int SEGS = 2 * UNROLL, RPI = 64 / SEGS, NINST = 32 / RPI;
  const int seg = lane % SEGS, srow0 = lane / SEGS;
  constexpr int XTILE = UNROLL * XS_STEP;  // one staged group of one activation tile; per wave: [2 buffers][MT tiles]
  constexpr int XBUFS = gemm_w32_xbufs<MT, NT>();  // same-wave LDS traffic is processed in issue order: one buffer is enough for correctness
  unsigned char* xs = smem_g + wave * (MT * XBUFS * XTILE);
  // one running base pointer + 32-bit per-row element offsets (8 address registers less than a pointer per row at MT = 2)
  const bf16_t* xbase = X + (size_t)w_lo * XU + seg * 8;
  unsigned xo[MT][NINST];
  int woff[NINST];
#pragma unroll
  for (int i = 0; i < NINST; ++i) {
    const int row = srow0 + i * RPI;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)  // rows >= M read row 0, never stored
      xo[mt][i] = (unsigned)(row_ok(32 * mt + row) ? grow(32 * mt + row) : 0) * (unsigned)ldx;
    woff[i] = (seg >> 1) * XS_STEP + (seg & 1) * XS_HALF + row * 16;
  }
  const int roff = hi * XS_HALF + j * 16;

  struct Regs {
    uint4 a[NT][LOADS * TPS];
    uint4 x[MT][NINST];
  };
  auto load = [&](Regs& g) {  // all loads unconditional plain global loads; X first (it is consumed first, through LDS)
    if (DBG != 1) {
#pragma unroll
      for (int mt = 0; mt < (DBG == 3 ? 1 : MT); ++mt)
#pragma unroll
        for (int i = 0; i < NINST; ++i) g.x[mt][i] = *reinterpret_cast<const uint4*>(xbase + xo[mt][i]);
    }
#pragma unroll
    for (int u = 0; u < LOADS * TPS; ++u) {
      g.a[0][u] = ld_weight(pa0 + u * 64);
      if (NT == 2) g.a[NT - 1][u] = ld_weight(pa1 + u * 64);
    }
    pa0 += 64 * LOADS * TPS;
    pa1 += 64 * LOADS * TPS;
    xbase += 16 * UNROLL;
  };
  auto compute = [&](const Regs& g, int buf) {
    unsigned char* xb = xs + (XBUFS == 2 ? buf : 0) * (MT * XTILE);
    if (DBG != 1) {
#pragma unroll
      for (int mt = 0; mt < (DBG == 3 ? 1 : MT); ++mt)
#pragma unroll
        for (int i = 0; i < NINST; ++i) *reinterpret_cast<uint4*>(xb + mt * XTILE + woff[i]) = g.x[mt][i];
    }
    // same-wave LDS traffic is processed in issue order: the fragment reads below see the writes above
    if (!W8) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const uint4 bv = (DBG == 1) ? make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u)
                                      : *reinterpret_cast<const uint4*>(xb + (DBG == 3 ? 0 : mt) * XTILE + u * XS_STEP + roff);
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(g.a[t][u]), as_bf16x8(bv), acc[t][mt], 0, 0, 0);
        }
      }
    } else if (A8) {
      // step sp = weight tiles 2 sp, 2 sp + 1: this lane's bytes are k = 64 sp + 16 hi + [0, 16) and + 32 — 16-byte slots 4 sp + hi and
      // 4 sp + 2 + hi of the staged 128-byte row chunk, i.e. staged images (2 sp, half hi) and (2 sp + 1, half hi)
#pragma unroll
      for (int sp = 0; sp < LOADS; ++sp) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned char* xs0 = xb + mt * XTILE + (2 * sp) * XS_STEP + hi * XS_HALF + j * 16;
          const uint4 b0 = *reinterpret_cast<const uint4*>(xs0);
          const uint4 b1 = *reinterpret_cast<const uint4*>(xs0 + XS_STEP);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][mt] = mfma_f8_64(g.a[t][(2 * sp) % (LOADS * TPS)], g.a[t][(2 * sp + 1) % (LOADS * TPS)], b0, b1, acc[t][mt]);
        }
      }
    } else {
      // fp8 tile c holds k = 32c + 16*hi + [0,16) for this lane: its two 8-k halves pair with the staged bf16 step (2c + hi),
      // half 0 / half 1 of the activation image
#pragma unroll
      for (int c = 0; c < LOADS; ++c) {
        uint4 a_lo[NT], a_hi[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fp8x16_to_bf16(g.a[t][c], a_lo[t], a_hi[t]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {  // each activation fragment is read once and feeds every row block
          const unsigned char* xstep = xb + mt * XTILE + (2 * c + hi) * XS_STEP + j * 16;
          const uint4 b0 = *reinterpret_cast<const uint4*>(xstep);
          const uint4 b1 = *reinterpret_cast<const uint4*>(xstep + XS_HALF);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo[t]), as_bf16x8(b0), acc[t][mt], 0, 0, 0);
            acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi[t]), as_bf16x8(b1), acc[t][mt], 0, 0, 0);
          }
        }
      }
    }
  };
  const int n_steps = w_hi - w_lo;
  const int n_groups = n_steps / LOADS;  // wave-uniform
  if (n_groups > 0) {
    // Register double-buffering: group g+1 is in flight while group g feeds the matrix core.  The steady-state loop issues
    // its prefetch UNCONDITIONALLY: a prefetch under `if (more)` makes hipcc size every s_waitcnt for the no-prefetch path,
    // i.e. it waits for the prefetch it just issued (seen in the ISA as vmcnt(3..0) with 16 loads outstanding).
    Regs r0, r1;
    load(r0);
    int g = 0;
    while (g + 2 < n_groups) {
      load(r1);
      compute(r0, 0);
      load(r0);
      compute(r1, 1);
      g += 2;
    }
    if (g + 1 < n_groups) {
      load(r1);
      compute(r0, 0);
      compute(r1, 1);
    } else {
      compute(r0, 0);
    }
  }
  {  // leftover steps (less than a group): fragment-shaped X loads straight from global
    const bf16_t* px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      px[mt] = X + (size_t)(row_ok(32 * mt + j) ? grow(32 * mt + j) : 0) * ldx + (A8 ? hi * 8 : (W8 ? hi * 16 : hi * 8)) + (size_t)(w_lo + n_groups * LOADS) * XU;
    for (int rstep = n_groups * LOADS; rstep < n_steps; ++rstep) {
      const uint4 av = pa0[0];
      const uint4 av1 = pa1[0];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const uint4 bv = *reinterpret_cast<const uint4*>(px[mt]);
        if (A8) {  // the step's second weight tile / second 16 bytes of the row (+ 32 bytes = 16 units)
          const uint4 bw = *reinterpret_cast<const uint4*>(px[mt] + 16);
          acc[0][mt] = mfma_f8_64(av, pa0[64], bv, bw, acc[0][mt]);
          if (NT == 2) acc[NT - 1][mt] = mfma_f8_64(av1, pa1[64], bv, bw, acc[NT - 1][mt]);
        } else if (!W8) {
          acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av), as_bf16x8(bv), acc[0][mt], 0, 0, 0);
          if (NT == 2) acc[NT - 1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av1), as_bf16x8(bv), acc[NT - 1][mt], 0, 0, 0);
        } else {
          const uint4 bv1 = *reinterpret_cast<const uint4*>(px[mt] + 8);
          uint4 a_lo, a_hi;
          fp8x16_to_bf16(av, a_lo, a_hi);
          acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(bv), acc[0][mt], 0, 0, 0);
          acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(bv1), acc[0][mt], 0, 0, 0);
          if (NT == 2) {
            fp8x16_to_bf16(av1, a_lo, a_hi);
            acc[NT - 1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(bv), acc[NT - 1][mt], 0, 0, 0);
            acc[NT - 1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(bv1), acc[NT - 1][mt], 0, 0, 0);
          }
        }
        px[mt] += XU;
      }
      pa0 += 64 * TPS;
      pa1 += 64 * TPS;
    }
  }
  if (DBG == 2) {
    if (acc[0][0][0] == 12345.f) reinterpret_cast<float*>(Yv)[0] = acc[0][0][1];
    return;
  }
  // cross-wave reduction through LDS (aliases the X staging area), fixed order wave 0 + 1 + ... (deterministic)
  float(*red)[NT][MT][64][16] = reinterpret_cast<float(*)[NT][MT][64][16]>(smem_g);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][t][mt][lane][r] = acc[t][mt][r];
  __syncthreads();
  if (EPI == EPI_ROPE) {
    // waves 0/1 own column groups q and q+2 of the tile: packed columns c = 8q + 4hi + r (< 16) and c + 16 = its rotate_half partner
    // (NT = 2: waves 2/3 do the same for the workgroup's second row block)
    static_assert(NW >= 2 * NT, "rope epilogue: two waves per row block");
    const int tb = (NT == 2) ? (wave >> 1) : 0;
    const int tile_b = tile + tb * tile2_off;
    for (int mt = 0; mt < MT; ++mt) {
      const int m = 32 * mt + j;
      const int rq = SLAB ? (m >> 3) : (m_tile > 0 ? mt : 0);  // request owning this tile (slab mode: this tile row)
      const int mr = SLAB ? (j & 7) : (m_tile > 0 ? j : m);    // row index inside the request
      if (wave < 2 * NT && row_ok(m)) {
        const int qq = wave & 1;
        float a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sa = red[0][tb][mt][lane][4 * qq + r], sb = red[0][tb][mt][lane][4 * (qq + 2) + r];
#pragma unroll
          for (int w = 1; w < NW; ++w) {
            sa += red[w][tb][mt][lane][4 * qq + r];
            sb += red[w][tb][mt][lane][4 * (qq + 2) + r];
          }
          a[r] = sa;
          b[r] = sb;
        }
        const int ncol = tile_b * 32 + 8 * qq + 4 * hi;  // packed column of a[0]
        const int h = ncol >> 7, t4 = (ncol & 127) >> 5, c = ncol & 31;
        // slab mode: the request differs from lane to lane — select its arguments under branches (an indexed kernel argument would be
        // copied to scratch, DESIGN.md "A compiler lesson")
        PosSpec ps_s = re.ps[0];
        bf16_t *kc_s = re.kc[0], *vc_s = re.vc[0];
        if (SLAB) {
          if (rq == 1) { ps_s = re.ps[1]; kc_s = re.kc[1]; vc_s = re.vc[1]; }
          if (rq == 2) { ps_s = re.ps[2]; kc_s = re.kc[2]; vc_s = re.vc[2]; }
          if (rq == 3) { ps_s = re.ps[3]; kc_s = re.kc[3]; vc_s = re.vc[3]; }
          if (MT == 2) {
            if (rq == 4) { ps_s = re.ps[4]; kc_s = re.kc[4]; vc_s = re.vc[4]; }
            if (rq == 5) { ps_s = re.ps[5]; kc_s = re.kc[5]; vc_s = re.vc[5]; }
            if (rq == 6) { ps_s = re.ps[6]; kc_s = re.kc[6]; vc_s = re.vc[6]; }
            if (rq == 7) { ps_s = re.ps[7]; kc_s = re.kc[7]; vc_s = re.vc[7]; }
          }
        }
        const PosSpec& ps_ = SLAB ? ps_s : re.ps[rq];
        bf16_t* const kc_ = SLAB ? kc_s : re.kc[rq];
        bf16_t* const vc_ = SLAB ? vc_s : re.vc[rq];
        const int kvrow = (ps_.kv_base ? *ps_.kv_base : 0) + ps_.kv_add + mr;
        if (h < re.H + re.H_kv) {
          const int d = 16 * t4 + c, c1 = h * 128 + d, c2 = c1 + 64;  // natural columns of a[] / b[]
          const int pos = (ps_.base ? *ps_.base : 0) + (ps_.base2 ? *ps_.base2 : 0) + ps_.add +
                          (ps_.off ? ps_.off[mr] : (ps_.row ? mr : 0));
          float o1[4], o2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x1 = a[r], x2 = b[r];
            if (W8) { x1 *= wscale[c1 + r]; x2 *= wscale[c2 + r]; }
            if (A8) { x1 *= xscale[grow(m)]; x2 *= xscale[grow(m)]; }
            if (bias) { x1 += bf2f(bias[c1 + r]); x2 += bf2f(bias[c2 + r]); }
            x1 = rdbf(x1);
            x2 = rdbf(x2);
            const float cs = bf2f(re.cosT[(size_t)pos * 128 + d + r]), sn = bf2f(re.sinT[(size_t)pos * 128 + d + r]);
            o1[r] = rdbf(rdbf(x1 * cs) + rdbf(-x2 * sn));
            o2[r] = rdbf(rdbf(x2 * cs) + rdbf(x1 * sn));
          }
          bf16_t* dst = (h < re.H) ? reinterpret_cast<bf16_t*>(Yv) + (size_t)grow(m) * ldy + c1
                                   : kc_ + ((size_t)(h - re.H) * re.s_max + kvrow) * 128 + d;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
          *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]));
        } else {  // v head: natural order, columns ncol + r and ncol + 16 + r
          float o1[4], o2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x1 = a[r], x2 = b[r];
            if (W8) { x1 *= wscale[ncol + r]; x2 *= wscale[ncol + 16 + r]; }
            if (A8) { x1 *= xscale[grow(m)]; x2 *= xscale[grow(m)]; }
            if (bias) { x1 += bf2f(bias[ncol + r]); x2 += bf2f(bias[ncol + 16 + r]); }
            o1[r] = rdbf(x1);
            o2[r] = rdbf(x2);
          }
          bf16_t* dst = vc_ + ((size_t)(h - re.H - re.H_kv) * re.s_max + kvrow) * 128 + (ncol & 127);
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
          *reinterpret_cast<uint2*>(dst + 16) = make_uint2(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]));
        }
      }
    }
    return;
  }
  if (EPI == EPI_SWIGLU) {
    // "SwiGLU order" (one 32-row tile = the 16 gate rows and the 16 up rows of the same 16 outputs, packed by the loader): column
    // group qq (< 2) holds gate values, group qq + 2 the matching up values, so act = silu(gate) * up closes inside one lane and a
    // gate|up GEMM is N/16 equal workgroups of ONE tile stream (688 for I = 11 008: 2.7 rounds of 256 KB instead of 1.34 of 512 KB).
    for (int idx = wave; idx < 2 * MT * NT; idx += NW) {
      const int qq = idx & 1, mt = (idx >> 1) % MT, tb = (idx >> 1) / MT;
      const int m = 32 * mt + j;
      float a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sa = red[0][tb][mt][lane][4 * qq + r], sb = red[0][tb][mt][lane][4 * (qq + 2) + r];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          sa += red[w][tb][mt][lane][4 * qq + r];
          sb += red[w][tb][mt][lane][4 * (qq + 2) + r];
        }
        a[r] = sa;
        b[r] = sb;
      }
      const int n = (tile + tb * tile2_off) * 16 + 8 * qq + 4 * hi;  // output column of a[0]; gate row n, up row N + n of the natural weight
      if (row_ok(m) && n < N) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = a[r], u = b[r];
          if (W8) { y *= wscale[n + r]; u *= wscale[N + n + r]; }
          if (A8) { y *= xscale[grow(m)]; u *= xscale[grow(m)]; }
          if (bias) { y += bf2f(bias[n + r]); u += bf2f(bias[N + n + r]); }
          y = rdbf(y);
          u = rdbf(u);
          const float act = rdbf(y / (1.0f + __expf(-y)));
          o[r] = rdbf(act * u);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Yv) + (size_t)grow(m) * ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
      }
    }
    return;
  }
  constexpr int NGROUPS = 4 * NT * MT;
  for (int idx = wave; idx < NGROUPS; idx += NW) {
    const int q = idx & 3, tm = idx >> 2;
    const int mt = tm % MT, tt = tm / MT;
    const int m = 32 * mt + j;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sum = red[0][tt][mt][lane][4 * q + r];
#pragma unroll
      for (int w = 1; w < NW; ++w) sum += red[w][tt][mt][lane][4 * q + r];
      v[r] = sum;
    }
    const int n = (tile + (tt ? tile2_off : 0)) * 32 + 8 * q + 4 * hi;
    if (W8 && n < N) {  // per-output-channel dequantisation scale on the fp32 accumulator
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= wscale[n + r];
    }
    if (A8 && row_ok(m)) {  // ... then the row's activation scale
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= xscale[grow(m)];
    }
    if (row_ok(m) && n < N) {
      if (EPI == EPI_PARTIAL) {
        float* part = reinterpret_cast<float*>(Yv) + ((size_t)split * (32 * MT) + m) * N + n;
        *reinterpret_cast<float4*>(part) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        bf16_t* Y = reinterpret_cast<bf16_t*>(Yv);
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = v[r];
          if (bias) y += bf2f(bias[n + r]);
          y = rdbf(y);
          if (EPI == EPI_RESIDUAL) y = rdbf(bf2f(R[(size_t)grow(m) * ldr + n + r]) + y);
          o[r] = y;
        }
        *reinterpret_cast<uint2*>(Y + (size_t)grow(m) * ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
      }
    }
  }
}

// Finishes a split-K GEMM: y = bf16(sum_s part[s] + bias) ; h = bf16(R + y) if R ; optional fused RMSNorm of h
// (cnets_ours.py:513-527) written to `normed`.  One workgroup per row m; every stage rounds where the reference's graph does.
__global__ __launch_bounds__(1024) void splitk_reduce_kernel(const float* __restrict__ part, int S, int Mpad, int N,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R, int ldr,
                                                            bf16_t* __restrict__ Y, int ldy, const bf16_t* __restrict__ norm_w,
                                                            bf16_t* __restrict__ normed, int ldn, float eps, int m_tile,
                                                            unsigned char* __restrict__ q8 = nullptr, float* __restrict__ sx8 = nullptr) {
  // q8 != nullptr (W8A8, with `normed`): the normed row is ALSO written as e4m3 codes q8[row][N] with its scale sx8[row] — exactly what
  // quant_rows_e4m3_kernel would make of `normed` (the next GEMM's input), without that launch.
  // Latency-bound (a few MB out of L2 per launch, 80 launches per round): every load of a row chunk — up to 8 partial slabs with
  // clamped (always valid) slab indices, residual, bias, norm weight — is issued before the first use, and a row that fits one
  // pass of the block (N <= 4 x threads: every model here) keeps its values in registers across the block-wide sum of squares.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
  float* hrow = reinterpret_cast<float*>(smem_r);  // [N] values of h (bf16-representable): only for rows longer than one pass
  __shared__ float partsum[16];
  const int m = blockIdx.x;
  if (m_tile > 0 && (m & 31) >= m_tile) return;  // cohort mode: padding rows between the requests' tiles
  if (m_tile < 0 && (m & 7) >= -m_tile) return;  // slab mode (gemm_w32_kernel SLAB): partial row m = 8 t + i is row 32 t + i of Y / R / normed
  const int mo = m_tile < 0 ? ((m >> 3) << 5) + (m & 7) : m;
  const int nthreads = blockDim.x;
  const bool one_pass = N <= nthreads * 4;
  float ss = 0.f;
  float keep[4] = {0.f, 0.f, 0.f, 0.f};
  uint2 wkeep = make_uint2(0, 0);
  for (int n = threadIdx.x * 4; n < N; n += nthreads * 4) {
    float4 p[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) p[sl] = ld_partial(reinterpret_cast<const float4*>(part + ((size_t)min(sl, S - 1) * Mpad + m) * N + n));
    uint2 rv = make_uint2(0, 0), bv = make_uint2(0, 0);
    if (R) rv = *reinterpret_cast<const uint2*>(R + (size_t)mo * ldr + n);
    if (bias) bv = *reinterpret_cast<const uint2*>(bias + n);
    if (normed && one_pass) wkeep = *reinterpret_cast<const uint2*>(norm_w + n);
    float4 a = p[0];
#pragma unroll
    for (int sl = 1; sl < 8; ++sl)
      if (sl < S) { a.x += p[sl].x; a.y += p[sl].y; a.z += p[sl].z; a.w += p[sl].w; }  // fixed order s = 0, 1, ... (deterministic)
    for (int sl = 8; sl < S; ++sl) {  // (never taken by the library: S <= 8)
      const float4 q = *reinterpret_cast<const float4*>(part + ((size_t)sl * Mpad + m) * N + n);
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    }
    const bf16_t* re = reinterpret_cast<const bf16_t*>(&rv);
    const bf16_t* be = reinterpret_cast<const bf16_t*>(&bv);
    float o[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = o[r];
      if (bias) y += bf2f(be[r]);
      y = rdbf(y);
      if (R) y = rdbf(bf2f(re[r]) + y);
      o[r] = y;
      ss += y * y;
    }
    if (Y) *reinterpret_cast<uint2*>(Y + (size_t)mo * ldy + n) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
    if (normed) {
      if (one_pass) { keep[0] = o[0]; keep[1] = o[1]; keep[2] = o[2]; keep[3] = o[3]; }
      else *reinterpret_cast<float4*>(hrow + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (!normed) return;
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) partsum[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (nthreads >> 6); ++w) tot += partsum[w];
  const float inv = 1.0f / sqrtf(tot / (float)N + eps);
  __shared__ float partmax[16];
  float amax = 0.f;
  if (one_pass) {
    const int n = threadIdx.x * 4;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    if (n < N) {
      const bf16_t* we = reinterpret_cast<const bf16_t*>(&wkeep);
      o0 = rdbf(bf2f(we[0]) * rdbf(keep[0] * inv)), o1 = rdbf(bf2f(we[1]) * rdbf(keep[1] * inv));
      o2 = rdbf(bf2f(we[2]) * rdbf(keep[2] * inv)), o3 = rdbf(bf2f(we[3]) * rdbf(keep[3] * inv));
      *reinterpret_cast<uint2*>(normed + (size_t)mo * ldn + n) = make_uint2(pack2(o0, o1), pack2(o2, o3));
    }
    if (!q8) return;
    amax = wave_max(fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3))));
    if ((threadIdx.x & 63) == 0) partmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = 0.f;
    for (int w = 0; w < (nthreads >> 6); ++w) amax = fmaxf(amax, partmax[w]);
    const float scale = __fdiv_rn(fmaxf(amax, 1e-12f), 448.0f);
    if (threadIdx.x == 0) sx8[mo] = scale;
    if (n < N) *reinterpret_cast<unsigned*>(q8 + (size_t)mo * N + n) = quant4_e4m3(o0, o1, o2, o3, scale);
    return;
  }
  for (int n = threadIdx.x * 4; n < N; n += nthreads * 4) {
    const float4 h = *reinterpret_cast<const float4*>(hrow + n);
    const uint2 wv = *reinterpret_cast<const uint2*>(norm_w + n);
    const bf16_t* we = reinterpret_cast<const bf16_t*>(&wv);
    const float o0 = rdbf(bf2f(we[0]) * rdbf(h.x * inv)), o1 = rdbf(bf2f(we[1]) * rdbf(h.y * inv));
    const float o2 = rdbf(bf2f(we[2]) * rdbf(h.z * inv)), o3 = rdbf(bf2f(we[3]) * rdbf(h.w * inv));
    *reinterpret_cast<uint2*>(normed + (size_t)mo * ldn + n) = make_uint2(pack2(o0, o1), pack2(o2, o3));
    if (q8) {  // (each thread re-reads only the elements it wrote: no barrier needed for hrow)
      *reinterpret_cast<float4*>(hrow + n) = make_float4(o0, o1, o2, o3);
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3))));
    }
  }
  if (!q8) return;
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) partmax[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = 0.f;
  for (int w = 0; w < (nthreads >> 6); ++w) amax = fmaxf(amax, partmax[w]);
  const float scale = __fdiv_rn(fmaxf(amax, 1e-12f), 448.0f);
  if (threadIdx.x == 0) sx8[mo] = scale;
  for (int n = threadIdx.x * 4; n < N; n += nthreads * 4) {
    const float4 h = *reinterpret_cast<const float4*>(hrow + n);
    *reinterpret_cast<unsigned*>(q8 + (size_t)mo * N + n) = quant4_e4m3(h.x, h.y, h.z, h.w, scale);
  }
}

// Epilogue of the fp8-weight PREFILL GEMMs (model/target.py scaled_linear): the library GEMM leaves fp32 accumulators acc[M, N] (activations
// x e4m3 codes); y = bf16(acc * scale[n] + bias[n]) — the rounding points of the decode GEMMs' W8A16 epilogue — in ONE pass instead of torch's
// mul / add / cast passes over the fp32 tensor.  Explicit rn mul and add: no FMA contraction, so the result equals torch's op sequence bit for bit.
__global__ __launch_bounds__(256) void scale_bias_cast_kernel(const float* __restrict__ acc, int ld, const float* __restrict__ scale,
                                                              const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int ldo, int N) {
  const int n = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (n >= N) return;
  const float* a = acc + (size_t)blockIdx.y * ld + n;
  const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(scale + n), s1 = *reinterpret_cast<const float4*>(scale + n + 4);
  float v[8] = {__fmul_rn(a0.x, s0.x), __fmul_rn(a0.y, s0.y), __fmul_rn(a0.z, s0.z), __fmul_rn(a0.w, s0.w),
                __fmul_rn(a1.x, s1.x), __fmul_rn(a1.y, s1.y), __fmul_rn(a1.z, s1.z), __fmul_rn(a1.w, s1.w)};
  if (bias) {
    const uint4 b = *reinterpret_cast<const uint4*>(bias + n);
    const bf16_t* be = reinterpret_cast<const bf16_t*>(&b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __fadd_rn(v[i], bf2f(be[i]));
  }
  *reinterpret_cast<uint4*>(out + (size_t)blockIdx.y * ldo + n) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// SwiGLU activation of a [M, 2I] gate|up row block (prefill side: the projections there are library GEMMs):
// out = bf16( bf16(silu(gate)) * up ), the rounding points of F.silu(g) * u on bf16 tensors.  8 outputs per thread.
__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ gu, int ld, bf16_t* __restrict__ out, int ldo, int I) {
  const bf16_t* g = gu + (size_t)blockIdx.y * ld;
  bf16_t* o = out + (size_t)blockIdx.y * ldo;
  const int d = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (d >= I) return;
  const uint4 gv = *reinterpret_cast<const uint4*>(g + d), uv = *reinterpret_cast<const uint4*>(g + I + d);
  const bf16_t* ge = reinterpret_cast<const bf16_t*>(&gv);
  const bf16_t* ue = reinterpret_cast<const bf16_t*>(&uv);
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float y = bf2f(ge[j]);
    r[j] = rdbf(y / (1.0f + __expf(-y))) * bf2f(ue[j]);
  }
  *reinterpret_cast<uint4*>(o + d) = make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
}

// ------------------------------------------------------------------------------------------------
// RMSNorm: y = w * bf16( x * rsqrt(mean(x^2) + eps) )     one workgroup per row
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ Y, int D, float eps) {
  __shared__ float part[4];
  const bf16_t* x = X + (size_t)blockIdx.x * D;
  bf16_t* y = Y + (size_t)blockIdx.x * D;
  float ss = 0.f;
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(x + d);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = bf2f(e[j]);
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = (part[0] + part[1]) + (part[2] + part[3]);
  float inv = 1.0f / sqrtf(tot / (float)D + eps);
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(x + d);
    uint4 wv = *reinterpret_cast<const uint4*>(w + d);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    const bf16_t* we = reinterpret_cast<const bf16_t*>(&wv);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f(we[j]) * rdbf(bf2f(e[j]) * inv);
    uint4 st = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
    *reinterpret_cast<uint4*>(y + d) = st;
  }
}

// Token embedding rows + the first layer's input RMSNorm in one launch (modeling_llama_kv.py:985 + :104-133): X[i] = table[ids[i]],
// Y[i] = w * bf16(X[i] * rsqrt(mean(X[i]^2) + eps)).  One workgroup per row.
__device__ __forceinline__ void embed_rmsnorm_body(const bf16_t* __restrict__ table, const int* __restrict__ ids, bf16_t* __restrict__ X,
                                                            const bf16_t* __restrict__ w, bf16_t* __restrict__ Y, int D, float eps) {
  __shared__ float part[4];
  const bf16_t* x = table + (size_t)ids[blockIdx.x] * D;
  bf16_t* xo = X + (size_t)blockIdx.x * D;
  bf16_t* y = Y + (size_t)blockIdx.x * D;
  float ss = 0.f;
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + d);
    *reinterpret_cast<uint4*>(xo + d) = v;
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf2f(e[j]);
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = (part[0] + part[1]) + (part[2] + part[3]);
  const float inv = 1.0f / sqrtf(tot / (float)D + eps);
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + d);
    const uint4 wv = *reinterpret_cast<const uint4*>(w + d);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    const bf16_t* we = reinterpret_cast<const bf16_t*>(&wv);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f(we[j]) * rdbf(bf2f(e[j]) * inv);
    *reinterpret_cast<uint4*>(y + d) = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
  }
}
__global__ __launch_bounds__(256) void embed_rmsnorm_kernel(const bf16_t* __restrict__ table, const int* __restrict__ ids, bf16_t* __restrict__ X,
                                                            const bf16_t* __restrict__ w, bf16_t* __restrict__ Y, int D, float eps) { embed_rmsnorm_body(table, ids, X, w, Y, D, eps); }
struct embed_rmsnorm_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { embed_rmsnorm_body(a...); }
};

// ------------------------------------------------------------------------------------------------
// Tree-masked attention, flash-decoding style.  hd = 128.
//   grid (nsplit, H_kv, q-tiles x requests), 256 threads.  A workgroup stages 128-key chunks of K and V of one KV head in LDS
//   (coalesced 16-byte loads, XOR-swizzled rows), each wave owns one 32-key tile per chunk:
//     S^T[key][q]  = K_tile · Q^T           8x mfma_32x32x16 (A = K from LDS, B = Q held in registers)
//     online softmax per query column (the 16 scores a lane holds all belong to ONE query; the other 16 are in lane^32)
//     O^T[hd][q]  += V_tile^T · P^T         8x mfma_32x32x16 (A = V^T gathered from LDS, B = P in registers)
//   The 4 waves' partial (m, l, O) are merged through LDS and written as one partial per (q-tile, split).
//   Visibility: key < prefix -> visible to all rows; prefix <= key < prefix+tail -> bit (key-prefix) of mask[row].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int att_swz(int row, int colbyte) { return row * 256 + (colbyte ^ ((row & 15) << 4)); }

// ---- the partial kernel (round-2 form; round 1's ran two head_dim-half waves per key tile) ---------------------------
// One partial (m, l, O^T) per (q-tile, key split):
//   * a workgroup stages 128-key chunks; each of the 4 waves owns ONE 32-key tile of the chunk completely — QK^T, softmax and P·V
//     over all 128 head_dim columns — so nothing is computed twice (the first form ran QK^T + softmax in both head_dim-half waves);
//   * the V^T operand of P·V comes from the hardware transposing read (ds_read_b64_tr_b16: 16 per 32-key tile instead of 64
//     ds_read_u16 + packing); V rows keep their natural order in LDS, 64-byte blocks XOR-ed with (row & 3) -> conflict-free;
//   * up to four requests per launch (blockIdx.z / NQT): a cohort's attention calls run side by side.
// LDS: K 32 KB + V 32 KB, single-buffered (the next chunk waits in registers while the current one is on the matrix cores); two
// workgroups per CU.  The four waves' (m, l, O) are merged through the same 64 KB at the end.
#define ATT2_CHUNK 128
#define ATT2_LDS_BYTES (2 * ATT2_CHUNK * 256 + 1024)
struct AttnReq {
  const bf16_t* Q;
  const bf16_t* Kc;
  const bf16_t* Vc;
  const int* prefix_dev;
  const unsigned long long* mask;
  float* part_o;
  float* part_ml;
  bf16_t* out;  // (reduce kernel / the fused merge)
  int* cnt;     // arrival counters [H_kv * NQT] of the fused merge (zero between launches), or nullptr: separate reduce launch
};
struct AttnArgs { AttnReq r[MAX_COHORT]; };  // up to eight requests of a cohort per launch
// (assignments under uniform branches: the nested ?: form of this selection was compiled as a dynamically indexed kernel argument —
//  the whole AttnArgs copied to scratch, 264 B per lane and 40 more SGPRs in both attention kernels: 15.7 -> 19.2 us per launch)
__device__ __forceinline__ AttnReq attn_req(const AttnArgs& a, int rq) {
  AttnReq r = a.r[0];
  if (rq == 1) r = a.r[1];
  else if (rq == 2) r = a.r[2];
  else if (rq == 3) r = a.r[3];
  else if (rq == 4) r = a.r[4];
  else if (rq == 5) r = a.r[5];
  else if (rq == 6) r = a.r[6];
  else if (rq == 7) r = a.r[7];
  return r;
}
__device__ __forceinline__ void tree_attn_reduce_body(const AttnReq& R, float* __restrict__ sh, int head, int mt, int dpart, int H, int H_kv, int M,
                                                      int tail, int keys_per_wg, int nsplit, int ldo);  // (below, next to its stand-alone kernel)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int att_vswz(int row, int colbyte) { return row * 256 + (colbyte ^ ((row & 3) << 6)); }

// K / V rows are read once per launch by one workgroup: loaded non-temporal (like the weight streams) they do not displace what IS re-read
// (activations, split-K and attention partials) from the L2 / Infinity Cache.  Same box, 4 lanes x cohort 8: 3340 / 3344 -> 3405 / 3413 tok/s
// (+2.0 %), one request's round 4.98 -> 4.90 ms (profiles/r05_ab_attention_kv_nontemporal.txt; -DVISPEC_ATT_KV_NT=0 = plain loads)
#ifndef VISPEC_ATT_KV_NT
#define VISPEC_ATT_KV_NT 1
#endif
__device__ __forceinline__ uint4 att_ld_kv(const uint4* p) {
#if VISPEC_ATT_KV_NT
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
template <bool EAGER>
__global__ __launch_bounds__(256, 2) void tree_attn2_partial_kernel(AttnArgs args, int ldq, int s_max, int H, int H_kv, int M, int tail,
                                                                    int keys_per_wg, int nsplit, int NQT, int ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WGCLK_BEGIN();
  unsigned char* sK = smem;
  unsigned char* sV = smem + ATT2_CHUNK * 256;
  const int split = blockIdx.x, kvh = blockIdx.y;
  const int rq = blockIdx.z / NQT, qt = blockIdx.z - rq * NQT;
  const AttnReq R = attn_req(args, rq);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, hi = lane >> 5;
  const int n_prefix = R.prefix_dev ? *R.prefix_dev : 0;
  const int n_total = n_prefix + tail;
  const int key0 = split * keys_per_wg;
  if (key0 >= n_total) return;
  const int key_end = min(key0 + keys_per_wg, n_total);
  const int G = H / H_kv, MT = (M + 31) >> 5;
  const bf16_t* Kh = R.Kc + (size_t)kvh * s_max * 128;
  const bf16_t* Vh = R.Vc + (size_t)kvh * s_max * 128;
  const int nchunk = (key_end - key0 + ATT2_CHUNK - 1) / ATT2_CHUNK;
  const float scale = 0.08838834764831845f;  // 1/sqrt(128)
  const float sqrt_hd = 11.313708498984761f;
  const float rsqrt_hd = 1.0f / sqrt_hd;
  const int last_key = n_total - 1;  // rows past the end are clamped to the last valid key (never visible): unconditional loads
  const int head = kvh * G + qt / MT, m0 = (qt % MT) * 32;
  const int mrow = m0 + j;
  const bool qvalid = mrow < M;
  uint4 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = *reinterpret_cast<const uint4*>(R.Q + (size_t)(qvalid ? mrow : 0) * ldq + head * 128 + ks * 16 + hi * 8);
  const unsigned long long mbits = (qvalid && R.mask) ? R.mask[mrow] : 0ull;
  float m_run = NEG_INF, l_run = 0.f;
  f32x16 O[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  // staging map: thread -> 8 x 16 B of K and of V per chunk (row = (tid >> 4) + 16 p, 16-B column = tid & 15)
  const int srow = threadIdx.x >> 4, sc16 = threadIdx.x & 15;
  // staging registers as named scalars + macros: arrays that are written under a condition in the loop end up in scratch memory
  uint4 k0r, k1r, k2r, k3r, k4r, k5r, k6r, k7r, v0r, v1r, v2r, v3r, v4r, v5r, v6r, v7r;
#define ATT2_G1(kk, vv, p, ch)                                                              \
  {                                                                                         \
    const int key_ = min(key0 + (ch) * ATT2_CHUNK + srow + 16 * (p), last_key);             \
    kk = att_ld_kv(reinterpret_cast<const uint4*>(Kh + (size_t)key_ * 128 + sc16 * 8));     \
    vv = att_ld_kv(reinterpret_cast<const uint4*>(Vh + (size_t)key_ * 128 + sc16 * 8));     \
  }
#define ATT2_GLOAD(ch)                                                                      \
  ATT2_G1(k0r, v0r, 0, ch) ATT2_G1(k1r, v1r, 1, ch) ATT2_G1(k2r, v2r, 2, ch) ATT2_G1(k3r, v3r, 3, ch) \
  ATT2_G1(k4r, v4r, 4, ch) ATT2_G1(k5r, v5r, 5, ch) ATT2_G1(k6r, v6r, 6, ch) ATT2_G1(k7r, v7r, 7, ch)
#define ATT2_W1(kk, vv, p)                                                                  \
  *reinterpret_cast<uint4*>(sK + att_swz(srow + 16 * (p), sc16 * 16)) = kk;                 \
  *reinterpret_cast<uint4*>(sV + att_vswz(srow + 16 * (p), sc16 * 16)) = vv;
#define ATT2_LWRITE()                                                                       \
  ATT2_W1(k0r, v0r, 0) ATT2_W1(k1r, v1r, 1) ATT2_W1(k2r, v2r, 2) ATT2_W1(k3r, v3r, 3)       \
  ATT2_W1(k4r, v4r, 4) ATT2_W1(k5r, v5r, 5) ATT2_W1(k6r, v6r, 6) ATT2_W1(k7r, v7r, 7)
  ATT2_GLOAD(0)
  ATT2_LWRITE()
  __syncthreads();
  // per-lane constants of the transposing V reads: 16-lane group g reads the [4 keys][16 cols] block of keys +0..3, columns
  // 32 dt + 16 (g & 1) ..+15 ; lane i of the group supplies row (i >> 2), columns 4 (i & 3) .. +3
  const int li = lane & 15, lg = lane >> 4;
  const int vrow_l = wave * 32 + 4 * hi + (li >> 2);
  const int vcol_l = (16 * (lg & 1) + 4 * (li & 3)) * 2;
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk) { ATT2_GLOAD(ch + 1) }  // in flight during the MFMA work below
    const int kbase = key0 + ch * ATT2_CHUNK + wave * 32;
    if (kbase < key_end) {
      f32x16 S;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
      const int krow = wave * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(sK + att_swz(krow, ks * 32 + hi * 16));
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(qf[ks]), S, 0, 0, 0);
      }
      float mx = NEG_INF;
      // eager scores: bf16(bf16(S) / sqrt(hd)) through one FMA-corrected reciprocal multiply (bit-identical to the division for
      // every bf16 input, tools/div_check.hip)
      if (kbase + 32 <= n_prefix) {  // wave-uniform: the whole tile lies in the committed context, every key visible
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sc;
          if (EAGER) {
            const float xb = rdbf(S[r]);
            float q = xb * rsqrt_hd;
            q = __builtin_fmaf(__builtin_fmaf(-q, sqrt_hd, xb), rsqrt_hd, q);
            sc = rdbf(q);
          } else {
            sc = S[r] * scale;
          }
          S[r] = sc;
          mx = fmaxf(mx, sc);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * hi;
          float sc;
          if (EAGER) {
            const float xb = rdbf(S[r]);
            float q = xb * rsqrt_hd;
            q = __builtin_fmaf(__builtin_fmaf(-q, sqrt_hd, xb), rsqrt_hd, q);
            sc = rdbf(q);
          } else {
            sc = S[r] * scale;
          }
          bool vis = key < n_prefix;
          if (!vis && key < n_total) vis = (mbits >> (key - n_prefix)) & 1ull;
          sc = vis ? sc : NEG_INF;
          S[r] = sc;
          mx = fmaxf(mx, sc);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_run == NEG_INF) ? 0.f : __expf(m_run - m_new);
      float psum = 0.f;
      unsigned pb[8];
      const float m_sub = (m_new == NEG_INF) ? 0.f : m_new;  // a row with nothing visible yet: exp(-inf - 0) = 0, never inf - inf
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __expf(S[r] - m_sub);
        const float p1 = __expf(S[r + 1] - m_sub);
        psum += p0 + p1;
        pb[r >> 1] = pack2(p0, p1);
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {  // the running maximum settles after the first tiles: most tiles rescale nothing
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
      }
      // O^T[d][q] += V^T[d][key] P^T[key][q]: k-slot (hi, t) of 16-key step kk is key 16 kk + 8 (t >> 2) + 4 hi + (t & 3) for BOTH
      // operands — P already sits that way in the accumulator layout, V^T is fetched to match by two transposing reads
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint4 pB = make_uint4(pb[kk * 4 + 0], pb[kk * 4 + 1], pb[kk * 4 + 2], pb[kk * 4 + 3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int r0 = vrow_l + 16 * kk;
          const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(sV + att_vswz(r0, 64 * dt + vcol_l)));
          const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(sV + att_vswz(r0 + 8, 64 * dt + vcol_l)));
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          const s16x8_t va = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
          O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&va), as_bf16x8(pB), O[dt], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // every wave is done with this chunk's LDS image
    if (ch + 1 < nchunk) {
      ATT2_LWRITE()
      __syncthreads();
    }
  }
#undef ATT2_GLOAD
#undef ATT2_LWRITE
#undef ATT2_G1
#undef ATT2_W1
  // ---- merge the four key-tile waves through LDS (the K/V image is dead: the loop ended with a barrier) ----
  float* sM = reinterpret_cast<float*>(smem + 2 * ATT2_CHUNK * 256);  // [4][32] m, then [4][32] l
  float* sO = reinterpret_cast<float*>(smem);                          // [4 waves][128 hd][32 q] fp32
  if (hi == 0) sM[wave * 32 + j] = m_run;
  __syncthreads();
  const float m_all = fmaxf(fmaxf(sM[j], sM[32 + j]), fmaxf(sM[64 + j], sM[96 + j]));
  const float f = (m_run == NEG_INF) ? 0.f : __expf(m_run - m_all);
  if (hi == 0) sM[128 + wave * 32 + j] = l_run * f;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int drow = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * hi;
      sO[wave * 4096 + drow * 32 + j] = O[dt][r] * f;
    }
  __syncthreads();
  const size_t pidx = ((size_t)(kvh * NQT + qt) * nsplit + split);
  float4* po = reinterpret_cast<float4*>(R.part_o + pidx * (128 * 32));
  const float4* s4 = reinterpret_cast<const float4*>(sO);
  const bool fused = R.cnt != nullptr;  // fused merge: the partial is published with write-through (sc1) stores — see below
#pragma unroll
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const float4 a = s4[e], b = s4[1024 + e], c = s4[2048 + e], d = s4[3072 + e];
    const float4 v = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    if (fused) {
      const f32x4 vv = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(po + e), "v"(vv) : "memory");
    } else {
      po[e] = v;
    }
  }
  if (threadIdx.x < 32) {
    const int q = threadIdx.x;
    const float pm = fmaxf(fmaxf(sM[q], sM[32 + q]), fmaxf(sM[64 + q], sM[96 + q]));
    const float pl = (sM[128 + q] + sM[160 + q]) + (sM[192 + q] + sM[224 + q]);
    if (fused) {
      __hip_atomic_store(&R.part_ml[pidx * 64 + q], pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (4-byte agent-scope stores are sc1 stores)
      __hip_atomic_store(&R.part_ml[pidx * 64 + 32 + q], pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      R.part_ml[pidx * 64 + q] = pm;
      R.part_ml[pidx * 64 + 32 + q] = pl;
    }
  }
  // ---- fused merge (round 5; OFF by default — measured slower, see launch_attention_n): the LAST workgroup to arrive for this (request, kv head, q-tile) merges the key splits itself — the arithmetic
  // of tree_attn_reduce_kernel, element for element — instead of a separate launch (4.6 us alone, 15-25 us when it queues behind other lanes'
  // GEMMs: profiles/r05_kernel_stats_4lanes_cohort8_first.csv).  Publish / acquire as MI355X_MICROARCH.md §inter-workgroup visibility
  // prescribes for tens of KB per workgroup: write-through (sc1) stores -> every wave drains vmcnt -> barrier -> relaxed agent ticket; the
  // last arriver: agent acquire -> barrier -> plain loads.  (First form: plain stores + a lane-0 agent RELEASE fence per workgroup — the L2
  // write-back of 16 KB of fresh partial per workgroup, 1 792 times a launch: 3252 -> 2556 tok/s, profiles/r05_ab_attention_fused_merge.txt.)
  // The ticket counter returns to zero with the last arriver (graph replays start from zero).
  WGCLK_END(EAGER ? 20 : 21, args.r[0].part_o);
  if (!fused) return;
  const int ns_act = max(1, min(nsplit, (n_total + keys_per_wg - 1) / keys_per_wg));  // workgroups that did not take the early exit above
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(&R.cnt[kvh * NQT + qt], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == ns_act - 1;
    if (last) {
      __hip_atomic_store(&R.cnt[kvh * NQT + qt], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  float* sh = reinterpret_cast<float*>(smem);  // (the K / V image and the wave-merge scratch are dead)
  for (int dpart = 0; dpart < 4; ++dpart) {
    tree_attn_reduce_body(R, sh, head, qt % MT, dpart, H, H_kv, M, tail, keys_per_wg, nsplit, ldo);
    __syncthreads();
  }
}

// ---- prefill attention (round 3): causal attention of the prompt's L query rows over the K/V rows [0, L) the prefill has just written ---
// The target prefill's GEMMs stay library calls, its attention no longer does: torch's flash SDPA runs this shape (L = 2704, 32 heads
// of 128) at 200 TFLOP/s (296 us per layer, a quarter of the prefill), and it is not the reference's arithmetic for LLaVA (eager:
// bf16 scores, fp32 softmax — modeling_llama_kv.py:602-623).  This kernel is the decode kernel's tile arithmetic (same MFMA operand
// layouts, same eager score rounding, same bf16 P, K image XOR-swizzled for ds_read_b128, V through ds_read_b64_tr_b16) turned
// around for many query rows: a workgroup = 128 query rows of one head, each of its 4 waves OWNS 32 of them (Q fragments, running
// max / sum and the 32 x 128 output tile in registers for the whole key loop), the workgroup stages 128-key chunks of K and V once
// for all four waves, every wave walks the chunk's four 32-key tiles with the online softmax — so there is no cross-wave merge and
// no partial tile: the output row is normalised and stored by the wave that owns it.  Causal structure: key k is visible to row r
// iff k <= r; a tile entirely below the diagonal takes the unmasked path, tiles above it are skipped, chunks above the
// workgroup's last row are never staged.  grid (ceil(ceil(L / 128) / 2), H): a workgroup takes a long and a short row block (see the kernel).
// Two scores at a time (the prefill kernel is VALU-bound: ~20 vector instructions per MFMA in the decode kernel's per-element form):
// bf16 rounding of a pair is ONE v_cvt_pk_bf16_f32 + two unpack ops, the scaling runs on packed-fp32 instructions.
typedef float f32x2_pk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_pk rdbf2(f32x2_pk v) {
  const unsigned u = pack2(v.x, v.y);
  f32x2_pk r;
  r.x = __uint_as_float(u << 16);
  r.y = __uint_as_float(u & 0xffff0000u);
  return r;
}
template <bool EAGER>
__device__ __forceinline__ f32x2_pk score_pair(float s0, float s1) {
  const float sqrt_hd = 11.313708498984761f, rsqrt_hd = 1.0f / 11.313708498984761f;
  f32x2_pk x = {s0, s1};
  if (EAGER) {  // bf16(bf16(S) / sqrt(hd)): FMA-corrected reciprocal multiply, bit-identical to the division (tools/div_check.hip)
    const f32x2_pk xb = rdbf2(x);
    f32x2_pk q = xb * rsqrt_hd;
    const f32x2_pk e = __builtin_elementwise_fma(-q, (f32x2_pk){sqrt_hd, sqrt_hd}, xb);
    q = __builtin_elementwise_fma(e, (f32x2_pk){rsqrt_hd, rsqrt_hd}, q);
    return rdbf2(q);
  }
  return x * 0.08838834764831845f;
}
template <bool EAGER>
__global__ __launch_bounds__(256, 2) void prefill_attn_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ Kc,
                                                              const bf16_t* __restrict__ Vc, int s_max, int H, int H_kv, int L,
                                                              bf16_t* __restrict__ out, int ldo, int paired) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WGCLK_BEGIN();
  unsigned char* sK = smem;
  unsigned char* sV = smem + ATT2_CHUNK * 256;
  const int head = blockIdx.y, kvh = head / (H / H_kv);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, hi = lane >> 5;
  // Row block b of a causal prompt walks b + 1 key chunks: a workgroup takes the PAIR (NB-1-x, x) — long block first — so that every
  // workgroup walks NB + 1 chunks (one block per workgroup made the last row blocks the critical path: 185 us per layer at L = 2704,
  // the chip half idle behind 22-chunk workgroups)
  const int NB = (L + 127) / 128;
  // (short prompts keep one block per workgroup, longest first: pairing would leave CUs without a workgroup)
  const int qb_long = NB - 1 - blockIdx.x, qb_short = paired ? (int)blockIdx.x : qb_long;
  for (int pass = 0; pass < (qb_long != qb_short ? 2 : 1); ++pass) {
  const int qb = pass == 0 ? qb_long : qb_short;
  const int row0 = qb * 128 + wave * 32, mrow = row0 + j;  // this wave's query rows / this lane's
  const bool qvalid = mrow < L;
  const bf16_t* Kh = Kc + (size_t)kvh * s_max * 128;
  const bf16_t* Vh = Vc + (size_t)kvh * s_max * 128;
  const int n_keys = min(L, qb * 128 + 128);  // keys any row of this workgroup can see
  const int nchunk = (n_keys + ATT2_CHUNK - 1) / ATT2_CHUNK;
  const int last_key = n_keys - 1;
  uint4 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = *reinterpret_cast<const uint4*>(Q + (size_t)(qvalid ? mrow : L - 1) * ldq + head * 128 + ks * 16 + hi * 8);
  float m_run = NEG_INF, l_run = 0.f;
  f32x16 O[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  const int srow = threadIdx.x >> 4, sc16 = threadIdx.x & 15;
  uint4 k0r, k1r, k2r, k3r, k4r, k5r, k6r, k7r, v0r, v1r, v2r, v3r, v4r, v5r, v6r, v7r;
#define PFA_G1(kk, vv, p, ch)                                                               \
  {                                                                                         \
    const int key_ = min((ch) * ATT2_CHUNK + srow + 16 * (p), last_key);                    \
    kk = *reinterpret_cast<const uint4*>(Kh + (size_t)key_ * 128 + sc16 * 8);               \
    vv = *reinterpret_cast<const uint4*>(Vh + (size_t)key_ * 128 + sc16 * 8);               \
  }
#define PFA_GLOAD(ch)                                                                       \
  PFA_G1(k0r, v0r, 0, ch) PFA_G1(k1r, v1r, 1, ch) PFA_G1(k2r, v2r, 2, ch) PFA_G1(k3r, v3r, 3, ch) \
  PFA_G1(k4r, v4r, 4, ch) PFA_G1(k5r, v5r, 5, ch) PFA_G1(k6r, v6r, 6, ch) PFA_G1(k7r, v7r, 7, ch)
#define PFA_W1(kk, vv, p)                                                                   \
  *reinterpret_cast<uint4*>(sK + att_swz(srow + 16 * (p), sc16 * 16)) = kk;                 \
  *reinterpret_cast<uint4*>(sV + att_vswz(srow + 16 * (p), sc16 * 16)) = vv;
#define PFA_LWRITE()                                                                        \
  PFA_W1(k0r, v0r, 0) PFA_W1(k1r, v1r, 1) PFA_W1(k2r, v2r, 2) PFA_W1(k3r, v3r, 3)           \
  PFA_W1(k4r, v4r, 4) PFA_W1(k5r, v5r, 5) PFA_W1(k6r, v6r, 6) PFA_W1(k7r, v7r, 7)
  PFA_GLOAD(0)
  PFA_LWRITE()
  __syncthreads();
  const int li = lane & 15, lg = lane >> 4;
  const int vrow_l = 4 * hi + (li >> 2);
  const int vcol_l = (16 * (lg & 1) + 4 * (li & 3)) * 2;
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk) { PFA_GLOAD(ch + 1) }  // in flight during the MFMA work below
    for (int kt = 0; kt < 4; ++kt) {
      const int kbase = ch * ATT2_CHUNK + kt * 32;
      if (kbase > row0 + 31 || kbase >= L) break;  // wave-uniform: the tile lies above this wave's last row
      f32x16 S;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
      const int krow = kt * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(sK + att_swz(krow, ks * 32 + hi * 16));
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(qf[ks]), S, 0, 0, 0);
      }
      float mx = NEG_INF;
      // eager scores: bf16(bf16(S) / sqrt(hd)) through one FMA-corrected reciprocal multiply (bit-identical to the division for
      // every bf16 input, tools/div_check.hip) — the decode kernel's arithmetic
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_pk sc = score_pair<EAGER>(S[r], S[r + 1]);
        S[r] = sc.x;
        S[r + 1] = sc.y;
      }
      if (kbase + 31 > row0) {  // wave-uniform: the tile crosses the diagonal (else every key is visible to every row of the wave)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * hi;
          S[r] = (key <= mrow) ? S[r] : NEG_INF;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_run == NEG_INF) ? 0.f : __expf(m_run - m_new);
      unsigned pb[8];
      const float m_sub = (m_new == NEG_INF) ? 0.f : m_new;
      // exp(s - m) = exp2(s log2e - m log2e): one packed FMA per two scores, then the two v_exp_f32
      const float kLog2e = 1.4426950408889634f;
      const f32x2_pk l2 = {kLog2e, kLog2e}, mneg = {-m_sub * kLog2e, -m_sub * kLog2e};
      f32x2_pk ps = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_pk t = __builtin_elementwise_fma((f32x2_pk){S[r], S[r + 1]}, l2, mneg);
        f32x2_pk pe;
        pe.x = __builtin_amdgcn_exp2f(t.x);
        pe.y = __builtin_amdgcn_exp2f(t.y);
        ps += pe;
        pb[r >> 1] = pack2(pe.x, pe.y);
      }
      float psum = ps.x + ps.y;
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint4 pB = make_uint4(pb[kk * 4 + 0], pb[kk * 4 + 1], pb[kk * 4 + 2], pb[kk * 4 + 3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int r0 = kt * 32 + vrow_l + 16 * kk;
          const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(sV + att_vswz(r0, 64 * dt + vcol_l)));
          const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(sV + att_vswz(r0 + 8, 64 * dt + vcol_l)));
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          const s16x8_t va = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
          O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&va), as_bf16x8(pB), O[dt], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // every wave is done with this chunk's LDS image
    if (ch + 1 < nchunk) {
      PFA_LWRITE()
      __syncthreads();
    }
  }
#undef PFA_GLOAD
#undef PFA_LWRITE
#undef PFA_G1
#undef PFA_W1
  if (qvalid) {
  // O^T[d][q] of this wave's 32 rows: lane (j, hi) holds, for its row, d = 32 dt + 8 g + 4 hi + (0..3) in O[dt][4 g .. 4 g + 3]
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  bf16_t* orow = out + (size_t)mrow * ldo + head * 128;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<uint2*>(orow + 32 * dt + 8 * g + 4 * hi) =
          make_uint2(pack2(O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv), pack2(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv));
  }
  __syncthreads();  // the next pass re-stages the LDS image
  }
  WGCLK_END(30, out);
}

// merge partials over splits — the body: (head, m-tile) of request R, d-rows [32 dpart, +32); 256 threads; `sh` = 10.5 KB of LDS scratch
// (wgt[64][32], smax[8][32], ssum[8][32], linv[32] floats).  One arithmetic for the stand-alone kernel and for the fused merge below.
#define ATT_RED_LDS_FLOATS (64 * 32 + 8 * 32 + 8 * 32 + 32)
__device__ __forceinline__ void tree_attn_reduce_body(const AttnReq& R, float* __restrict__ sh, int head, int mt, int dpart, int H, int H_kv, int M,
                                                      int tail, int keys_per_wg, int nsplit, int ldo) {
  const float* __restrict__ part_o = R.part_o;
  const float* __restrict__ part_ml = R.part_ml;
  const int* __restrict__ prefix_dev = R.prefix_dev;
  bf16_t* __restrict__ out = R.out;
  // Latency-bound (a few hundred KB per launch): every load of a phase is issued before the first use, with clamped
  // (always valid) addresses instead of guards, so the dependent chain is prefix -> {m,l and the first 16 partial tiles} -> out.
  float(*wgt)[32] = reinterpret_cast<float(*)[32]>(sh);
  float(*smax)[32] = reinterpret_cast<float(*)[32]>(sh + 64 * 32);
  float(*ssum)[32] = reinterpret_cast<float(*)[32]>(sh + 64 * 32 + 8 * 32);
  float* linv = sh + 64 * 32 + 16 * 32;
  const int G = H / H_kv, MT = (M + 31) >> 5, NQT = G * MT;
  const int kvh = head / G, qt = (head % G) * MT + mt;
  const int n_total = (prefix_dev ? *prefix_dev : 0) + tail;
  const int ns = max(1, min(nsplit, (n_total + keys_per_wg - 1) / keys_per_wg));
  const size_t base = (size_t)(kvh * NQT + qt) * nsplit;
  const int q = threadIdx.x & 31, s8 = threadIdx.x >> 5;
  const int e4 = dpart * 1024 + threadIdx.x * 4;  // 4 consecutive elements of the [128 d][32 q] tile: one d, queries q4..q4+3
  float mv[8], lv[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int sc = min(s8 + 8 * t, ns - 1);
    mv[t] = part_ml[(base + sc) * 64 + q];
    lv[t] = part_ml[(base + sc) * 64 + 32 + q];
  }
  float4 po[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) po[t] = ld_partial(reinterpret_cast<const float4*>(part_o + (base + min(t, ns - 1)) * (128 * 32) + e4));
  float mm = NEG_INF;
#pragma unroll
  for (int t = 0; t < 8; ++t) mm = fmaxf(mm, mv[t]);  // clamped duplicates do not change a max
  smax[s8][q] = mm;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) mm = fmaxf(mm, smax[t][q]);
  float L = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int sp = s8 + 8 * t;
    const float w = (sp < ns && mv[t] != NEG_INF) ? __expf(mv[t] - mm) : 0.f;
    wgt[sp][q] = w;  // rows >= ns get weight 0: the clamped partial loads above contribute nothing
    L += w * lv[t];
  }
  ssum[s8][q] = L;
  __syncthreads();
  if (s8 == 0) {
    float Lt = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) Lt += ssum[t][q];
    linv[q] = (Lt > 0.f) ? 1.0f / Lt : 0.f;
  }
  __syncthreads();
  const int d = e4 >> 5, q4 = e4 & 31;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 w = *reinterpret_cast<const float4*>(&wgt[t][q4]);
    acc.x += w.x * po[t].x; acc.y += w.y * po[t].y; acc.z += w.z * po[t].z; acc.w += w.w * po[t].w;
  }
  for (int sp = 16; sp < ns; ++sp) {  // long contexts only (> 16 key splits)
    const float4 w = *reinterpret_cast<const float4*>(&wgt[sp][q4]);
    const float4 v = *reinterpret_cast<const float4*>(part_o + (base + sp) * (128 * 32) + e4);
    acc.x += w.x * v.x; acc.y += w.y * v.y; acc.z += w.z * v.z; acc.w += w.w * v.w;
  }
  const float4 li = *reinterpret_cast<const float4*>(&linv[q4]);
  const float o[4] = {acc.x * li.x, acc.y * li.y, acc.z * li.z, acc.w * li.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mt * 32 + q4 + i;
    if (m < M) out[(size_t)m * ldo + head * 128 + d] = f2bf(o[i]);
  }
}

// merge partials over splits, stand-alone: grid (H*MT, 4, requests), 256 threads
__global__ __launch_bounds__(256) void tree_attn_reduce_kernel(AttnArgs args, int H, int H_kv, int M, int tail,
                                                               int keys_per_wg, int nsplit, int ldo) {
  // blockIdx.z = request of a cohort (0 otherwise)
  const AttnReq R = attn_req(args, blockIdx.z);
  __shared__ float sh[ATT_RED_LDS_FLOATS];
  const int MT = (M + 31) >> 5;
  WGCLK_BEGIN();
  tree_attn_reduce_body(R, sh, blockIdx.x / MT, blockIdx.x % MT, blockIdx.y, H, H_kv, M, tail, keys_per_wg, nsplit, ldo);
  WGCLK_END(22, args.r[0].part_o);
}

// x <- bf16(x + r) (the residual add of the decoder layer, modeling_llama_kv.py:742-756) and y = RMSNorm(x) * w in ONE pass over the row:
// the prefill's `x = x + o_proj(...)` / `x = x + down_proj(...)` followed by the next norm (two torch launches + one of ours before).
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(bf16_t* __restrict__ X, const bf16_t* __restrict__ R, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ Y, int D, float eps) {
  __shared__ float part[4];
  bf16_t* x = X + (size_t)blockIdx.x * D;
  const bf16_t* r = R + (size_t)blockIdx.x * D;
  bf16_t* y = Y + (size_t)blockIdx.x * D;
  float ss = 0.f;
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + d), rv = *reinterpret_cast<const uint4*>(r + d);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    const bf16_t* re = reinterpret_cast<const bf16_t*>(&rv);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = rdbf(bf2f(e[j]) + bf2f(re[j]));
      ss += f[j] * f[j];
    }
    *reinterpret_cast<uint4*>(x + d) = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = (part[0] + part[1]) + (part[2] + part[3]);
  const float inv = 1.0f / sqrtf(tot / (float)D + eps);
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {  // (every thread re-reads exactly the elements it has just written)
    const uint4 v = *reinterpret_cast<const uint4*>(x + d), wv = *reinterpret_cast<const uint4*>(w + d);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    const bf16_t* we = reinterpret_cast<const bf16_t*>(&wv);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f(we[j]) * rdbf(bf2f(e[j]) * inv);
    *reinterpret_cast<uint4*>(y + d) = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
  }
}

// ------------------------------------------------------------------------------------------------
// Row gathers / small elementwise helpers
// ------------------------------------------------------------------------------------------------
// out[i, col0 : col0+D] = table[idx(i)] ; idx from a device int array (+ optional device base offset)
__global__ void gather_rows_kernel(const bf16_t* __restrict__ table, int ld_t, const int* __restrict__ idx, int idx_off,
                                   const int* __restrict__ idx_base_dev, bf16_t* __restrict__ out, int ld_o, int D) {
  const int i = blockIdx.x;
  int r = idx ? idx[(idx_base_dev ? *idx_base_dev : 0) + idx_off + i] : (idx_base_dev ? *idx_base_dev : 0) + idx_off + i;
  const bf16_t* src = table + (size_t)r * ld_t;
  bf16_t* dst = out + (size_t)i * ld_o;
  for (int d = threadIdx.x * 8; d < D; d += blockDim.x * 8)
    *reinterpret_cast<uint4*>(dst + d) = *reinterpret_cast<const uint4*>(src + d);
}
// out[i, :] = vec (broadcast one row) ; used for the global image feature g (cnets_ours.py:984)
__device__ __forceinline__ void bcast_row_body(const bf16_t* __restrict__ vec, bf16_t* __restrict__ out, int ld_o, int D) {
  bf16_t* dst = out + (size_t)blockIdx.x * ld_o;
  for (int d = threadIdx.x * 8; d < D; d += blockDim.x * 8)
    *reinterpret_cast<uint4*>(dst + d) = *reinterpret_cast<const uint4*>(vec + d);
}
__global__ void bcast_row_kernel(const bf16_t* __restrict__ vec, bf16_t* __restrict__ out, int ld_o, int D) { bcast_row_body(vec, out, ld_o, D); }
struct bcast_row_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { bcast_row_body(a...); }
};

// ------------------------------------------------------------------------------------------------
// Row-wise argmax (first max wins) and log-softmax + top-k (value desc, index asc) over bf16 logits
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ void argmax_rows_body(const bf16_t* __restrict__ logits, int ld, int V,
                                                           int* __restrict__ out) {
  // one workgroup per row; a row of 32 064 logits is ONE pass of the block with four 16-byte loads per thread issued back to back
  // (the 256-thread form walked 16 dependent iterations: 13 us per launch)
  __shared__ float sv[16];
  __shared__ int si[16];
  const bf16_t* x = logits + (size_t)blockIdx.x * ld;
  float bv = NEG_INF;
  int bi = 0x7fffffff;
  for (int base = 0; base < V; base += 1024 * 8 * 4) {
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = base + (i * 1024 + threadIdx.x) * 8;
      v[i] = d < V ? *reinterpret_cast<const uint4*>(x + d) : make_uint4(0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = base + (i * 1024 + threadIdx.x) * 8;
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (d + j < V) {
          const float f = bf2f(e[j]);
          if (better(f, d + j, bv, bi)) { bv = f; bi = d + j; }
        }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o);
    int oi = __shfl_xor(bi, o);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    out[blockIdx.x] = bi;
  }
}
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const bf16_t* __restrict__ logits, int ld, int V,
                                                           int* __restrict__ out) { argmax_rows_body(logits, ld, V, out); }
struct argmax_rows_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { argmax_rows_body(a...); }
};

#define TOPK_MAX 16
#define LSTK_CHUNKS 64
// log-softmax + top-k of bf16 logit rows, three small multi-workgroup passes (a single workgroup per 32k-wide row took 255 us):
//   1. per-chunk max / sum-exp                         grid (rows, LSTK_CHUNKS)
//   2. per-chunk top-k under the exact key (bf16(x - lse) desc, index asc)   grid (rows, LSTK_CHUNKS)
//   3. merge of the LSTK_CHUNKS*k candidates            grid (rows)
__device__ __forceinline__ void lstk_chunk_range(int V, int chunk, int& lo, int& hi) {
  const int per = ((V + LSTK_CHUNKS - 1) / LSTK_CHUNKS + 7) & ~7;
  lo = chunk * per;
  hi = min(V, lo + per);
}

// ---- one-launch form: ONE workgroup of 1024 threads per row holds the whole row in registers (8*NV logits per thread).
//   1. block max and sum-exp -> lse (fp32, the oracle's formula: m + log(sum exp(x - m)))
//   2. keys: 64-bit (order-preserving image of bf16(x - lse), ~index): a larger key is a better candidate (value desc, index asc)
//   3. threshold: the k-th largest of the 1024 per-LANE maxima (k rounds of shuffle-only wave arg-max per wave, then over the 16*k
//      survivors).  At most k lanes own an element >= it, so at most k * 8*NV elements of the row pass the threshold.
//   4. those candidates are appended to an LDS list (capacity 16*8*NV... never exceeded, see 3.) and one wave picks the k best.
// Exactly the selection of the three-pass form (same key), 4-5 us instead of 36 us per call at V = 32 064, k = 8.
__device__ __forceinline__ unsigned long long lstk_key(float v, int idx) {
  const unsigned u = __float_as_uint(v);
  const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}
__device__ __forceinline__ float lstk_key_value(unsigned long long key) {
  const unsigned ord = (unsigned)(key >> 32);
  return __uint_as_float((ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord);
}
__device__ __forceinline__ int lstk_key_index(unsigned long long key) { return (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)); }
// Wave-wide reductions on the DPP path (row-local quad permutes / mirrors, then row broadcasts; the result is read from lane 63):
// a __shfl_xor ladder is six dependent ds_bpermute round trips (~700 cycles), this is six VALU steps.  `old` = the identity of the
// operation, so lanes a row-masked step does not write contribute nothing.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v, unsigned ident) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define VS_STEP(CTRL, RM)                                                                                            \
  {                                                                                                                  \
    const unsigned lo_ = dpp_u32<CTRL, RM>((unsigned)v, 0u), hi_ = dpp_u32<CTRL, RM>((unsigned)(v >> 32), 0u);      \
    const unsigned long long w_ = ((unsigned long long)hi_ << 32) | lo_;                                           \
    v = w_ > v ? w_ : v;                                                                                             \
  }
  VS_STEP(0xB1, 0xf) VS_STEP(0x4E, 0xf) VS_STEP(0x141, 0xf) VS_STEP(0x140, 0xf) VS_STEP(0x142, 0xa) VS_STEP(0x143, 0xc)
#undef VS_STEP
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
#define VS_STEP(CTRL, RM) v = fmaxf(v, __uint_as_float(dpp_u32<CTRL, RM>(__float_as_uint(v), 0xFF800000u)));
  VS_STEP(0xB1, 0xf) VS_STEP(0x4E, 0xf) VS_STEP(0x141, 0xf) VS_STEP(0x140, 0xf) VS_STEP(0x142, 0xa) VS_STEP(0x143, 0xc)
#undef VS_STEP
  return __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {  // fixed association: deterministic
#define VS_STEP(CTRL, RM) v += __uint_as_float(dpp_u32<CTRL, RM>(__float_as_uint(v), 0u));
  VS_STEP(0xB1, 0xf) VS_STEP(0x4E, 0xf) VS_STEP(0x141, 0xf) VS_STEP(0x140, 0xf) VS_STEP(0x142, 0xa) VS_STEP(0x143, 0xc)
#undef VS_STEP
  return __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
#define LSTK_ROW_CAND_MAX (TOPK_MAX * 8 * 20)
template <int NV>
__device__ __forceinline__ void lstk_row_body(const bf16_t* __restrict__ logits, int ld, int V, int k, int* __restrict__ out_idx,
                                                        float* __restrict__ out_logp) {
  __shared__ float s_red[16];
  __shared__ unsigned long long s_lmax[1024];
  __shared__ unsigned long long s_cand[LSTK_ROW_CAND_MAX];
  __shared__ unsigned long long s_thr;
  __shared__ int s_n;
  const bf16_t* x = logits + (size_t)blockIdx.x * ld;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  uint4 raw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e0 = (i * 1024 + tid) * 8;
    raw[i] = e0 < V ? *reinterpret_cast<const uint4*>(x + e0) : make_uint4(0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u);  // -inf
  }
  if (tid == 0) s_n = 0;
  // ---- 1. lse
  float mx = NEG_INF;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, bf2f(e[j]));
  }
  mx = wave_max_dpp(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = s_red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_red[w]);
  __syncthreads();
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) se += __expf(bf2f(e[j]) - mx);  // exp(-inf) = 0 for the padding
  }
  se = wave_sum_dpp(se);
  if (lane == 0) s_red[wave] = se;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += s_red[w];  // fixed order
  const float lse = mx + __logf(tot);
  // ---- 2. per-lane maximum key
  unsigned long long best = 0ull;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
    const int e0 = (i * 1024 + tid) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned long long key = (e0 + j < V) ? lstk_key(rdbf(bf2f(e[j]) - lse), e0 + j) : 0ull;
      best = key > best ? key : best;
    }
  }
  s_lmax[tid] = best;
  __syncthreads();
  // ---- 3. threshold = k-th largest of the 1024 lane maxima (one wave, 16 values per lane)
  if (wave == 0) {
    unsigned long long c[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) c[q] = s_lmax[lane + 64 * q];
    unsigned long long thr = 0ull;
    for (int sel = 0; sel < k; ++sel) {
      unsigned long long m = c[0];
#pragma unroll
      for (int q = 1; q < 16; ++q) m = c[q] > m ? c[q] : m;
      thr = wave_max_u64(m);
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (c[q] == thr) c[q] = 0ull;  // keys are distinct (they carry the index); 0 = removed / padding
    }
    if (lane == 0) s_thr = thr;
  }
  __syncthreads();
  // ---- 4. candidates: every element whose key reaches the threshold (only lanes whose maximum does can own one)
  const unsigned long long thr = s_thr;
  if (best >= thr && best != 0ull) {  // (thr == 0: fewer than k lanes hold data, i.e. a row of < 8k elements — everything is a candidate)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
      const int e0 = (i * 1024 + tid) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned long long key = (e0 + j < V) ? lstk_key(rdbf(bf2f(e[j]) - lse), e0 + j) : 0ull;
        if (key >= thr && key != 0ull) {
          const int slot = atomicAdd(&s_n, 1);
          if (slot < LSTK_ROW_CAND_MAX) s_cand[slot] = key;
        }
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int n = min(s_n, LSTK_ROW_CAND_MAX);
    for (int sel = 0; sel < k; ++sel) {
      unsigned long long m = 0ull;
      for (int c = lane; c < n; c += 64) {
        const unsigned long long key = s_cand[c];
        m = key > m ? key : m;
      }
      const unsigned long long w = wave_max_u64(m);
      for (int c = lane; c < n; c += 64)
        if (s_cand[c] == w) s_cand[c] = 0ull;
      if (lane == 0) {
        out_idx[blockIdx.x * k + sel] = lstk_key_index(w);
        out_logp[blockIdx.x * k + sel] = lstk_key_value(w);
      }
    }
  }
}
template <int NV>
__global__ __launch_bounds__(1024) void lstk_row_kernel(const bf16_t* __restrict__ logits, int ld, int V, int k, int* __restrict__ out_idx,
                                                        float* __restrict__ out_logp) { lstk_row_body<NV>(logits, ld, V, k, out_idx, out_logp); }
template <int NV> struct lstk_row_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { lstk_row_body<NV>(a...); }
};

// ---- chunked form for LARGE vocabularies (Qwen2.5-VL: V = 152 064): the one-workgroup row kernel above walks 300 KB per row on ONE CU
// (62-73 us per call); here a row is cut into C chunks of <= 32 768 logits, each handled by its own 1024-thread workgroup with the row
// kernel's in-register technique, in three short launches: per-chunk (max, sum-exp) -> per-chunk top-k keys under the GLOBAL lse (the
// key is the ROUNDED log-prob, so the lse must be known before anything can be ranked) -> merge of the C*k survivors.
template <int NV>
__device__ __forceinline__ void lstk2_load(const bf16_t* x, int lo, int hi, uint4 (&raw)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e0 = lo + (i * 1024 + (int)threadIdx.x) * 8;
    raw[i] = e0 < hi ? *reinterpret_cast<const uint4*>(x + e0) : make_uint4(0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u);  // -inf
  }
}
template <int NV>
__device__ __forceinline__ void lstk2_stats_body(const bf16_t* __restrict__ logits, int ld, int V, int chunk, float* __restrict__ stats) {
  __shared__ float s_red[16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, C = gridDim.y, c = blockIdx.y;
  const int lo = c * chunk, hi = min(V, lo + chunk);  // chunk and V are multiples of 8
  uint4 raw[NV];
  lstk2_load<NV>(logits + (size_t)blockIdx.x * ld, lo, hi, raw);
  float mx = NEG_INF;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, bf2f(e[j]));
  }
  mx = wave_max_dpp(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = s_red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_red[w]);
  __syncthreads();
  float se = 0.f;
  if (mx != NEG_INF) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) se += __expf(bf2f(e[j]) - mx);
    }
  }
  se = wave_sum_dpp(se);
  if (lane == 0) s_red[wave] = se;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += s_red[w];  // fixed order
    stats[((size_t)blockIdx.x * C + c) * 2 + 0] = mx;
    stats[((size_t)blockIdx.x * C + c) * 2 + 1] = tot;
  }
}
template <int NV>
__global__ __launch_bounds__(1024) void lstk2_stats_kernel(const bf16_t* __restrict__ logits, int ld, int V, int chunk, float* __restrict__ stats) { lstk2_stats_body<NV>(logits, ld, V, chunk, stats); }
template <int NV> struct lstk2_stats_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { lstk2_stats_body<NV>(a...); }
};
template <int NV>
__device__ __forceinline__ void lstk2_select_body(const bf16_t* __restrict__ logits, int ld, int V, int chunk, int k,
                                                            const float* __restrict__ stats, unsigned long long* __restrict__ cand) {
  __shared__ unsigned long long s_lmax[1024];
  __shared__ unsigned long long s_cand[TOPK_MAX * 8 * NV];
  __shared__ unsigned long long s_thr;
  __shared__ int s_n;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, C = gridDim.y, c = blockIdx.y;
  const int lo = c * chunk, hi = min(V, lo + chunk);
  uint4 raw[NV];
  lstk2_load<NV>(logits + (size_t)blockIdx.x * ld, lo, hi, raw);
  if (tid == 0) s_n = 0;
  // global lse from the C chunk statistics, the same fixed order in every workgroup of the row
  const float* st = stats + (size_t)blockIdx.x * C * 2;
  float mx = NEG_INF;
  for (int q = 0; q < C; ++q) mx = fmaxf(mx, st[2 * q]);
  float tot = 0.f;
  for (int q = 0; q < C; ++q) tot += st[2 * q + 1] * __expf(st[2 * q] - mx);  // an empty chunk contributes 0 * exp(-inf) = 0 * 0
  const float lse = mx + __logf(tot);
  unsigned long long best = 0ull;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
    const int e0 = lo + (i * 1024 + tid) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned long long key = (e0 + j < hi) ? lstk_key(rdbf(bf2f(e[j]) - lse), e0 + j) : 0ull;
      best = key > best ? key : best;
    }
  }
  s_lmax[tid] = best;
  __syncthreads();
  if (wave == 0) {  // threshold = k-th largest of the 1024 lane maxima
    unsigned long long cq[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) cq[q] = s_lmax[lane + 64 * q];
    unsigned long long thr = 0ull;
    for (int sel = 0; sel < k; ++sel) {
      unsigned long long m = cq[0];
#pragma unroll
      for (int q = 1; q < 16; ++q) m = cq[q] > m ? cq[q] : m;
      thr = wave_max_u64(m);
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (cq[q] == thr) cq[q] = 0ull;
    }
    if (lane == 0) s_thr = thr;
  }
  __syncthreads();
  const unsigned long long thr = s_thr;
  if (best >= thr && best != 0ull) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw[i]);
      const int e0 = lo + (i * 1024 + tid) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned long long key = (e0 + j < hi) ? lstk_key(rdbf(bf2f(e[j]) - lse), e0 + j) : 0ull;
        if (key >= thr && key != 0ull) {
          const int slot = atomicAdd(&s_n, 1);
          if (slot < TOPK_MAX * 8 * NV) s_cand[slot] = key;
        }
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int n = min(s_n, TOPK_MAX * 8 * NV);
    for (int sel = 0; sel < k; ++sel) {
      unsigned long long m = 0ull;
      for (int q = lane; q < n; q += 64) {
        const unsigned long long key = s_cand[q];
        m = key > m ? key : m;
      }
      const unsigned long long w = wave_max_u64(m);
      for (int q = lane; q < n; q += 64)
        if (s_cand[q] == w) s_cand[q] = 0ull;
      if (lane == 0) cand[((size_t)blockIdx.x * C + c) * TOPK_MAX + sel] = w;  // 0 = this chunk has fewer than k elements
    }
  }
}
template <int NV>
__global__ __launch_bounds__(1024) void lstk2_select_kernel(const bf16_t* __restrict__ logits, int ld, int V, int chunk, int k,
                                                            const float* __restrict__ stats, unsigned long long* __restrict__ cand) { lstk2_select_body<NV>(logits, ld, V, chunk, k, stats, cand); }
template <int NV> struct lstk2_select_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { lstk2_select_body<NV>(a...); }
};
__device__ __forceinline__ void lstk2_merge_body(const unsigned long long* __restrict__ cand, int C, int k, int* __restrict__ out_idx,
                                                         float* __restrict__ out_logp) {
  __shared__ unsigned long long s_c[64 * TOPK_MAX];
  const int lane = threadIdx.x, n = C * k;
  for (int q = lane; q < n; q += 64) s_c[q] = cand[((size_t)blockIdx.x * C + q / k) * TOPK_MAX + q % k];
  __syncthreads();
  for (int sel = 0; sel < k; ++sel) {
    unsigned long long m = 0ull;
    for (int q = lane; q < n; q += 64) m = s_c[q] > m ? s_c[q] : m;
    const unsigned long long w = wave_max_u64(m);
    for (int q = lane; q < n; q += 64)
      if (s_c[q] == w) s_c[q] = 0ull;
    if (lane == 0) {
      out_idx[blockIdx.x * k + sel] = lstk_key_index(w);
      out_logp[blockIdx.x * k + sel] = lstk_key_value(w);
    }
  }
}
__global__ __launch_bounds__(64) void lstk2_merge_kernel(const unsigned long long* __restrict__ cand, int C, int k, int* __restrict__ out_idx,
                                                         float* __restrict__ out_logp) { lstk2_merge_body(cand, C, k, out_idx, out_logp); }
struct lstk2_merge_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { lstk2_merge_body(a...); }
};

__global__ __launch_bounds__(256) void lstk_stats_kernel(const bf16_t* __restrict__ logits, int ld, int V, float* __restrict__ stats) {
  __shared__ float s_red[4];
  const bf16_t* x = logits + (size_t)blockIdx.x * ld;
  int lo, hi;
  lstk_chunk_range(V, blockIdx.y, lo, hi);
  const int tid = threadIdx.x;
  float mx = NEG_INF;
  for (int d = lo + tid; d < hi; d += 256) mx = fmaxf(mx, bf2f(x[d]));
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float se = 0.f;
  for (int d = lo + tid; d < hi; d += 256) se += __expf(bf2f(x[d]) - mx);
  se = wave_sum(se);
  if ((tid & 63) == 0) s_red[tid >> 6] = se;
  __syncthreads();
  if (tid == 0) {
    float* o = stats + ((size_t)blockIdx.x * LSTK_CHUNKS + blockIdx.y) * 2;
    o[0] = mx;
    o[1] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
}

__device__ __forceinline__ float lstk_lse(const float* __restrict__ st) {  // fixed-order combine of the chunk stats
  float mx = NEG_INF;
  for (int c = 0; c < LSTK_CHUNKS; ++c) mx = fmaxf(mx, st[2 * c]);
  float se = 0.f;
  for (int c = 0; c < LSTK_CHUNKS; ++c) se += (st[2 * c] == NEG_INF) ? 0.f : st[2 * c + 1] * __expf(st[2 * c] - mx);
  return mx + __logf(se);
}

// block-wide selection of the k best (value desc, index asc) among per-thread candidate pairs held in LDS
__device__ __forceinline__ void block_select_topk(float* s_v, int* s_i, int n_cand, int k, float* out_v, int* out_i) {
  __shared__ float w_v[4];
  __shared__ int w_i[4], w_s[4];
  const int tid = threadIdx.x;
  for (int sel = 0; sel < k; ++sel) {
    float bv = NEG_INF;
    int bi = 0x7fffffff, bs = -1;
    for (int c = tid; c < n_cand; c += 256) {
      const float v = s_v[c];
      const int ix = s_i[c];
      if (better(v, ix, bv, bi)) { bv = v; bi = ix; bs = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o), os = __shfl_xor(bs, o);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bs = os; }
    }
    if ((tid & 63) == 0) { w_v[tid >> 6] = bv; w_i[tid >> 6] = bi; w_s[tid >> 6] = bs; }
    __syncthreads();
    float gv = w_v[0];
    int gi = w_i[0], gs = w_s[0];
    for (int w = 1; w < 4; ++w)
      if (better(w_v[w], w_i[w], gv, gi)) { gv = w_v[w]; gi = w_i[w]; gs = w_s[w]; }
    if (tid == 0) {
      out_v[sel] = gv;
      out_i[sel] = gi;
      if (gs >= 0) { s_v[gs] = NEG_INF; s_i[gs] = 0x7fffffff; }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void lstk_select_kernel(const bf16_t* __restrict__ logits, int ld, int V, int k,
                                                          const float* __restrict__ stats, float* __restrict__ cand_v,
                                                          int* __restrict__ cand_i) {
  __shared__ float s_v[1024];
  __shared__ int s_i[1024];
  __shared__ float s_lse;
  const bf16_t* x = logits + (size_t)blockIdx.x * ld;
  int lo, hi;
  lstk_chunk_range(V, blockIdx.y, lo, hi);
  if (threadIdx.x == 0) s_lse = lstk_lse(stats + (size_t)blockIdx.x * LSTK_CHUNKS * 2);
  __syncthreads();
  const float lse = s_lse;
  const int n = max(0, hi - lo);
  float* ov = cand_v + ((size_t)blockIdx.x * LSTK_CHUNKS + blockIdx.y) * TOPK_MAX;
  int* oi = cand_i + ((size_t)blockIdx.x * LSTK_CHUNKS + blockIdx.y) * TOPK_MAX;
  for (int base = 0; base == 0 || base < n; base += 1024 - TOPK_MAX) {
    // (chunks are <= 1024 wide for every vocabulary up to 64k; wider chunks are folded in rounds that keep the running best)
    const int m = min(n - base, 1024 - (base ? TOPK_MAX : 0));
    const int off = base ? TOPK_MAX : 0;
    if (base) {
      if (threadIdx.x < TOPK_MAX) { s_v[threadIdx.x] = threadIdx.x < k ? ov[threadIdx.x] : NEG_INF; s_i[threadIdx.x] = threadIdx.x < k ? oi[threadIdx.x] : 0x7fffffff; }
    }
    for (int c = threadIdx.x; c < m; c += 256) {
      s_v[off + c] = rdbf(bf2f(x[lo + base + c]) - lse);
      s_i[off + c] = lo + base + c;
    }
    __syncthreads();
    block_select_topk(s_v, s_i, off + max(m, 0), k, ov, oi);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void lstk_merge_kernel(int k, const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                         int* __restrict__ out_idx, float* __restrict__ out_logp) {
  __shared__ float s_v[LSTK_CHUNKS * TOPK_MAX];
  __shared__ int s_i[LSTK_CHUNKS * TOPK_MAX];
  __shared__ float o_v[TOPK_MAX];
  __shared__ int o_i[TOPK_MAX];
  const float* cv = cand_v + (size_t)blockIdx.x * LSTK_CHUNKS * TOPK_MAX;
  const int* ci = cand_i + (size_t)blockIdx.x * LSTK_CHUNKS * TOPK_MAX;
  for (int c = threadIdx.x; c < LSTK_CHUNKS * TOPK_MAX; c += 256) {
    const bool valid = (c % TOPK_MAX) < k;
    s_v[c] = valid ? cv[c] : NEG_INF;
    s_i[c] = valid ? ci[c] : 0x7fffffff;
  }
  __syncthreads();
  block_select_topk(s_v, s_i, LSTK_CHUNKS * TOPK_MAX, k, o_v, o_i);
  if (threadIdx.x < k) {
    out_idx[blockIdx.x * k + threadIdx.x] = o_i[threadIdx.x];
    out_logp[blockIdx.x * k + threadIdx.x] = o_v[threadIdx.x];
  }
}
