// Per-workgroup clocks — a DIAGNOSTIC build only (hipcc -DVISPEC_WG_CLOCK via $VISPEC_HIPCC_FLAGS; the product build compiles every macro
// below to nothing and exports none of the entry points).  Round 6: rocprofv3's kernel durations under four lanes cannot tell "a workgroup
// runs slower because the memory system is loaded" from "a workgroup waits for a CU that another lane's kernel holds" — and its --pmc
// passes serialise the kernels, so they do not see the load at all.  With this flag every instrumented workgroup appends one record
//   { t0, t1 : s_memrealtime (100 MHz, chip-wide) at its first and last instruction; kid : kernel id; tag : identifies the launch stream
//     (bits of the output pointer: every lane has its own workspaces); blk / nblk : linear block id and grid size; hw : XCC id and HW_ID;
//     pad0 : shader-clock cycles (s_memtime) between the two — with t1 - t0 the core clock the workgroup ran at }
// to a device buffer (tools/wg_clock.py reads and reduces it): run time per workgroup, start stagger and span per launch, workgroups per CU.
#pragma once
#ifdef VISPEC_WG_CLOCK
struct WgClkRec {
  unsigned long long t0, t1;
  unsigned kid, tag, blk, nblk, hw, xcc, pad0, pad1;
};
__device__ WgClkRec* g_wgclk_buf = nullptr;
__device__ unsigned g_wgclk_cap = 0;
__device__ unsigned g_wgclk_n = 0;
__device__ __forceinline__ void wgclk_end(unsigned long long t0, unsigned kid, const void* tagp, unsigned long long c0 = 0) {
  if (threadIdx.x != 0 || !g_wgclk_buf) return;
  const unsigned long long t1 = wall_clock64();
  const unsigned long long c1 = __builtin_readcyclecounter();  // s_memtime: shader-clock cycles (t0 / t1 are the constant 100 MHz counter)
  const unsigned i = atomicAdd(&g_wgclk_n, 1u);
  if (i >= g_wgclk_cap) return;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  WgClkRec r;
  r.t0 = t0; r.t1 = t1; r.kid = kid; r.tag = (unsigned)((size_t)tagp >> 12);
  r.blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  r.nblk = gridDim.x * gridDim.y * gridDim.z;
  r.hw = hw; r.xcc = xcc;
  r.pad0 = c0 ? (unsigned)(c1 - c0) : 0u;  // shader cycles the workgroup took: / (t1 - t0) x 100 MHz = the core clock it ran at
  r.pad1 = 0;
  g_wgclk_buf[i] = r;
}
struct WgClkScope {  // records at scope exit, whichever return the workgroup's thread 0 takes
  unsigned long long t0;
  unsigned kid;
  const void* tag;
  unsigned long long c0;
  __device__ __forceinline__ WgClkScope(unsigned k, const void* t) : t0(wall_clock64()), kid(k), tag(t), c0(__builtin_readcyclecounter()) {}
  __device__ __forceinline__ ~WgClkScope() { wgclk_end(t0, kid, tag, c0); }
};
#define WGCLK_BEGIN() const unsigned long long wgclk_t0_ = wall_clock64(), wgclk_c0_ = __builtin_readcyclecounter()
#define WGCLK_END(kid, tagp) wgclk_end(wgclk_t0_, (kid), (const void*)(tagp), wgclk_c0_)
#define WGCLK_SCOPE(kid, tagp) WgClkScope wgclk_scope_((kid), (const void*)(tagp))
#else
#define WGCLK_BEGIN() ((void)0)
#define WGCLK_END(kid, tagp) ((void)0)
#define WGCLK_SCOPE(kid, tagp) ((void)0)
#endif
