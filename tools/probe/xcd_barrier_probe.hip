// Probe: what does a grid-wide barrier cost INSIDE a persistent launch on this chip, in the XCD-hierarchical form the MI355X guide prices at
// 4.1 us (price list row "barrier-xcd"), next to the flat one-counter form round 2 measured at 8.1 us (tools/probe/grid_barrier_probe.hip)?
//     hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_barrier_probe.hip -o /tmp/xb && /tmp/xb
// One workgroup (256 threads) per CU, G = 256 (and 512 = two per CU) workgroups, R barriers per launch, host-timed with events; variants:
//   flat      one monotonic device-scope counter, relaxed sc1 polls + s_sleep, release fence before the arrive, acquire after
//   xcd       per-XCC arrival counter -> the XCD's LAST arriver releases (buffer_wbl2 sc1), bumps the top counter, waits for all 8 XCDs, acquires,
//             publishes the XCD's generation word; the other workgroups of the XCD poll that word and acquire (buffer_inv sc1)
//   xcd+data  the same with every workgroup publishing a 128-B record (plain stores) before each barrier and reading ANOTHER workgroup's record after
//             it (checked: a wrong protocol shows up as stale records)
//   xcd+slab  the same with a 64 KiB fp32 slab per workgroup written before the barrier (what a split-K seam publishes)
// Why it matters (DESIGN.md / NOTEBOOK.md "persistent per-layer engine"): a single request's round is ~330 dependent launches; a persistent
// layer would trade each kernel boundary for one of these.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

struct Sync {
  unsigned cnt[8 * 32];   // per-XCC arrival counters, one 128-B line each
  unsigned gen[8 * 32];   // per-XCC generation words
  unsigned top[32];       // XCD leaders' counter
  unsigned flat[32];      // flat variant's counter
  unsigned n_xcc[8];      // census: workgroups per XCC
  unsigned n_active;      // XCCs that hold at least one workgroup
  unsigned timeout;       // set when a bounded spin gave up
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned want, Sync* s) {
  for (unsigned i = 0; i < (1u << 22); ++i) {
    if (ld_rlx(p) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  s->timeout = 1;
  return false;
}

__global__ void census_kernel(Sync* s) {
  if (threadIdx.x == 0) atomicAdd(&s->n_xcc[xcc_id()], 1u);
}
__global__ void census_finish(Sync* s) {
  unsigned n = 0;
  for (int i = 0; i < 8; ++i) n += s->n_xcc[i] != 0;
  s->n_active = n;
}

// MODE 0 flat, 1 xcd, 2 xcd + 128-B record, 3 xcd + 64 KiB slab
template <int MODE>
__global__ __launch_bounds__(256) void barrier_kernel(Sync* s, int R, float* rec, float* slab, unsigned* bad) {
  const unsigned x = xcc_id();
  const unsigned n_local = s->n_xcc[x], n_act = s->n_active, G = gridDim.x;
  __shared__ float seen;
  for (int r = 0; r < R; ++r) {
    const unsigned phase = (unsigned)r + 1;
    if (MODE == 2) {  // publish: this workgroup's record for this phase (32 floats = 128 B)
      if (threadIdx.x < 32) rec[((size_t)(phase & 1) * 512 + blockIdx.x) * 32 + threadIdx.x] = (float)(phase * 1000 + blockIdx.x % 1000);
    }
    if (MODE == 3) {  // a 64 KiB slab per workgroup: 256 threads x 64 floats, 16-byte stores
      float4* d = reinterpret_cast<float4*>(slab + ((size_t)(phase & 1) * 512 + blockIdx.x) * 16384);  // (double-buffered by phase parity: a
                                                                                                    // fast neighbour may already be writing phase r + 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i * 256 + threadIdx.x] = make_float4((float)phase, 1.f, 2.f, 3.f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its own stores before the rendezvous (the guide's rule 14)
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&s->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&s->flat[0], phase * G, s);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else {
        // every workgroup's data must be visible device-wide before the XCD's arrival is counted at the top: a workgroup that published
        // something writes it back itself (plain stores sit in ITS XCD's L2 — the leader's release covers the same L2, so only the
        // leader needs the wbl2; the others just drain their stores)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(&s->cnt[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == phase * n_local - 1) {  // the XCD's last arriver: leader of this phase
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          spin_until(&s->top[0], phase * n_act, s);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(&s->gen[x * 32], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          spin_until(&s->gen[x * 32], phase, s);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
      }
    }
    __syncthreads();
    if (MODE == 2) {  // consume: the record of the workgroup "across the chip"
      const unsigned other = (blockIdx.x + G / 2 + 3) % G;
      if (threadIdx.x < 32) {
        const float v = rec[((size_t)(phase & 1) * 512 + other) * 32 + threadIdx.x];
        if (v != (float)(phase * 1000 + other % 1000)) atomicAdd(bad, 1u);
        if (threadIdx.x == 0) seen = v;
      }
      __syncthreads();  // (the record is re-written next phase only after everybody has read it: one more rendezvous per phase would be
                        //  needed in a real pipeline; here the reader and the next writer of a record are separated by this phase's barrier
                        //  plus the writer's own __syncthreads — the check above would catch a violation)
    }
    if (MODE == 3) {
      const unsigned other = (blockIdx.x + G / 2 + 3) % G;
      const float v = slab[((size_t)(phase & 1) * 512 + other) * 16384 + threadIdx.x * 4];
      if (v != (float)phase) atomicAdd(bad, 1u);
    }
  }
  if (MODE >= 2 && threadIdx.x == 0 && seen < 0.f) bad[1] = 1;
}

template <int MODE>
static int run(const char* name, int G, Sync* s, float* rec, float* slab, unsigned* bad) {
  const int R = 200;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned hbad = 0, tmo = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(s, 0, sizeof(Sync)));
    CK(hipMemset(bad, 0, 8));
    census_kernel<<<G, 64>>>(s);
    census_finish<<<1, 1>>>(s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    barrier_kernel<MODE><<<G, 256>>>(s, R, rec, slab, bad);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    unsigned hb[2]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost)); hbad += hb[0];
    Sync hs; CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost)); tmo |= hs.timeout;
    if (rep == 0) { printf("  census G=%d:", G); for (int i = 0; i < 8; ++i) printf(" %u", hs.n_xcc[i]); printf("\n"); }
  }
  printf("%-10s G = %3d : %6.2f us per barrier (best of 5 launches of %d barriers)%s%s\n", name, G, best * 1e3f / R, R, hbad ? "  STALE DATA SEEN" : "",
         tmo ? "  SPIN TIMEOUT" : "");
  fflush(stdout);
  return 0;
}

int main() {
  Sync* s; float *rec, *slab; unsigned* bad;
  CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&rec, 2 * 512 * 32 * 4)); CK(hipMalloc(&slab, (size_t)2 * 512 * 16384 * 4)); CK(hipMalloc(&bad, 8));
  for (int G : {256, 512}) {
    if (run<0>("flat", G, s, rec, slab, bad)) return 1;
    if (run<1>("xcd", G, s, rec, slab, bad)) return 1;
    if (run<2>("xcd+data", G, s, rec, slab, bad)) return 1;
    if (run<3>("xcd+slab", G, s, rec, slab, bad)) return 1;
  }
  // for scale: the same number of trivial dependent kernel boundaries on one stream
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(s, 0, sizeof(Sync)));
    for (int i = 0; i < 20; ++i) census_kernel<<<256, 256>>>(s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) census_kernel<<<256, 256>>>(s);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("200 trivial dependent 256-workgroup launches on one stream: %.2f us per launch (eager; a graph replays them no faster per the guide)\n", ms * 1e3f / 200);
  }
  return 0;
}
