// libvispec_hip: C-ABI + stream-ordered orchestration of the ViSpec draft-and-verify round on MI355X.
// See include/vispec_hip.h for the contract and the reference functions each entry point replaces.
#include "../../include/vispec_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"
#include "gemm_prefill.h"
#include <mutex>
#include "gemm_wide.h"
#include "gemm_c8.h"
#include "tree_kernels.h"

static thread_local std::string g_err;
static int fail(const std::string& s) {
  g_err = s;
  return -1;
}
#define HIPCHK(x)                                                                                          \
  do {                                                                                                     \
    hipError_t e_ = (x);                                                                                   \
    if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));                     \
  } while (0)
#define KCHK() HIPCHK(hipGetLastError())
// every extern "C" entry point that takes a ctx starts with one of these: a leader destroyed while members were alive is a ZOMBIE (its
// workspaces live on for the members that alias them, its handle is dead) — include/vispec_hip.h: "every entry point refuses it"
#define CTX_LIVE(ctx)                                                                                         \
  do {                                                                                                        \
    if (!(ctx) || (ctx)->zombie) return fail("null ctx (or a leader already destroyed: its handle is invalid)"); \
  } while (0)
#define CTX_LIVE_OPT(ctx) /* entry points that accept ctx == nullptr (bare GEMM launches without workspaces) */ \
  do {                                                                                                        \
    if ((ctx) && (ctx)->zombie) return fail("the leader ctx was already destroyed: its handle is invalid");      \
  } while (0)

struct vispec_ctx {
  vispec_config c;
  std::vector<vispec_layer_weights> layers;
  vispec_target_misc tm{};
  vispec_draft_weights dw{};
  bf16_t* target_kv = nullptr;
  bf16_t* draft_kv = nullptr;
  // ---- workspace (device) ----
  DevState* st = nullptr;
  int* tokens = nullptr;  // committed token ids [tokens_cap]
  int tokens_cap = 0;
  int* accept_log = nullptr;
  int log_cap = 0;
  // target verify buffers (64 rows each)
  bf16_t *xa, *xn, *qkv, *attn_o, *act, *hidden_new, *logits;
  int *am, *sel, *draft_ids;
  bf16_t* accept_hidden;  // [16, D]
  // draft buffers
  bf16_t *dx1, *dx2, *dx, *dqkv, *dattn, *dh, *dn, *dact, *dout, *dlast, *dlogits, *dg;
  bf16_t *xc, *emb_shift, *ad_kv, *ad_out, *ad_tmp, *pf_t1;  // prefill-only scratch
  int *top_idx, *pos_c, *idx_tmp, *idx_img, *scratch_int;
  int* h_pin = nullptr;  // pinned host staging for the prefill's index lists [3 * draft_max_pos]
  void* h_state = nullptr;       // pinned: two DevState snapshots (vispec_cohort_state_enqueue / _wait: the pipelined round loop)
  hipEvent_t st_ev[2] = {nullptr, nullptr};  // (leader) one event per snapshot slot
  float* top_logp;
  unsigned long long* causal_mask;  // [64] row i sees tail keys 0..i
  TreeBufs tb{};
  // attention partials
  float *part_o, *part_ml;
  int* att_cnt = nullptr;  // arrival counters of the fused attention merge (tree_attn2_partial_kernel): zero between launches
  float *lstk_stats = nullptr, *lstk_cv = nullptr;  // log-softmax/top-k scratch [64 rows][LSTK_CHUNKS]...
  int* lstk_ci = nullptr;
  unsigned long long* lstk2_cand = nullptr;  // chunked form (large vocabularies): [64 rows][<= 64 chunks][TOPK_MAX] keys; statistics in lstk_stats
  float* gemm_part = nullptr;  // split-K partial sums [S<=8][32][N]
  size_t gemm_part_elems = 0;
  size_t part_cap_tiles = 0;  // number of (q-tile, split) partial tiles that fit
  int n_hint = 0;             // upper bound of keys any attention call of this request can see
  int scr_rows = 0;           // rows of the prefill-only scratch (a prompt may be as long as the TARGET cache; only its compressed form must fit the draft's)
  int rope_rows = 0;          // rows of the draft rotary tables
  // hipGraph cache: the launch sequence of a round is static for a given (n_hint, forced_accept) — see run_graphed()
  // A captured graph bakes its kernel arguments in, so the cache key is the full tuple of everything that reaches a launch as an
  // argument or shapes the launch sequence — compared field by field (a hashed single integer could collide and replay a graph
  // with the wrong sampling parameters).
  struct GraphKey {
    int n_req = 0, forced_accept = 0, total_token = 0, wide_rb = 0, u_over = 0;
    // per request of the (cohort) round, leader first.  `who` = the ctx itself: a captured graph bakes that request's pointers in
    // (state, tree, KV, selections ...), so the same leader with a DIFFERENT member set must not replay it.
    const void* who[MAX_COHORT] = {};
    int n_hint[MAX_COHORT] = {-1, -1, -1, -1, -1, -1, -1, -1}, sample_top_k[MAX_COHORT] = {};
    float temperature[MAX_COHORT] = {};
    unsigned long long seed[MAX_COHORT] = {};
    bool operator==(const GraphKey& o) const {
      return n_req == o.n_req && forced_accept == o.forced_accept && total_token == o.total_token && wide_rb == o.wide_rb && u_over == o.u_over &&
             memcmp(who, o.who, sizeof(who)) == 0 &&
             memcmp(n_hint, o.n_hint, sizeof(n_hint)) == 0 && memcmp(sample_top_k, o.sample_top_k, sizeof(sample_top_k)) == 0 &&
             memcmp(temperature, o.temperature, sizeof(temperature)) == 0 && memcmp(seed, o.seed, sizeof(seed)) == 0;
    }
  };
  // A few graphs per round function, least recently used one replaced: a stream of requests through the slots of a cohort
  // (specgenerate_stream) alternates between a handful of keys (context-length buckets), and a re-capture costs milliseconds.
  struct GraphSlot {
    static constexpr int CAP = 4;
    hipGraphExec_t exec[CAP] = {nullptr, nullptr, nullptr, nullptr};
    GraphKey key[CAP];
    unsigned long stamp[CAP] = {0, 0, 0, 0}, clock = 0;
    void clear() {
      for (int i = 0; i < CAP; ++i) {
        if (exec[i]) (void)hipGraphExecDestroy(exec[i]);
        exec[i] = nullptr;
        key[i] = GraphKey{};
        stamp[i] = 0;
      }
    }
  };
  GraphSlot g_verify, g_draft, g_ar, g_cverify, g_cdraft, g_car;
  bool a8 = false;               // fp8 ACTIVATIONS for the target's q|k|v, gate|up and down GEMMs (needs fp8 weights): vispec_set_fp8_activations
  unsigned char* xq = nullptr;   // [128 rows][max(D, H * 128, I)] e4m3 codes of the GEMM input at hand (stream-ordered scratch)
  float* sx = nullptr;           // [128] their per-row scales
  float* u_over = nullptr;  // tests only: uniforms of the sampling accept taken from here (vispec_set_uniform_override_host)
  bool u_over_on = false;
  bool zombie = false;  // a leader destroyed while members were alive: its workspaces (which the members alias) are freed with the last member
  vispec_ctx* leader = nullptr;  // non-null: a cohort member — its activation buffers are 32-row tile `slot` of the leader's
  int slot = 0;                  // activation tile of this request inside the leader's 256-row workspaces (leader: 0, members: 1..7)
  vispec_ctx* members[MAX_COHORT - 1] = {};  // (leader) the member that owns tile 1 .. 7
  bool has_members() const { for (const vispec_ctx* m : members) if (m) return true; return false; }
  float temperature = 0.f;          // > 1e-5: sampling path (spec_model_ours.py:272-277)
  int sample_top_k = 0;             // > 0: TopKLogitsWarper after the temperature (utils.py:52-53)
  unsigned long long seed = 0;
  bool use_graphs = true;
  int wide_rb = 4;                  // weight row blocks per workgroup of the wide-cohort GEMM: 4, 2, or 0 = two where four leave half the CUs idle
  long graph_replays = 0, graph_captures = 0, direct_runs = 0;
  std::vector<void*> allocs;
};

template <class T>
static int dalloc(vispec_ctx* ctx, T** p, size_t n) {
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, n * sizeof(T) + 256));
  HIPCHK(hipMemset(q, 0, n * sizeof(T) + 256));
  ctx->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}

extern "C" const char* vispec_last_error(void) { return g_err.c_str(); }
extern "C" int vispec_version(void) { return 1; }

#define ROWS (32 * MAX_COHORT)  /* rows of the activation workspaces: eight 32-row tiles (a cohort of up to eight requests shares one weight pass) */
// A member's TARGET-side activation views (what the verify forward of a cohort round touches): tile `slot` of its leader's workspaces for trees
// of up to 32 nodes (32 rows per request), tiles 2 slot and 2 slot + 1 for trees of 33..64 nodes (64 rows per request; slots 0..3) — so that
// `total_token = -1` autotuning (spec_model_ours.py:179-201: 40..60 nodes) and cohorts compose.  The DRAFT-side views (<= 16 live rows per
// request) always stay at 32 rows per slot.  Called at creation and whenever the tree size changes (vispec_set_total_token).
static void retarget_views(vispec_ctx* ctx) {
  vispec_ctx* ld = ctx->leader;
  if (!ld) return;
  const vispec_config& c = ctx->c;
  const size_t rows = (size_t)(c.total_token > 32 ? 64 : 32) * ctx->slot;
  const size_t D = c.hidden_size, QKV = (size_t)(c.num_heads + 2 * c.num_kv_heads) * c.head_dim;
  ctx->xa = ld->xa + rows * D; ctx->xn = ld->xn + rows * D; ctx->qkv = ld->qkv + rows * QKV;
  ctx->attn_o = ld->attn_o + rows * (size_t)c.num_heads * c.head_dim; ctx->act = ld->act + rows * (size_t)c.intermediate_size;
  ctx->hidden_new = ld->hidden_new + rows * D; ctx->logits = ld->logits + rows * (size_t)c.vocab_size; ctx->am = ld->am + rows;
}
#define CHUNK 32  /* rows per skinny-GEMM pass when a prefill stage walks a long sequence */
static int ctx_create_impl(const vispec_config* cfg, vispec_ctx* leader, vispec_ctx** out) {
  if (!cfg || !out) return fail("null argument");
  if (leader && leader->zombie) return fail("ctx_create_member: the leader was destroyed");
  if (leader && (leader->leader || memcmp(&leader->c, cfg, sizeof(vispec_config)) != 0))
    return fail("ctx_create_member: the leader must be an ordinary ctx created with the same config");
  int slot = 0;
  if (leader) {
    for (slot = 1; slot < MAX_COHORT && leader->members[slot - 1]; ++slot) {}
    if (slot >= MAX_COHORT) return fail("ctx_create_member: the leader already has seven members (a cohort is at most eight requests)");
    // trees of 33..64 nodes (autotune_total_token): a request then owns TWO activation tiles of the target-side workspaces (retarget_views),
    // which the first four request slots have
    if (cfg->total_token > 32 && slot > 3) return fail("ctx_create_member: trees of more than 32 nodes take two activation tiles per request: at most four requests per cohort");
  }
  const vispec_config& c = *cfg;
  if (c.head_dim != 128) return fail("head_dim must be 128 (attention tiles are written for 128)");
  if (c.hidden_size % 64 || c.intermediate_size % 64 || c.draft_intermediate % 64 || c.vocab_size % 16)
    return fail("GEMM dims: K %% 64 == 0 and N %% 16 == 0 required");
  if (c.total_token < 1 || c.total_token > TREE_MAX_T) return fail("total_token must be in [1,64]");
  if (c.top_k < 1 || c.top_k > TREE_MAX_K || c.depth < 1 || c.depth > TREE_MAX_DEPTH) return fail("top_k<=16, depth<=8");
  if (c.top_k * (c.depth + 1) > 64) return fail("top_k*(depth+1) must be <= 64 (one mask word per draft row)");
  if (c.total_token - 1 > c.top_k + c.depth * c.top_k * c.top_k) return fail("total_token larger than the candidate pool");
  if (c.hidden_size != c.draft_heads * c.head_dim) return fail("draft must be MHA with head_dim 128 over hidden_size");
  if (c.num_heads % c.num_kv_heads) return fail("num_heads %% num_kv_heads != 0");
  if (c.draft_rope_rows < 0) return fail("draft_rope_rows must be >= 0");
  vispec_ctx* ctx = new vispec_ctx();
  ctx->c = c;
  ctx->leader = leader;
  ctx->slot = slot;
  ctx->layers.resize(c.num_layers);
  const size_t D = c.hidden_size, I = c.intermediate_size, Id = c.draft_intermediate, V = c.vocab_size;
  const size_t QKV = (size_t)(c.num_heads + 2 * c.num_kv_heads) * c.head_dim;
#define A(p, n)                                   \
  if (dalloc(ctx, &ctx->p, (n))) {                \
    vispec_ctx_destroy(ctx);                      \
    return -1;                                    \
  }
  // row-wise activation workspaces (ROWS = 128 rows of `ld` elements): a cohort member's are 32-row tile `slot` of its leader's,
  // so that a GEMM launched once on the leader's rows serves every request of the cohort
#define AL(p, ld)                                           \
  if (leader) ctx->p = leader->p + (size_t)32 * slot * (ld); \
  else A(p, (size_t)ROWS * (ld))
  A(st, 1);
  ctx->tokens_cap = c.max_pos + 64;
  A(tokens, ctx->tokens_cap);
  ctx->log_cap = c.max_pos;
  A(accept_log, ctx->log_cap);
  AL(xa, D); AL(xn, D); AL(qkv, QKV); AL(attn_o, (size_t)c.num_heads * c.head_dim);
  AL(act, I); AL(hidden_new, D); AL(logits, V);
  AL(am, 1); A(sel, 16); A(draft_ids, 16); A(accept_hidden, 16 * D); A(u_over, TREE_MAX_T * TREE_RET_W + 1);
  if (!leader) {
    const size_t kmax = (size_t)std::max(std::max((int)c.hidden_size, (int)(c.num_heads * c.head_dim)), (int)c.intermediate_size);
    A(xq, (size_t)ROWS * kmax); A(sx, ROWS);
  } else {
    ctx->xq = leader->xq; ctx->sx = leader->sx;  // (a member's own GEMMs are stream-ordered with its leader's, like gemm_part: one scratch)
  }
  AL(dx1, 2 * D); AL(dx2, 2 * D); AL(dx, D); AL(dqkv, 3 * D); AL(dattn, D);
  AL(dh, D); AL(dn, D); AL(dact, Id); AL(dout, D); AL(dlast, D);
  AL(dlogits, V); A(dg, D);
  const int scr = c.max_pos > c.draft_max_pos ? c.max_pos : c.draft_max_pos;
  ctx->scr_rows = scr;
  ctx->rope_rows = c.draft_rope_rows > 0 ? c.draft_rope_rows : c.draft_max_pos;
  A(xc, (size_t)scr * D); A(emb_shift, (size_t)scr * D); A(pf_t1, (size_t)scr * D);
  A(ad_kv, (size_t)2 * scr * D); A(ad_out, 16 * D); A(ad_tmp, ROWS * 2 * D);
  A(top_idx, TREE_MAX_K * TREE_MAX_K); A(top_logp, TREE_MAX_K * TREE_MAX_K); A(pos_c, scr);
  A(idx_tmp, scr); A(idx_img, scr); A(scratch_int, 4);
  if (hipHostMalloc((void**)&ctx->h_pin, sizeof(int) * 3 * (size_t)scr) != hipSuccess) {
    vispec_ctx_destroy(ctx);
    return fail("hipHostMalloc failed");
  }
  A(causal_mask, 64);
  {
    unsigned long long cm[64];
    for (int i = 0; i < 64; ++i) cm[i] = (i == 63) ? ~0ull : ((2ull << i) - 1ull);
    if (hipMemcpy(ctx->causal_mask, cm, sizeof(cm), hipMemcpyHostToDevice) != hipSuccess) {
      vispec_ctx_destroy(ctx);
      return fail("causal mask upload failed");
    }
  }
  A(tb.scores_all, TREE_MAX_SCORES); A(tb.tokens_all, TREE_MAX_SCORES); A(tb.parents_all, 1 + TREE_MAX_DEPTH * TREE_MAX_K);
  A(tb.cur_scores, TREE_MAX_K); A(tb.cs_idx, TREE_MAX_K); A(tb.in_ids, TREE_MAX_K); A(tb.lvl_mask, TREE_MAX_K);
  A(tb.tree_tokens, TREE_MAX_T); A(tb.tree_pos, TREE_MAX_T); A(tb.tree_mask, TREE_MAX_T);
  A(tb.retrieve, TREE_MAX_T * TREE_RET_W); A(tb.mask_index, TREE_MAX_T);
  {
    // partial tiles: (H * ceil(64/32)) q-tiles x up to 64 splits
    int maxpos = c.max_pos > c.draft_max_pos ? c.max_pos : c.draft_max_pos;
    if (maxpos < 4096) maxpos = 4096;  // the unit-level C-ABI entry may be used with caches larger than this model's
    const size_t nsplit = (size_t)(maxpos + 64 + 127) / 128 + 1;
    const int heads = c.num_heads > 16 ? c.num_heads : 16;
    ctx->part_cap_tiles = (size_t)heads * 2 * (nsplit > 64 ? 64 : nsplit);
    A(part_o, ctx->part_cap_tiles * 128 * 32);
    A(part_ml, ctx->part_cap_tiles * 64);
    A(att_cnt, 512);
  }
  {
    size_t nmax = (size_t)(c.num_heads + 2 * c.num_kv_heads) * c.head_dim;
    if (nmax < (size_t)3 * c.hidden_size) nmax = (size_t)3 * c.hidden_size;
    if (nmax < 16384) nmax = 16384;
    ctx->gemm_part_elems = (size_t)8 * ROWS * nmax;  // [S <= 8][up to eight 32-row tiles][N] fp32
    if (leader) ctx->gemm_part = leader->gemm_part;  // launches of a cohort are stream-ordered: one partial workspace serves both
    else A(gemm_part, ctx->gemm_part_elems);
    A(lstk_stats, 64 * LSTK_CHUNKS * 2); A(lstk_cv, 64 * LSTK_CHUNKS * TOPK_MAX); A(lstk_ci, 64 * LSTK_CHUNKS * TOPK_MAX);
    A(lstk2_cand, 64 * LSTK_CHUNKS * TOPK_MAX);
  }
#undef A
#undef AL
  retarget_views(ctx);
  ctx->n_hint = c.max_pos;
  const int lds = ATT2_LDS_BYTES;
  if (hipFuncSetAttribute((const void*)tree_attn2_partial_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
          hipSuccess ||
      hipFuncSetAttribute((const void*)tree_attn2_partial_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
          hipSuccess) {
    (void)hipGetLastError();  // a device-less build/load check must still be able to create nothing; report lazily
  }
  {  // the wide-cohort GEMM declares 128 KiB of dynamic LDS (WIDE_LDS_BYTES)
#define WIDE_ALL_EPI_RB(W8_, NL_, RB_)                                                                                              \
  (const void*)gemm_w32_wide_kernel<EPI_NONE, W8_, NL_, 0, RB_>, (const void*)gemm_w32_wide_kernel<EPI_RESIDUAL, W8_, NL_, 0, RB_>,   \
      (const void*)gemm_w32_wide_kernel<EPI_SWIGLU, W8_, NL_, 0, RB_>, (const void*)gemm_w32_wide_kernel<EPI_PARTIAL, W8_, NL_, 0, RB_>, \
      (const void*)gemm_w32_wide_kernel<EPI_ROPE, W8_, NL_, 0, RB_>
#define WIDE_ALL_EPI(W8_, NL_) WIDE_ALL_EPI_RB(W8_, NL_, 4), WIDE_ALL_EPI_RB(W8_, NL_, 3), WIDE_ALL_EPI_RB(W8_, NL_, 2)
    const void* wide[] = {WIDE_ALL_EPI(0, 3), WIDE_ALL_EPI(0, 4), WIDE_ALL_EPI(1, 3), WIDE_ALL_EPI(1, 4), WIDE_ALL_EPI_RB(2, 3, 4), WIDE_ALL_EPI_RB(2, 4, 4),
                          (const void*)gemm_w32_wide_kernel<EPI_PARTIAL, false, 4, 1>, (const void*)gemm_w32_wide_kernel<EPI_PARTIAL, false, 4, 2>};
#undef WIDE_ALL_EPI
#undef WIDE_ALL_EPI_RB
    for (const void* f : wide)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS_BYTES) != hipSuccess) (void)hipGetLastError();
    // the eight-row-block form (gemm_w32_wide8_kernel): 64 KiB (bf16) / 96 KiB (fp8 weights) of dynamic LDS
#define WIDE8_ALL_EPI(W8_, NL_)                                                                                                      \
  (const void*)gemm_w32_wide8_kernel<EPI_NONE, W8_, NL_>, (const void*)gemm_w32_wide8_kernel<EPI_RESIDUAL, W8_, NL_>,                  \
      (const void*)gemm_w32_wide8_kernel<EPI_SWIGLU, W8_, NL_>, (const void*)gemm_w32_wide8_kernel<EPI_PARTIAL, W8_, NL_>,             \
      (const void*)gemm_w32_wide8_kernel<EPI_ROPE, W8_, NL_>
    const void* wide8_bf16[] = {WIDE8_ALL_EPI(0, 3), WIDE8_ALL_EPI(0, 4), WIDE8_ALL_EPI(2, 3), WIDE8_ALL_EPI(2, 4)};  // (W8A8: the bf16 form's ring)
    const void* wide8_fp8[] = {WIDE8_ALL_EPI(1, 3), WIDE8_ALL_EPI(1, 4)};
#undef WIDE8_ALL_EPI
    for (const void* f : wide8_bf16)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, wide8_lds_bytes<0>()) != hipSuccess) (void)hipGetLastError();
    for (const void* f : wide8_fp8)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, wide8_lds_bytes<1>()) != hipSuccess) (void)hipGetLastError();
    // the cohort-8 form (gemm_w32_c8_kernel): C8_LDS_BYTES = (C8_LA + 2) x 32 KiB = 128 KiB of dynamic LDS (160 KiB with -DC8_LA=3: the whole CU)
#define C8_ALL_EPI(W8_)                                                                                                     \
  (const void*)gemm_w32_c8_kernel<EPI_NONE, W8_>, (const void*)gemm_w32_c8_kernel<EPI_RESIDUAL, W8_>,                          \
      (const void*)gemm_w32_c8_kernel<EPI_SWIGLU, W8_>, (const void*)gemm_w32_c8_kernel<EPI_PARTIAL, W8_>,                     \
      (const void*)gemm_w32_c8_kernel<EPI_ROPE, W8_>
    const void* c8[] = {C8_ALL_EPI(0), C8_ALL_EPI(1), C8_ALL_EPI(2)};
#undef C8_ALL_EPI
    for (const void* f : c8)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS_BYTES) != hipSuccess) (void)hipGetLastError();
  }
  if (leader) leader->members[slot - 1] = ctx;
  *out = ctx;
  return 0;
}

extern "C" int vispec_ctx_create(const vispec_config* cfg, vispec_ctx** out) { return ctx_create_impl(cfg, nullptr, out); }
extern "C" int vispec_ctx_create_member(const vispec_config* cfg, vispec_ctx* leader, vispec_ctx** out) {
  if (!leader) return fail("ctx_create_member: null leader");
  return ctx_create_impl(cfg, leader, out);
}

static void ctx_free(vispec_ctx* ctx) {
  for (auto* g : {&ctx->g_verify, &ctx->g_draft, &ctx->g_ar, &ctx->g_cverify, &ctx->g_cdraft, &ctx->g_car}) g->clear();
  for (void* p : ctx->allocs) (void)hipFree(p);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->h_state) (void)hipHostFree(ctx->h_state);
  for (hipEvent_t e : ctx->st_ev)
    if (e) (void)hipEventDestroy(e);
  delete ctx;
}
extern "C" void vispec_ctx_destroy(vispec_ctx* ctx) {
  if (!ctx || ctx->zombie) return;
  if (ctx->leader && ctx->slot >= 1 && ctx->leader->members[ctx->slot - 1] == ctx) {
    vispec_ctx* ld = ctx->leader;
    ld->members[ctx->slot - 1] = nullptr;
    // the leader's cached cohort graphs bake this member's buffers in, and a later member may be allocated at this very address
    // (the graph key compares ctx pointers): drop them with the member
    for (auto* g : {&ld->g_cverify, &ld->g_cdraft, &ld->g_car}) g->clear();
    if (ld->zombie && !ld->has_members()) ctx_free(ld);  // the last member of a destroyed leader
  }
  // A member's GEMM workspaces ARE rows of its leader's (ctx_create_impl, AL): a leader destroyed while members are alive keeps its
  // allocations until the last of them is gone (the members stay usable as single requests; the destroyed leader handle must not be used)
  if (ctx->has_members()) {
    for (auto* g : {&ctx->g_verify, &ctx->g_draft, &ctx->g_ar, &ctx->g_cverify, &ctx->g_cdraft, &ctx->g_car}) g->clear();
    ctx->zombie = true;
    return;
  }
  ctx_free(ctx);
}

extern "C" int vispec_set_target_layer(vispec_ctx* ctx, int layer, const vispec_layer_weights* w) {
  CTX_LIVE(ctx);
  if (!ctx || !w || layer < 0 || layer >= ctx->c.num_layers) return fail("bad layer");
  ctx->layers[layer] = *w;
  return 0;
}
extern "C" int vispec_set_target_misc(vispec_ctx* ctx, const vispec_target_misc* m) {
  CTX_LIVE(ctx);
  if (!ctx || !m) return fail("null");
  ctx->tm = *m;
  return 0;
}
extern "C" int vispec_set_draft_weights(vispec_ctx* ctx, const vispec_draft_weights* w) {
  CTX_LIVE(ctx);
  if (!ctx || !w) return fail("null");
  ctx->dw = *w;
  return 0;
}
extern "C" int vispec_set_kv(vispec_ctx* ctx, void* target_kv, void* draft_kv) {
  CTX_LIVE(ctx);
  if (!ctx) return fail("null");
  ctx->target_kv = (bf16_t*)target_kv;
  ctx->draft_kv = (bf16_t*)draft_kv;
  return 0;
}

// ------------------------------------------------------------------------------------------------ in-library profiling
// Per-dispatch device timestamps of the kernels of a "kind", used by bench.py for the roofline object: in profiling mode every
// skinny-GEMM / attention launch goes through hipExtLaunchKernel with a start/stop event pair, which carry the begin/end timestamps
// of THAT dispatch on the stream it runs on (what rocprofv3 --kernel-trace reports; a plain hipEventRecord pair around a launch
// would also see the dispatch gap in front of it).  Off by default; profiling disables graph replay.
enum { PROF_KINDS = 16, PROF_QKV_ROPE = 5, PROF_ATT_PARTIAL = 9, PROF_ATT_REDUCE = 10, PROF_ATT_PARTIAL_DRAFT = 12, PROF_ATT_REDUCE_DRAFT = 13 };
// 0..4: skinny GEMM none / residual / swiglu / split-K partial / split-K reduce(+norm); 5: q|k|v + rotary + KV append
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;  // 2 per record
  std::vector<int> kind;
  std::vector<double> bytes;
  std::vector<double> wgs;  // workgroups of the bracketed launch (PLAUNCH): CUs asked for
  size_t used = 0;
  hipEvent_t cur_a = nullptr, cur_b = nullptr;  // events of the launch being bracketed
} g_prof;
static void prof_begin(hipStream_t, int kind, double bytes) {
  if (!g_prof.on) return;
  if (g_prof.used * 2 + 2 > g_prof.ev.size()) {
    for (int i = 0; i < 512; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; g_prof.ev.push_back(e); }
  }
  g_prof.kind.resize(g_prof.used + 1);
  g_prof.bytes.resize(g_prof.used + 1);
  g_prof.wgs.resize(g_prof.used + 1);
  g_prof.wgs[g_prof.used] = 0.0;
  g_prof.kind[g_prof.used] = kind;
  g_prof.bytes[g_prof.used] = bytes;
  g_prof.cur_a = g_prof.ev[2 * g_prof.used];
  g_prof.cur_b = g_prof.ev[2 * g_prof.used + 1];
}
static void prof_end(hipStream_t) {
  if (!g_prof.on || !g_prof.cur_a) return;
  g_prof.cur_a = g_prof.cur_b = nullptr;
  ++g_prof.used;
}
// launch inside a prof_begin / prof_end bracket
#define PLAUNCH(kernel, grid, block, lds, s, ...)                                                                          \
  do {                                                                                                                      \
    if (g_prof.cur_a) {                                                                                                     \
      const dim3 g_ = (grid);                                                                                               \
      g_prof.wgs[g_prof.used] = (double)g_.x * g_.y * g_.z;                                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, s, g_prof.cur_a, g_prof.cur_b, 0, __VA_ARGS__);                       \
    } else hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                                                    \
  } while (0)
// One launch of a small per-request kernel for ALL requests of a cohort (csrc/kernels.h: batch4_kernel runs the kernel's body with the
// argument pack of request blockIdx.z).
template <class Fn, int TPB, class P>
static void launch_batch(hipStream_t s, dim3 grid, size_t lds, const P* packs, int n) {
  Packs4<P> a;
  for (int t = 0; t < MAX_COHORT; ++t) a.p[t] = packs[t < n ? t : 0];
  grid.z = n;
  hipLaunchKernelGGL((batch4_kernel<Fn, TPB, P>), grid, dim3(TPB), lds, s, a);
}
extern "C" int vispec_prof_enable(vispec_ctx*, int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  g_prof.cur_a = g_prof.cur_b = nullptr;
  return 0;
}
// out[kind*4 + {0,1,2,3}] = {launches, total ms, total algorithmic bytes, total workgroups}; blocking.
extern "C" int vispec_prof_report_host(vispec_ctx*, void* stream, double* out, int n_kinds) {
  if (!out || n_kinds < 1) return fail("bad args");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < n_kinds * 4; ++i) out[i] = 0.0;
  for (size_t r = 0; r < g_prof.used; ++r) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[2 * r], g_prof.ev[2 * r + 1]));
    const int k = g_prof.kind[r];
    if (k < 0 || k >= n_kinds) continue;
    out[k * 4 + 0] += 1.0;
    out[k * 4 + 1] += ms;
    out[k * 4 + 2] += g_prof.bytes[r];
    out[k * 4 + 3] += g_prof.wgs[r];
  }
  g_prof.used = 0;
  return 0;
}

#ifdef VISPEC_WG_CLOCK
// Diagnostic build only (csrc/wgclock.h; tools/wg_clock.py): the record buffer of the per-workgroup clocks.  Not declared in
// include/vispec_hip.h and absent from the product build.
extern "C" int vispec_debug_wgclock_set(void* buf, unsigned cap) {
  WgClkRec* b = (WgClkRec*)buf;
  const unsigned zero = 0;
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wgclk_buf), &b, sizeof(b)));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wgclk_cap), &cap, sizeof(cap)));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wgclk_n), &zero, sizeof(zero)));
  return 0;
}
extern "C" long long vispec_debug_wgclock_count(void) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgclk_n), sizeof(n)) != hipSuccess) return -1;
  return (long long)n;
}
#endif

// ------------------------------------------------------------------------------------------------ launch helpers
// packed size in bf16 elements of an [N, K] weight in the W32 layout (rows padded to a multiple of 32)
static size_t packed_elems(int N, int K) { return (size_t)((N + 31) / 32) * 32 * K; }

// tuning / A-B switch (VISPEC_MT2_SINGLE_BLOCK=1): two-tile GEMMs with one row block per workgroup, the round-2a form
static const bool g_mt2_single_block = getenv("VISPEC_MT2_SINGLE_BLOCK") && atoi(getenv("VISPEC_MT2_SINGLE_BLOCK")) != 0;

// fp8 tiles too: the paired instantiation needs 256 VGPRs + 76 B of scratch per lane and still wins clearly — with half-size weights the
// activation re-read is FOUR times the weight bytes in the single-block form (Qwen2.5-VL-7B fp8, 4 lanes x cohort 2: 1155 -> 1465 tok/s)
#ifndef VISPEC_W8_PAIR
#define VISPEC_W8_PAIR 1
#endif
// fp8 tiles at ONE activation tile: the bf16 activations are already twice the weight bytes there (the ratio the bf16 path only has
// with two tiles), so the paired form is tried for them as well
#ifndef VISPEC_W8_PAIR_MT1
#define VISPEC_W8_PAIR_MT1 1
#endif
static int launch_pack(hipStream_t s, const void* W, int N, int K, void* P) {
  if (K % 16) return fail("pack: K %% 16");
  hipLaunchKernelGGL(pack_w32_kernel, dim3(K / 16, (N + 31) / 32), dim3(64), 0, s, (const bf16_t*)W, N, K, (bf16_t*)P);
  KCHK();
  return 0;
}

// Split-K factor from the measured launch model of the skinny GEMM (tools/gemm_balance.py, kernel alone):
//   t ~= 5 us + 11.3 us x (workgroup bytes / 256 KiB) x (full rounds of 256 workgroups + 0.55 + 0.45 x fill of a partial round)
// i.e. the chip runs workgroups in rounds of one per CU, a round that is barely started still costs more than half a full one, and
// every launch pays ~5 us of ramp.  S > 1 adds the reduce launch unless a fused norm needs that launch anyway.  Ties go to the
// decomposition closest to two workgroups per CU.
static int choose_split(int tiles, int KS, double tile_bytes, bool reduce_is_free) {
  int best = 1;
  double best_t = 1e30;
  for (int S = reduce_is_free ? 2 : 1; S <= 8; ++S) {
    if (S > 1 && KS / S < 8) break;
    const int wgs = tiles * S, full = wgs / 256, part = wgs % 256;
    const double rounds = full + (part ? 0.55 + 0.45 * part / 256.0 : 0.0);
    double t = 5.0 + 11.3 * (tile_bytes / S / 262144.0) * rounds + (S > 1 && !reduce_is_free ? 5.5 : 0.0) + 0.05 * S;
    if (t < best_t * 0.97 || (t < best_t * 1.03 && std::abs(wgs - 512) < std::abs(tiles * best - 512) && t < best_t * 1.03)) {
      if (t < best_t) best_t = t;
      best = S;
    }
  }
  return best;
}

// One skinny GEMM on a W32-packed weight.  Small-N problems are split over K across workgroups (fp32 partials in
// ctx->gemm_part) and finished by splitk_reduce_kernel, which also applies bias / residual and, when `norm_w` is given, the
// RMSNorm that follows in the layer (one kernel boundary and one activation round trip less).
//   Y (bf16, ld ldy) may be null when only the normed output is wanted.  M <= 32.
struct GemmOut {
  const float* wscale = nullptr;  // non-null: P is an fp8 (e4m3) W32 image with per-output-channel scales (W8A16)
  const float* xscale = nullptr;  // non-null (with wscale): X holds e4m3 CODES (ldx = row pitch in 2-byte units) with these per-row scales (W8A8)
  void* Y = nullptr; int ldy = 0;
  const void* R = nullptr; int ldr = 0;
  const void* norm_w = nullptr; void* normed = nullptr; int ldn = 0; float eps = 0.f;
  unsigned char* q8 = nullptr; float* sx8 = nullptr;  // with normed (W8A8): the normed rows also as e4m3 codes [row][N] + per-row scales (the next GEMM's input)
  int m_tile = 0;  // > 0: cohort mode — two requests share the weight pass: tile t holds request t's rows 32t .. 32t + m_tile - 1 (M = 32 + m_tile)
                   // < 0: slab mode — 2..4 requests of -m_tile <= 8 rows each packed into ONE activation tile (M = 8 (n - 1) - m_tile): tile row
                   //      8t + i is row 32t + i of X / Y / R (kernels.h, gemm_w32_kernel SLAB)
};
// MT = number of 32-row activation tiles (M <= 32*MT): each weight tile held in registers feeds MT MFMAs, so trees of 33..64
// nodes (and two requests sharing a launch) still stream every weight once.
template <int MT>
static int launch_gemm_mt(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, int M, int N, int K,
                          int epi, const GemmOut& o, int force_split) {
  if (M < 1 || M > 32 * MT) return fail("gemm_skinny: M out of range");
  if (o.m_tile < 0 && o.xscale) return fail("gemm_skinny: no fp8-activation slab form");
  if (N % 8 || K % 16 || (o.wscale && K % 32)) return fail("gemm_skinny: N %% 8 == 0 and K %% 16 (fp8: 32) == 0 required");
  if (epi == EPI_RESIDUAL && !o.R) return fail("gemm_skinny: residual epilogue without R");
  const bf16_t *x = (const bf16_t*)X, *w = (const bf16_t*)P, *b = (const bf16_t*)bias, *r = (const bf16_t*)o.R;
  if (o.xscale && (!o.wscale || K % 64)) return fail("gemm_skinny: fp8 activations need fp8 weights and K %% 64 == 0");
  const int tiles = (N + 31) / 32, KS = K / (o.xscale ? 64 : (o.wscale ? 32 : 16));
#define VISPEC_GEMM(NT_, EPI_, W8_, GRID, T2OFF, BIAS, YPTR, LDY, RPTR, LDR, SPLITS, SCALE)                                               \
  do {                                                                                                                                    \
    if (o.m_tile < 0)                                                                                                                     \
      PLAUNCH((gemm_w32_kernel<NT_, EPI_, 4, 4, 0, ((W8_) == 2 ? 1 : (W8_)), MT, true>), GRID, dim3(256), (gemm_w32_lds_bytes<NT_, 4, 4, MT>()), s, x, ldx, w, \
              T2OFF, BIAS, YPTR, LDY, RPTR, LDR, M, N, K, SPLITS, SCALE, RopeEpi{}, o.m_tile, (const float*)nullptr);                      \
    else                                                                                                                                  \
      PLAUNCH((gemm_w32_kernel<NT_, EPI_, 4, 4, 0, W8_, MT>), GRID, dim3(256), (gemm_w32_lds_bytes<NT_, 4, 4, MT>()), s, x, ldx, w,       \
              T2OFF, BIAS, YPTR, LDY, RPTR, LDR, M, N, K, SPLITS, SCALE, RopeEpi{}, o.m_tile, o.xscale);                                  \
  } while (0)
  // Two activation tiles: a workgroup takes TWO row blocks (tile, tile + tiles/2) and feeds both from each staged activation
  // fragment (NT = 2) — the activation re-read from L2 is what the second tile costs (kernels.h), and with several lanes in flight
  // the halved workgroup count costs nothing (tools/gemm_mt2_concurrency.py: 4.03 -> 5.14 TB/s aggregate on 3 streams).
#define VISPEC_GEMM_NT(EPI_, W8_, TILES, BIAS, YPTR, LDY, RPTR, LDR, SPLITS, SCALE)                                     \
  do {                                                                                                                  \
    bool done_ = false;                                                                                                 \
    if constexpr ((MT == 2 && (VISPEC_W8_PAIR || !(W8_))) || (MT == 1 && (W8_) && VISPEC_W8_PAIR_MT1)) {               \
      if ((TILES) % 2 == 0 && !g_mt2_single_block) {                                                                    \
        VISPEC_GEMM(2, EPI_, W8_, dim3((TILES) / 2, SPLITS), (TILES) / 2, BIAS, YPTR, LDY, RPTR, LDR, SPLITS, SCALE);   \
        done_ = true;                                                                                                   \
      }                                                                                                                 \
    }                                                                                                                   \
    if (!done_) VISPEC_GEMM(1, EPI_, W8_, dim3(TILES, SPLITS), 0, BIAS, YPTR, LDY, RPTR, LDR, SPLITS, SCALE);           \
  } while (0)
  if (epi == EPI_SWIGLU) {  // weight in "SwiGLU order" (vispec_pack_weight docs): N/16 32-row tiles
    if (N % 16) return fail("gemm_skinny: SwiGLU needs N %% 16 == 0");
    if (o.norm_w) return fail("gemm_skinny: no fused norm after SwiGLU");
    prof_begin(s, 2, (double)N * K * (o.wscale ? 2.0 : 4.0));
    if (o.xscale) VISPEC_GEMM_NT(EPI_SWIGLU, 2, N / 16, b, o.Y, o.ldy, r, o.ldr, 1, o.wscale);
    else if (o.wscale) VISPEC_GEMM_NT(EPI_SWIGLU, 1, N / 16, b, o.Y, o.ldy, r, o.ldr, 1, o.wscale);
    else VISPEC_GEMM_NT(EPI_SWIGLU, 0, N / 16, b, o.Y, o.ldy, r, o.ldr, 1, nullptr);
    KCHK();
    prof_end(s);
    return 0;
  }
  // split-K when the row blocks alone cannot fill the chip, or when a fused norm is requested (the reduce kernel owns it)
  // (the split factor is the single-tile one whatever the workgroup shape: a row's partial sums then group the same k ranges in
  //  every instantiation, which is what keeps a cohort request bit-identical to the same request run alone)
  int S = 1;
  if (tiles < 256 || o.norm_w) {
    S = choose_split(tiles, KS, (double)32 * K * (o.wscale ? 1.0 : 2.0), o.norm_w != nullptr);
    if (!ctx) S = 1;
  }
  if (force_split > 0) S = force_split;
  if (S == 1 && !o.norm_w) {
    prof_begin(s, epi == EPI_RESIDUAL ? 1 : 0, (double)N * K * (o.wscale ? 1.0 : 2.0));
    if (epi == EPI_RESIDUAL && o.xscale) VISPEC_GEMM_NT(EPI_RESIDUAL, 2, tiles, b, o.Y, o.ldy, r, o.ldr, 1, o.wscale);
    else if (epi == EPI_RESIDUAL && o.wscale) VISPEC_GEMM_NT(EPI_RESIDUAL, 1, tiles, b, o.Y, o.ldy, r, o.ldr, 1, o.wscale);
    else if (epi == EPI_RESIDUAL) VISPEC_GEMM_NT(EPI_RESIDUAL, 0, tiles, b, o.Y, o.ldy, r, o.ldr, 1, nullptr);
    else if (o.xscale) VISPEC_GEMM_NT(EPI_NONE, 2, tiles, b, o.Y, o.ldy, r, o.ldr, 1, o.wscale);
    else if (o.wscale) VISPEC_GEMM_NT(EPI_NONE, 1, tiles, b, o.Y, o.ldy, r, o.ldr, 1, o.wscale);
    else VISPEC_GEMM_NT(EPI_NONE, 0, tiles, b, o.Y, o.ldy, r, o.ldr, 1, nullptr);
    KCHK();
    prof_end(s);
    return 0;
  }
  if (!ctx) return fail("gemm_skinny: split-K needs a ctx (partial workspace)");
  if ((size_t)S * 32 * MT * N > ctx->gemm_part_elems) return fail("gemm_skinny: partial workspace too small");
  prof_begin(s, 3, (double)N * K * (o.wscale ? 1.0 : 2.0));
  if (o.xscale) VISPEC_GEMM_NT(EPI_PARTIAL, 2, tiles, nullptr, ctx->gemm_part, 0, nullptr, 0, S, o.wscale);
  else if (o.wscale) VISPEC_GEMM_NT(EPI_PARTIAL, 1, tiles, nullptr, ctx->gemm_part, 0, nullptr, 0, S, o.wscale);
  else VISPEC_GEMM_NT(EPI_PARTIAL, 0, tiles, nullptr, ctx->gemm_part, 0, nullptr, 0, S, nullptr);
#undef VISPEC_GEMM_NT
#undef VISPEC_GEMM
  KCHK();
  prof_end(s);
  prof_begin(s, 4, 0.0);
  // 512 threads per row: a 1024-thread workgroup needs 4 free waves on every SIMD of ONE CU, which it waits for when other lanes'
  // GEMMs hold the slots (rocprofv3, 4 lanes: 24 us average against 5 us alone); 256 threads make the row pass itself slower
  static const int red_threads = getenv("VISPEC_REDUCE_THREADS") ? atoi(getenv("VISPEC_REDUCE_THREADS")) : 512;
  PLAUNCH(splitk_reduce_kernel, dim3(M), dim3(N >= 2048 ? red_threads : 256), o.normed ? sizeof(float) * N : 0, s, ctx->gemm_part, S, 32 * MT,
                     N, b, epi == EPI_RESIDUAL ? r : nullptr, o.ldr, (bf16_t*)o.Y, o.ldy, (const bf16_t*)o.norm_w, (bf16_t*)o.normed,
                     o.ldn, o.eps, o.m_tile, o.normed ? o.q8 : nullptr, o.sx8);
  KCHK();
  prof_end(s);
  return 0;
}

// gemm_w32_wide8_kernel walks every K-quarter of every split in whole 64-k groups (plus a tail group that re-reads the quarter's last
// 64 k): each quarter must hold at least one group
static bool wide8_ok(int KS, int S, int loads) {
  for (int sp = 0; sp < S; ++sp) {
    const int lo = (int)((long)KS * sp / S), hi = (int)((long)KS * (sp + 1) / S), len = hi - lo;
    for (int q = 0; q < 4; ++q)
      if ((int)((long)len * (q + 1) / 4) - (int)((long)len * q / 4) < loads) return false;
  }
  return true;
}

// Three or four requests per weight pass (csrc/gemm_wide.h): tile t of X / Y / R (rows 32t ..) belongs to request t, rows
// 32t .. 32t + m_tile - 1 are live.  Same decomposition rules as launch_gemm_mt — in particular the SAME split-K factor as the
// single-request launch of the same GEMM, so every row's partial sums group the same k ranges.
static int launch_gemm_wide(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, int n_req, int N, int K,
                            int epi, const GemmOut& o, int force_split, const RopeEpi* re = nullptr) {
  if (n_req < 3 || n_req > 4 || o.m_tile < 1 || o.m_tile > 32) return fail("gemm_wide: 3 or 4 requests of 1..32 rows");
  if (N % 8 || K % 16 || (o.wscale && K % 32)) return fail("gemm_wide: N %% 8 == 0 and K %% 16 (fp8: 32) == 0 required");
  if (epi == EPI_RESIDUAL && !o.R) return fail("gemm_wide: residual epilogue without R");
  const bf16_t *x = (const bf16_t*)X, *w = (const bf16_t*)P, *b = (const bf16_t*)bias, *r = (const bf16_t*)o.R;
  if (o.xscale && (!o.wscale || K % 64)) return fail("gemm_wide: fp8 activations need fp8 weights and K %% 64 == 0");
  const int tiles = epi == EPI_SWIGLU ? N / 16 : (N + 31) / 32, KS = K / (o.xscale ? 64 : (o.wscale ? 32 : 16));
  const int M = 32 * (n_req - 1) + o.m_tile;
  // Row blocks per workgroup (vispec_set_wide_row_blocks): four = one byte of X per byte of W, the form for a GPU that several lanes
  // keep full; two = twice the workgroups at twice the X traffic, which pays on a single stream where four row blocks leave half
  // of the CUs without a workgroup; three for gate|up (230 instead of 172 workgroups).  0 = the smallest of {2, 3, 4} whose grid still
  // runs in one round of CUs.  tools/wide_bench.py; one cohort lane 9.5 -> 8.7 ms per round with 0, four lanes 2064 -> 1980 tok/s.
  const int rb_opt = ctx ? ctx->wide_rb : 4;
  // (experiments: restrict the eight-row-block form to GEMMs of a given size, in 32-row tiles)
  static const int w8_tiles_min = getenv("VISPEC_WIDE8_TILES_MIN") ? atoi(getenv("VISPEC_WIDE8_TILES_MIN")) : 0;
  static const int w8_tiles_max = getenv("VISPEC_WIDE8_TILES_MAX") ? atoi(getenv("VISPEC_WIDE8_TILES_MAX")) : 1 << 30;
#define WIDE_LRB(EPI_, W8_, NL_, RB_, YPTR, LDY, SPLITS)                                                                                 \
  PLAUNCH((gemm_w32_wide_kernel<EPI_, W8_, NL_, 0, RB_>), dim3((tiles + RB_ - 1) / RB_, SPLITS), dim3(RB_ * 256), WIDE_LDS_BYTES, s, x, ldx, w, \
          b, YPTR, LDY, r, o.ldr, o.m_tile, N, K, SPLITS, o.wscale, re ? *re : RopeEpi{}, tiles, o.xscale)
#define WIDE_L(EPI_, W8_, NL_, YPTR, LDY, SPLITS)                                                                                       \
  do {                                                                                                                                  \
    int rb_ = rb_opt;                                                                                                                   \
    /* 84 = what several lanes per GPU run: eight row blocks for bf16 weights (+4 % on the bf16 lines in four same-box A/Bs; restricting  \
       the eight-row-block form to the GEMMs whose four-row-block grid is small or spills into a second round keeps only half of that),    \
       four for fp8 weights with bf16 activations (their eight-row-block kernel sits on the matrix pipe: twice the MFMAs per weight byte   \
       plus the up-conversion on two waves per SIMD, 29 GB/s per CU instead of the 54 of the bf16 one), eight again for W8A8 (a quarter   \
       of the MFMAs: +2.6 % over four on the line) — profiles/README.md, round 4 */                                                        \
    if (rb_ == 84) rb_ = (W8_) == 1 ? 4 : 8;                                                                                              \
    if ((W8_) == 2 && rb_ != 8) rb_ = 4;  /* fp8 activations: four or eight row blocks */                                                  \
    if (rb_ == 8 && tiles >= w8_tiles_min && tiles <= w8_tiles_max && wide8_ok(KS, SPLITS, (W8_) ? 2 : 4)) {  /* eight row blocks per workgroup */ \
      PLAUNCH((gemm_w32_wide8_kernel<EPI_, W8_, NL_>), dim3((tiles + 7) / 8, SPLITS), dim3(512), (wide8_lds_bytes<W8_>()), s, x, ldx, w, b, \
              YPTR, LDY, r, o.ldr, o.m_tile, N, K, SPLITS, o.wscale, re ? *re : RopeEpi{}, tiles, o.xscale);                            \
      break;                                                                                                                            \
    }                                                                                                                                   \
    if (rb_ == 8) rb_ = 4;                                                                                                              \
    if (rb_ == 0) rb_ = ((tiles + 1) / 2) * (SPLITS) <= 256 ? 2 : (((tiles + 2) / 3) * (SPLITS) <= 256 ? 3 : 4);  /* most workgroups in one round of CUs */ \
    if constexpr ((W8_) == 2) WIDE_LRB(EPI_, W8_, NL_, 4, YPTR, LDY, SPLITS);                                                           \
    else if (rb_ == 2) WIDE_LRB(EPI_, W8_, NL_, 2, YPTR, LDY, SPLITS);                                                                  \
    else if (rb_ == 3) WIDE_LRB(EPI_, W8_, NL_, 3, YPTR, LDY, SPLITS);                                                                  \
    else WIDE_LRB(EPI_, W8_, NL_, 4, YPTR, LDY, SPLITS);                                                                                \
  } while (0)
#define WIDE_D(EPI_, YPTR, LDY, SPLITS)                                                                       \
  do {                                                                                                        \
    if (o.xscale) { if (n_req == 3) WIDE_L(EPI_, 2, 3, YPTR, LDY, SPLITS); else WIDE_L(EPI_, 2, 4, YPTR, LDY, SPLITS); }           \
    else if (o.wscale) { if (n_req == 3) WIDE_L(EPI_, 1, 3, YPTR, LDY, SPLITS); else WIDE_L(EPI_, 1, 4, YPTR, LDY, SPLITS); }      \
    else { if (n_req == 3) WIDE_L(EPI_, 0, 3, YPTR, LDY, SPLITS); else WIDE_L(EPI_, 0, 4, YPTR, LDY, SPLITS); }                    \
  } while (0)
  if (epi == EPI_ROPE) {
    if (!re) return fail("gemm_wide: rope epilogue without its arguments");
    prof_begin(s, PROF_QKV_ROPE, (double)N * K * (o.wscale ? 1.0 : 2.0));
    WIDE_D(EPI_ROPE, o.Y, o.ldy, 1);
    KCHK();
    prof_end(s);
    return 0;
  }
  if (epi == EPI_SWIGLU) {
    if (N % 16) return fail("gemm_wide: SwiGLU needs N %% 16 == 0");
    if (o.norm_w) return fail("gemm_wide: no fused norm after SwiGLU");
    prof_begin(s, 2, (double)N * K * (o.wscale ? 2.0 : 4.0));
    WIDE_D(EPI_SWIGLU, o.Y, o.ldy, 1);
    KCHK();
    prof_end(s);
    return 0;
  }
  int S = 1;
  if (tiles < 256 || o.norm_w) {
    S = choose_split(tiles, KS, (double)32 * K * (o.wscale ? 1.0 : 2.0), o.norm_w != nullptr);
    if (!ctx) S = 1;
  }
  if (force_split > 0) S = force_split;
  if (S == 1 && !o.norm_w) {
    prof_begin(s, epi == EPI_RESIDUAL ? 1 : 0, (double)N * K * (o.wscale ? 1.0 : 2.0));
    if (epi == EPI_RESIDUAL) WIDE_D(EPI_RESIDUAL, o.Y, o.ldy, 1); else WIDE_D(EPI_NONE, o.Y, o.ldy, 1);
    KCHK();
    prof_end(s);
    return 0;
  }
  if (!ctx) return fail("gemm_wide: split-K needs a ctx (partial workspace)");
  if ((size_t)S * WIDE_MPAD * N > ctx->gemm_part_elems) return fail("gemm_wide: partial workspace too small");
  prof_begin(s, 3, (double)N * K * (o.wscale ? 1.0 : 2.0));
  {
    const bf16_t* b_keep = b;
    b = nullptr;  // bias belongs to the reduce
    WIDE_D(EPI_PARTIAL, ctx->gemm_part, 0, S);
    b = b_keep;
  }
#undef WIDE_D
#undef WIDE_L
#undef WIDE_LRB
  KCHK();
  prof_end(s);
  prof_begin(s, 4, 0.0);
  static const int red_threads = getenv("VISPEC_REDUCE_THREADS") ? atoi(getenv("VISPEC_REDUCE_THREADS")) : 512;
  PLAUNCH(splitk_reduce_kernel, dim3(M), dim3(N >= 2048 ? red_threads : 256), o.normed ? sizeof(float) * N : 0, s, ctx->gemm_part, S, WIDE_MPAD, N, b,
          epi == EPI_RESIDUAL ? r : nullptr, o.ldr, (bf16_t*)o.Y, o.ldy, (const bf16_t*)o.norm_w, (bf16_t*)o.normed, o.ldn, o.eps, o.m_tile,
          o.normed ? o.q8 : nullptr, o.sx8);
  KCHK();
  prof_end(s);
  return 0;
}

// Five to eight requests per weight pass (csrc/gemm_c8.h): tile t of X / Y / R (rows 32t ..) belongs to request t, rows 32t .. 32t + m_tile - 1
// are live.  ONE accumulator chain per output element over the split's K range ("the c8 order"): a row does not depend on what shares its
// launch, but differs from the single-request kernel's four-quarter order in fp32 rounding.  Split-K factor = the single-request policy's.
static int launch_gemm_c8(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, int n_req, int N, int K,
                          int epi, const GemmOut& o, int force_split, const RopeEpi* re = nullptr) {
  if (n_req < 5 || n_req > MAX_COHORT || o.m_tile < 1 || o.m_tile > 32) return fail("gemm_c8: 5..8 requests of 1..32 rows");
  if (N % 8 || K % 16 || (o.wscale && K % 32)) return fail("gemm_c8: N %% 8 == 0 and K %% 16 (fp8: 32) == 0 required");
  if (epi == EPI_RESIDUAL && !o.R) return fail("gemm_c8: residual epilogue without R");
  const bf16_t *x = (const bf16_t*)X, *w = (const bf16_t*)P, *b = (const bf16_t*)bias, *r = (const bf16_t*)o.R;
  if (o.xscale && (!o.wscale || K % 64)) return fail("gemm_c8: fp8 activations need fp8 weights and K %% 64 == 0");
  const int tiles = epi == EPI_SWIGLU ? N / 16 : (N + 31) / 32, KS = K / (o.xscale ? 64 : (o.wscale ? 32 : 16));
  const int M = 32 * (n_req - 1) + o.m_tile;
  static const int c8_force_small = getenv("VISPEC_C8_SMALL") ? atoi(getenv("VISPEC_C8_SMALL")) : 0;  // tests: the fragment-shaped form at every size
#define C8_L(EPI_, W8_, YPTR, LDY, SPLITS)                                                                                              \
  do {                                                                                                                                  \
    if (c8_fast_ok(K, SPLITS, W8_) && !c8_force_small)                                                                                  \
      PLAUNCH((gemm_w32_c8_kernel<EPI_, W8_>), dim3((tiles + 7) / 8, SPLITS), dim3(512), C8_LDS_BYTES, s, x, ldx, w, b, YPTR, LDY, r, o.ldr,  \
              o.m_tile, n_req, N, K, SPLITS, o.wscale, re ? *re : RopeEpi{}, tiles, o.xscale);                                           \
    else                                                                                                                                \
      PLAUNCH((gemm_w32_c8_small_kernel<EPI_, W8_>), dim3(tiles, SPLITS), dim3(64), 0, s, x, ldx, w, b, YPTR, LDY, r, o.ldr, o.m_tile,    \
              n_req, N, K, SPLITS, o.wscale, re ? *re : RopeEpi{}, tiles, o.xscale);                                                     \
  } while (0)
#define C8_D(EPI_, YPTR, LDY, SPLITS)                                                                         \
  do {                                                                                                        \
    if (o.xscale) C8_L(EPI_, 2, YPTR, LDY, SPLITS);                                                           \
    else if (o.wscale) C8_L(EPI_, 1, YPTR, LDY, SPLITS);                                                      \
    else C8_L(EPI_, 0, YPTR, LDY, SPLITS);                                                                    \
  } while (0)
  if (epi == EPI_ROPE) {
    if (!re) return fail("gemm_c8: rope epilogue without its arguments");
    prof_begin(s, PROF_QKV_ROPE, (double)N * K * (o.wscale ? 1.0 : 2.0));
    C8_D(EPI_ROPE, o.Y, o.ldy, 1);
    KCHK();
    prof_end(s);
    return 0;
  }
  if (epi == EPI_SWIGLU) {
    if (N % 16) return fail("gemm_c8: SwiGLU needs N %% 16 == 0");
    if (o.norm_w) return fail("gemm_c8: no fused norm after SwiGLU");
    prof_begin(s, 2, (double)N * K * (o.wscale ? 2.0 : 4.0));
    C8_D(EPI_SWIGLU, o.Y, o.ldy, 1);
    KCHK();
    prof_end(s);
    return 0;
  }
  int S = 1;
  if (tiles < 256 || o.norm_w) {
    S = choose_split(tiles, KS, (double)32 * K * (o.wscale ? 1.0 : 2.0), o.norm_w != nullptr);
    if (!ctx) S = 1;
  }
  if (force_split > 0) S = force_split;
  static const int c8_s_env = getenv("VISPEC_C8_SPLIT") ? atoi(getenv("VISPEC_C8_SPLIT")) : 0;  // experiments: the split factor of the split-K GEMMs
  if (c8_s_env > 0 && S > 1 && force_split <= 0) S = std::min(c8_s_env, std::max(1, KS / 8));
  if (S == 1 && !o.norm_w) {
    prof_begin(s, epi == EPI_RESIDUAL ? 1 : 0, (double)N * K * (o.wscale ? 1.0 : 2.0));
    if (epi == EPI_RESIDUAL) C8_D(EPI_RESIDUAL, o.Y, o.ldy, 1); else C8_D(EPI_NONE, o.Y, o.ldy, 1);
    KCHK();
    prof_end(s);
    return 0;
  }
  if (!ctx) return fail("gemm_c8: split-K needs a ctx (partial workspace)");
  if ((size_t)S * C8_MPAD * N > ctx->gemm_part_elems) return fail("gemm_c8: partial workspace too small");
  prof_begin(s, 3, (double)N * K * (o.wscale ? 1.0 : 2.0));
  {
    const bf16_t* b_keep = b;
    b = nullptr;  // bias belongs to the reduce
    C8_D(EPI_PARTIAL, ctx->gemm_part, 0, S);
    b = b_keep;
  }
#undef C8_D
#undef C8_L
  KCHK();
  prof_end(s);
  prof_begin(s, 4, 0.0);
  static const int red_threads = getenv("VISPEC_REDUCE_THREADS") ? atoi(getenv("VISPEC_REDUCE_THREADS")) : 512;
  PLAUNCH(splitk_reduce_kernel, dim3(M), dim3(N >= 2048 ? red_threads : 256), o.normed ? sizeof(float) * N : 0, s, ctx->gemm_part, S, C8_MPAD, N, b,
          epi == EPI_RESIDUAL ? r : nullptr, o.ldr, (bf16_t*)o.Y, o.ldy, (const bf16_t*)o.norm_w, (bf16_t*)o.normed, o.ldn, o.eps, o.m_tile,
          o.normed ? o.q8 : nullptr, o.sx8);
  KCHK();
  prof_end(s);
  return 0;
}

// m_tile > 0: cohort mode, M = 32 (n_req - 1) + m_tile with n_req in [2,8] requests of m_tile rows each (tile t = request t)
static int launch_gemm_ex(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, int M, int N, int K,
                          int epi, const GemmOut& o, int force_split = -1) {
  if (o.m_tile < 0) {  // slab mode: the requests' <= 8 live rows share one activation tile
    const int rows = -o.m_tile, n_req = (M - rows) / 8 + 1;
    if (rows > 8 || (M - rows) % 8 || n_req < 2 || n_req > MAX_COHORT) return fail("gemm_skinny: slab mode wants M = 8 (n - 1) + rows, rows <= 8, n in [2,8]");
    if (n_req > 4) return launch_gemm_mt<2>(ctx, s, X, ldx, P, bias, M, N, K, epi, o, force_split);  // five to eight requests: two slab tiles, one weight pass
    return launch_gemm_mt<1>(ctx, s, X, ldx, P, bias, M, N, K, epi, o, force_split);
  }
  if (o.m_tile > 0) {
    const int n_req = (M - o.m_tile) / 32 + 1;
    if (o.m_tile > 32 || (M - o.m_tile) % 32 || n_req < 2 || n_req > MAX_COHORT) return fail("gemm_skinny: cohort mode wants M = 32 (n - 1) + m_tile, n in [2,8]");
    if (n_req > 4) return launch_gemm_c8(ctx, s, X, ldx, P, bias, n_req, N, K, epi, o, force_split);
    if (n_req > 2) return launch_gemm_wide(ctx, s, X, ldx, P, bias, n_req, N, K, epi, o, force_split);
  }
  if (M <= 32) return launch_gemm_mt<1>(ctx, s, X, ldx, P, bias, M, N, K, epi, o, force_split);
  if (M <= 64) return launch_gemm_mt<2>(ctx, s, X, ldx, P, bias, M, N, K, epi, o, force_split);
  return fail("gemm_skinny: M must be in [1,64]");
}

// legacy-shaped helper used by most call sites
static int launch_gemm(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, void* Y, int ldy,
                       const void* R, int ldr, int M, int N, int K, int epi, const void* wscale = nullptr, int m_tile = 0) {
  GemmOut o;
  o.wscale = (const float*)wscale;
  o.m_tile = m_tile;
  o.Y = Y; o.ldy = ldy; o.R = R; o.ldr = ldr;
  return launch_gemm_ex(ctx, s, X, ldx, P, bias, M, N, K, epi, o);
}

static int launch_rmsnorm(hipStream_t s, const void* X, const void* w, void* Y, int M, int D, float eps) {
  if (D % 8) return fail("rmsnorm: D %% 8");
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(M), dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)w, (bf16_t*)Y, D, eps);
  KCHK();
  return 0;
}

// 8 threads per (row, head): thread t rotates d = 8t .. 8t+7 against d + 64 with 16-byte accesses (round 3: the scalar form — one bf16 pair
// per thread — took 58 us per layer of a 2704-row prefill for 130 MB of traffic).  Same arithmetic, element for element.
__global__ __launch_bounds__(256) void rope_append2_kernel(bf16_t* __restrict__ qkv, int M, int H, int H_kv,
                                                           const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                                                           PosSpec ps, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                           int s_max, int do_rope) {
  constexpr int HD = 128, HALF = 64;
  const int NH = H + 2 * H_kv;
  const int item = blockIdx.x * 32 + (threadIdx.x >> 3), t = threadIdx.x & 7;
  if (item >= M * NH) return;
  const int m = item / NH, h = item - m * NH, d = 8 * t;
  const int ld = NH * HD;
  bf16_t* x = qkv + (size_t)m * ld + (size_t)h * HD;
  const int row = (ps.kv_base ? *ps.kv_base : 0) + ps.kv_add + m;
  const uint4 lo = *reinterpret_cast<const uint4*>(x + d), hi = *reinterpret_cast<const uint4*>(x + d + HALF);
  if (h >= H + H_kv) {
    bf16_t* dst = vc + ((size_t)(h - H - H_kv) * s_max + row) * HD;
    *reinterpret_cast<uint4*>(dst + d) = lo;
    *reinterpret_cast<uint4*>(dst + d + HALF) = hi;
    return;
  }
  uint4 olo = lo, ohi = hi;
  if (do_rope) {
    const int pos = (ps.base ? *ps.base : 0) + (ps.base2 ? *ps.base2 : 0) + ps.add + (ps.off ? ps.off[m] : (ps.row ? m : 0));
    const uint4 cv = *reinterpret_cast<const uint4*>(cosT + (size_t)pos * HD + d), sv = *reinterpret_cast<const uint4*>(sinT + (size_t)pos * HD + d);
    const bf16_t *e1 = reinterpret_cast<const bf16_t*>(&lo), *e2 = reinterpret_cast<const bf16_t*>(&hi);
    const bf16_t *ce = reinterpret_cast<const bf16_t*>(&cv), *se = reinterpret_cast<const bf16_t*>(&sv);
    float o1[8], o2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x1 = bf2f(e1[i]), x2 = bf2f(e2[i]), c = bf2f(ce[i]), sn = bf2f(se[i]);
      o1[i] = rdbf(rdbf(x1 * c) + rdbf(-x2 * sn));
      o2[i] = rdbf(rdbf(x2 * c) + rdbf(x1 * sn));
    }
    olo = make_uint4(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]), pack2(o1[4], o1[5]), pack2(o1[6], o1[7]));
    ohi = make_uint4(pack2(o2[0], o2[1]), pack2(o2[2], o2[3]), pack2(o2[4], o2[5]), pack2(o2[6], o2[7]));
  }
  bf16_t* dst = (h < H) ? x : kc + ((size_t)(h - H) * s_max + row) * HD;
  *reinterpret_cast<uint4*>(dst + d) = olo;
  *reinterpret_cast<uint4*>(dst + d + HALF) = ohi;
}

static int launch_rope(hipStream_t s, void* qkv, int M, int H, int H_kv, const void* cosT, const void* sinT, PosSpec ps,
                       void* kc, void* vc, int s_max, int do_rope) {
  const int items = M * (H + 2 * H_kv);
  hipLaunchKernelGGL(rope_append2_kernel, dim3((items + 31) / 32), dim3(256), 0, s, (bf16_t*)qkv, M, H, H_kv, (const bf16_t*)cosT,
                     (const bf16_t*)sinT, ps, (bf16_t*)kc, (bf16_t*)vc, s_max, do_rope);
  KCHK();
  return 0;
}

// q|k|v projection + rotary + KV append.  When split-K does not pay (>= 128 row blocks) the three run as ONE launch
// (EPI_ROPE; the weight must have been packed in rope order — vispec_qkv_rope_fused() tells the loader); otherwise the
// split-K GEMM is followed by rope_append2_kernel on a naturally ordered weight.
// (un-split: with >= 128 row blocks one launch beats split-K + reduce + rotary launches even on a half-filled chip — choose_split)
static bool qkv_rope_fused(int n_rows) { return n_rows % 128 == 0 && n_rows / 32 >= 128; }
struct QkvReq {  // per-request part of a q|k|v projection: positions and the cache it appends to
  PosSpec ps;
  void* kc = nullptr;
  void* vc = nullptr;
};
// n_req = 1: rows 0 .. M-1 of X belong to rq[0].  n_req = 2..4 (cohort): request t owns rows 32t .. 32t + M - 1, the weights are
// streamed once for all of them.
static int launch_qkv_rope(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, const void* wscale,
                           void* qkv, int M, int H, int H_kv, int K, const void* cosT, const void* sinT, const QkvReq* rq, int n_req,
                           int s_max, bool slab = false, const float* xscale = nullptr /* X = e4m3 codes, ldx in 2-byte units */,
                           bool two_tiles = false /* cohort of 33..64-row requests: request t owns rows 64t .. 64t + M - 1 (tiles 2t, 2t + 1) */) {
  const int N = (H + 2 * H_kv) * 128;
  two_tiles = two_tiles && n_req >= 2;
  if (two_tiles && (M < 1 || M > 64 || n_req > 4 || slab)) return fail("gemm_qkv_rope: two tiles per request: 2..4 requests of 1..64 rows");
  slab = slab && n_req >= 2 && M <= 8;  // slab: the requests' rows packed into one activation tile (request t still at rows 32t .. of X / qkv)
  const int n_tiles2 = M > 32 ? 2 * n_req : 2 * n_req - 1;  // (two_tiles, M <= 32: the last request's second tile is not part of the launch)
  const int m_tile = slab ? -M : (two_tiles ? std::min(M, 32) : (n_req >= 2 ? M : 0));
  const int Mk = slab ? 8 * (n_req - 1) + M : (two_tiles ? 32 * (n_tiles2 - 1) + std::min(M, 32) : (n_req >= 2 ? 32 * (n_req - 1) + M : M));
  if (!qkv_rope_fused(N)) {
    GemmOut o0;
    o0.wscale = (const float*)wscale; o0.xscale = xscale; o0.m_tile = m_tile; o0.Y = qkv; o0.ldy = N;
    if (launch_gemm_ex(ctx, s, X, ldx, P, bias, Mk, N, K, EPI_NONE, o0)) return -1;
    for (int t = 0; t < n_req; ++t)
      if (launch_rope(s, (bf16_t*)qkv + (size_t)(two_tiles ? 64 : 32) * t * N, M, H, H_kv, cosT, sinT, rq[t].ps, rq[t].kc, rq[t].vc, s_max, 1)) return -1;
    return 0;
  }
  if (M < 1 || Mk > ROWS || (n_req == 1 && M > 64) || (n_req > 1 && !two_tiles && M > 32))
    return fail("gemm_qkv_rope: M must be in [1,64] (cohort: [1,32], or 33..64 as two tiles per request)");

  if (K % 16 || (wscale && K % 32)) return fail("gemm_qkv_rope: K %% 16 (fp8: 32) == 0 required");
  RopeEpi re;
  re.cosT = (const bf16_t*)cosT; re.sinT = (const bf16_t*)sinT;
  re.s_max = s_max; re.H = H; re.H_kv = H_kv;
  if (two_tiles) {  // tile 2t = rows 0..31 of request t, tile 2t + 1 = its rows 32 .. M-1: positions and cache rows continue at +32
    for (int t = 0; t < n_req; ++t)
      for (int h2 = 0; h2 < 2; ++h2) {
        PosSpec ps = rq[t].ps;
        if (h2) { if (ps.off) ps.off += 32; else if (ps.row) ps.add += 32; ps.kv_add += 32; }
        re.ps[2 * t + h2] = ps; re.kc[2 * t + h2] = (bf16_t*)rq[t].kc; re.vc[2 * t + h2] = (bf16_t*)rq[t].vc;
        re.rows[2 * t + h2] = (unsigned char)(h2 ? (M > 32 ? M - 32 : 255 /* dead tile */) : std::min(M, 32));
      }
    GemmOut o;
    o.wscale = (const float*)wscale; o.xscale = xscale; o.m_tile = m_tile; o.Y = qkv; o.ldy = N;
    if (n_tiles2 > 4) return launch_gemm_c8(ctx, s, X, ldx, P, bias, n_tiles2, N, K, EPI_ROPE, o, -1, &re);
    return launch_gemm_wide(ctx, s, X, ldx, P, bias, n_tiles2, N, K, EPI_ROPE, o, -1, &re);
  }
  for (int t = 0; t < n_req; ++t) { re.ps[t] = rq[t].ps; re.kc[t] = (bf16_t*)rq[t].kc; re.vc[t] = (bf16_t*)rq[t].vc; }
  if (n_req > 2 && !slab) {  // three or four requests: the wide-cohort kernel; five to eight: the cohort-8 kernel
    GemmOut o;
    o.wscale = (const float*)wscale; o.xscale = xscale; o.m_tile = m_tile; o.Y = qkv; o.ldy = N;
    if (n_req > 4) return launch_gemm_c8(ctx, s, X, ldx, P, bias, n_req, N, K, EPI_ROPE, o, -1, &re);
    return launch_gemm_wide(ctx, s, X, ldx, P, bias, n_req, N, K, EPI_ROPE, o, -1, &re);
  }
  prof_begin(s, PROF_QKV_ROPE, (double)N * K * (wscale ? 1.0 : 2.0));
#define VISPEC_QKV(W8_, MT_, NT_)                                                                                                         \
  PLAUNCH((gemm_w32_kernel<NT_, EPI_ROPE, 4, 4, 0, W8_, MT_>), dim3(N / 32 / NT_, 1), dim3(256), (gemm_w32_lds_bytes<NT_, 4, 4, MT_>()), s, \
                     (const bf16_t*)X, ldx, (const bf16_t*)P, (NT_ == 2 ? N / 64 : 0), (const bf16_t*)bias, qkv, N, nullptr, 0, Mk, N, K, 1,    \
                     (const float*)wscale, re, m_tile, xscale)
  if (xscale && (slab || K % 64)) return fail("gemm_qkv_rope: fp8 activations: no slab form, K %% 64 == 0");
  if (slab) {
#define VISPEC_QKV_SLAB(W8_, NT_)                                                                                                          \
  PLAUNCH((gemm_w32_kernel<NT_, EPI_ROPE, 4, 4, 0, W8_, 1, true>), dim3(N / 32 / NT_, 1), dim3(256), (gemm_w32_lds_bytes<NT_, 4, 4, 1>()), s, \
                     (const bf16_t*)X, ldx, (const bf16_t*)P, (NT_ == 2 ? N / 64 : 0), (const bf16_t*)bias, qkv, N, nullptr, 0, Mk, N, K, 1,    \
                     (const float*)wscale, re, m_tile, (const float*)nullptr)
    if (n_req > 4) {  // five to eight requests: two slab tiles (N %% 128 == 0: the tile count is even -> the paired form)
#define VISPEC_QKV_SLAB2(W8_)                                                                                                              \
  PLAUNCH((gemm_w32_kernel<2, EPI_ROPE, 4, 4, 0, W8_, 2, true>), dim3(N / 64, 1), dim3(256), (gemm_w32_lds_bytes<2, 4, 4, 2>()), s,             \
                     (const bf16_t*)X, ldx, (const bf16_t*)P, N / 64, (const bf16_t*)bias, qkv, N, nullptr, 0, Mk, N, K, 1,                  \
                     (const float*)wscale, re, m_tile, (const float*)nullptr)
      if (wscale) VISPEC_QKV_SLAB2(true); else VISPEC_QKV_SLAB2(false);
#undef VISPEC_QKV_SLAB2
    } else if (wscale && VISPEC_W8_PAIR_MT1 && !g_mt2_single_block) VISPEC_QKV_SLAB(true, 2); else if (wscale) VISPEC_QKV_SLAB(true, 1); else VISPEC_QKV_SLAB(false, 1);
#undef VISPEC_QKV_SLAB
  } else if (xscale) {  // e4m3 activations: the paired form (two row blocks per workgroup), one or two activation tiles
    if (Mk <= 32) VISPEC_QKV(2, 1, 2); else VISPEC_QKV(2, 2, 2);
  } else if (Mk <= 32) { if (wscale && VISPEC_W8_PAIR_MT1 && !g_mt2_single_block) VISPEC_QKV(true, 1, 2); else if (wscale) VISPEC_QKV(true, 1, 1); else VISPEC_QKV(false, 1, 1); }
  else if (g_mt2_single_block || (wscale && !VISPEC_W8_PAIR)) { if (wscale) VISPEC_QKV(true, 2, 1); else VISPEC_QKV(false, 2, 1); }
  else { if (wscale) VISPEC_QKV(true, 2, 2); else VISPEC_QKV(false, 2, 2); }  // N %% 128 == 0: the tile count is even
#undef VISPEC_QKV
  KCHK();
  prof_end(s);
  return 0;
}
static int launch_qkv_rope1(vispec_ctx* ctx, hipStream_t s, const void* X, int ldx, const void* P, const void* bias, const void* wscale,
                            void* qkv, int M, int H, int H_kv, int K, const void* cosT, const void* sinT, PosSpec ps, void* kc, void* vc,
                            int s_max) {
  QkvReq rq;
  rq.ps = ps; rq.kc = kc; rq.vc = vc;
  return launch_qkv_rope(ctx, s, X, ldx, P, bias, wscale, qkv, M, H, H_kv, K, cosT, sinT, &rq, 1, s_max);
}

// prefix = (prefix_dev ? *prefix_dev : 0) + prefix_add is folded by a tiny helper kernel into a scratch int when needed
__global__ void add_scalar_kernel(const int* src, int add, int* dst) { *dst = (src ? *src : 0) + add; }

struct AttnCall {  // per-request part of an attention call
  vispec_ctx* ctx;  // owner of the partial workspace
  const void* q;
  const void* kc;
  const void* vc;
  const int* prefix_dev;
  const unsigned long long* mask;
  void* out;
};
// n = 1, or 2..4 for a cohort: all requests' attention runs in ONE partial launch and ONE reduce launch (blockIdx.z)
static int launch_attention_n(hipStream_t s, const AttnCall* calls, int n, int ldq, int s_max, int H, int H_kv, int M, int tail, int ldo,
                              int eager, int max_keys) {
  if (M < 1 || M > 64) return fail("tree_attention: M must be in [1,64]");
  if (tail < 0 || tail > 64) return fail("tree_attention: tail must be in [0,64]");
  const int MT = (M + 31) / 32, NQT = (H / H_kv) * MT;
  static const int kpw_env = getenv("VISPEC_ATT_KPW") ? atoi(getenv("VISPEC_ATT_KPW")) : 0;  // tuning experiments only
  // 512 keys (four 128-key chunks) per workgroup: round 2 ran 256; with the four requests of a wide cohort in one launch the 256-key
  // grid is three rounds of workgroups per CU slot and twice the partial tiles to merge (4 lanes x cohort 4: 1980 -> 2070 tok/s; one
  // request alone: 5.29 -> 5.21 ms per round)
  // cohorts of five to eight requests (round 5): 768 keys — eight requests are 1 792 workgroups at 512 keys (seven rounds of the CU slots two
  // workgroups hold each), a third fewer partial tiles to write and merge at 768: 3369 / 3379 -> 3418 tok/s on the same box, while ONE request
  // alone loses its parallelism (4.90 -> 5.01 ms per round: single requests and cohorts of up to four keep 512, and with it their split
  // boundaries — the bit-identity of a cohort row with the single-request row).  All cohorts of 5..8 share the 768-key boundaries, so a request's
  // result still does not depend on the cohort's size or composition (profiles/r05_env_knobs_sweep.txt).
  static const int kpw_c8 = getenv("VISPEC_ATT_KPW_C8") ? atoi(getenv("VISPEC_ATT_KPW_C8")) : 768;
  int kpw = (kpw_env >= ATT2_CHUNK && kpw_env % ATT2_CHUNK == 0) ? kpw_env : ((n > 4 && kpw_c8 >= ATT2_CHUNK && kpw_c8 % ATT2_CHUNK == 0) ? kpw_c8 : 512);
  if (max_keys < 1) max_keys = 1;
  if (max_keys > s_max) max_keys = s_max;
  // keys per workgroup from the CACHE CAPACITY, never from the requests' current lengths: the split boundaries (and with them the order in
  // which partial results are merged) of a request are then the same whatever it shares a launch with — a cohort request stays
  // bit-identical to the same request alone at every context length
  while ((s_max + kpw - 1) / kpw > 64) kpw *= 2;
  const int nsplit = (max_keys + kpw - 1) / kpw;
  // Round 5 built the merge of a (request, head, q-tile)'s key splits by the LAST workgroup of the partial launch to arrive (one launch per
  // attention call less; the stand-alone kernel's arithmetic, element for element: every test passes in both forms) — and measured it
  // SLOWER on the same box, twice: with a release fence per workgroup 3252 -> 2556 tok/s (one request's round 4.97 -> 5.27 ms), with
  // write-through (sc1) partial stores 3254 -> 3062 tok/s (4.98 -> 5.14 ms): publishing 16 KB per workgroup 1 792 times a launch costs more
  // than the 5 us launch it removes (profiles/r05_ab_attention_fused_merge.txt).  Off; VISPEC_ATT_FUSED_MERGE=1 runs it.
  static const bool fused_merge = getenv("VISPEC_ATT_FUSED_MERGE") && atoi(getenv("VISPEC_ATT_FUSED_MERGE")) != 0;
  if (fused_merge && H_kv * NQT > 512) return fail("tree_attention: more (head, q-tile) pairs than merge counters");
  AttnArgs args{};
  for (int t = 0; t < n; ++t) {
    if ((size_t)H_kv * NQT * nsplit > calls[t].ctx->part_cap_tiles) return fail("tree_attention: partial workspace too small");
    AttnReq& r = args.r[t];
    r.Q = (const bf16_t*)calls[t].q; r.Kc = (const bf16_t*)calls[t].kc; r.Vc = (const bf16_t*)calls[t].vc;
    r.prefix_dev = calls[t].prefix_dev; r.mask = calls[t].mask; r.part_o = calls[t].ctx->part_o; r.part_ml = calls[t].ctx->part_ml;
    r.out = (bf16_t*)calls[t].out;
    r.cnt = fused_merge ? calls[t].ctx->att_cnt : nullptr;
  }
  dim3 grid(nsplit, H_kv, NQT * n), block(256);
  prof_begin(s, eager ? PROF_ATT_PARTIAL : PROF_ATT_PARTIAL_DRAFT, 0.0);
  if (eager) PLAUNCH(tree_attn2_partial_kernel<true>, grid, block, ATT2_LDS_BYTES, s, args, ldq, s_max, H, H_kv, M, tail, kpw, nsplit, NQT, ldo);
  else PLAUNCH(tree_attn2_partial_kernel<false>, grid, block, ATT2_LDS_BYTES, s, args, ldq, s_max, H, H_kv, M, tail, kpw, nsplit, NQT, ldo);
  KCHK();
  prof_end(s);
  if (fused_merge) return 0;
  prof_begin(s, eager ? PROF_ATT_REDUCE : PROF_ATT_REDUCE_DRAFT, 0.0);
  PLAUNCH(tree_attn_reduce_kernel, dim3(H * MT, 4, n), dim3(256), 0, s, args, H, H_kv, M, tail, kpw, nsplit, ldo);
  KCHK();
  prof_end(s);
  return 0;
}
static int launch_attention(vispec_ctx* ctx, hipStream_t s, const void* q, int ldq, const void* kc, const void* vc, int s_max,
                            int H, int H_kv, int M, const int* prefix_dev, int tail, const unsigned long long* mask, void* out,
                            int ldo, int eager, int max_keys) {
  const AttnCall call{ctx, q, kc, vc, prefix_dev, mask, out};
  return launch_attention_n(s, &call, 1, ldq, s_max, H, H_kv, M, tail, ldo, eager, max_keys);
}

static int launch_gather(hipStream_t s, const void* table, int ld_t, const int* idx, int idx_off, const int* idx_base_dev,
                         void* out, int ld_o, int rows, int D) {
  if (D % 8) return fail("gather: D %% 8");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, s, (const bf16_t*)table, ld_t, idx, idx_off, idx_base_dev,
                     (bf16_t*)out, ld_o, D);
  KCHK();
  return 0;
}
static int launch_bcast(hipStream_t s, const void* vec, void* out, int ld_o, int rows, int D) {
  hipLaunchKernelGGL(bcast_row_kernel, dim3(rows), dim3(256), 0, s, (const bf16_t*)vec, (bf16_t*)out, ld_o, D);
  KCHK();
  return 0;
}

static int launch_lstopk(vispec_ctx* ctx, hipStream_t s, const void* logits, int ld, int M, int V, int k, int* out_idx, float* out_logp) {
  if (k < 1 || k > TOPK_MAX) return fail("logsoftmax_topk: k must be in [1,16]");
  if (M < 1 || M > 64) return fail("logsoftmax_topk: M must be in [1,64]");
  // one launch: a 1024-thread workgroup per row keeps the row in registers (8 * NV logits per thread)
  static const int row_max_v = getenv("VISPEC_LSTK_ROW_MAX_V") ? atoi(getenv("VISPEC_LSTK_ROW_MAX_V")) : 1024 * 8 * 20;  // experiments
  // large vocabularies (Qwen2.5-VL's 152 064): a row is cut into chunks of <= 32 768 logits, one workgroup each — three short launches
  // instead of one workgroup walking 300 KB on a single CU (tools/lstk_bench.py: 62-73 us -> see DESIGN.md)
  static const int chunk_min_v = getenv("VISPEC_LSTK_CHUNK_MIN_V") ? atoi(getenv("VISPEC_LSTK_CHUNK_MIN_V")) : 65536;
  if (ctx && V % 8 == 0 && ld % 8 == 0 && V > chunk_min_v && V <= 32768 * LSTK_CHUNKS) {
    int C = (V + 32767) / 32768;
    if (C < 8) C = 8;
    const int chunk = ((V / 8 + C - 1) / C) * 8, nv = (chunk / 8 + 1023) / 1024;
    const bf16_t* lg = (const bf16_t*)logits;
#define LSTK2(NV_)                                                                                                                 \
  do {                                                                                                                             \
    hipLaunchKernelGGL(lstk2_stats_kernel<NV_>, dim3(M, C), dim3(1024), 0, s, lg, ld, V, chunk, ctx->lstk_stats);                  \
    hipLaunchKernelGGL(lstk2_select_kernel<NV_>, dim3(M, C), dim3(1024), 0, s, lg, ld, V, chunk, k, ctx->lstk_stats, ctx->lstk2_cand); \
  } while (0)
    if (nv <= 1) LSTK2(1); else if (nv == 2) LSTK2(2); else if (nv == 3) LSTK2(3); else LSTK2(4);
#undef LSTK2
    KCHK();
    hipLaunchKernelGGL(lstk2_merge_kernel, dim3(M), dim3(64), 0, s, ctx->lstk2_cand, C, k, out_idx, out_logp);
    KCHK();
    return 0;
  }
  // (rows of up to 65 536 logits: larger vocabularies take the chunked form above, which needs a ctx — the 20-values-per-thread
  //  instantiation that used to cover them without one spilled 408 B and was unreachable from every ctx-holding caller: removed in round 6)
  if (V % 8 == 0 && ld % 8 == 0 && V <= 1024 * 8 * 8 && V <= row_max_v) {
    if (V <= 1024 * 8 * 4) hipLaunchKernelGGL(lstk_row_kernel<4>, dim3(M), dim3(1024), 0, s, (const bf16_t*)logits, ld, V, k, out_idx, out_logp);
    else hipLaunchKernelGGL(lstk_row_kernel<8>, dim3(M), dim3(1024), 0, s, (const bf16_t*)logits, ld, V, k, out_idx, out_logp);
    KCHK();
    return 0;
  }
  if (!ctx) return fail("logsoftmax_topk: needs a ctx (scratch)");
  hipLaunchKernelGGL(lstk_stats_kernel, dim3(M, LSTK_CHUNKS), dim3(256), 0, s, (const bf16_t*)logits, ld, V, ctx->lstk_stats);
  KCHK();
  hipLaunchKernelGGL(lstk_select_kernel, dim3(M, LSTK_CHUNKS), dim3(256), 0, s, (const bf16_t*)logits, ld, V, k, ctx->lstk_stats,
                     ctx->lstk_cv, ctx->lstk_ci);
  KCHK();
  hipLaunchKernelGGL(lstk_merge_kernel, dim3(M), dim3(256), 0, s, k, ctx->lstk_cv, ctx->lstk_ci, out_idx, out_logp);
  KCHK();
  return 0;
}

// The same selection for every request of a cohort in ONE launch per pass (row form: one launch; chunked form: three): request t's
// rows are x[t]->dlogits, its results x[t]->top_idx / top_logp.  Falls back to the per-request launches for shapes only the generic
// three-pass form covers.
struct Cohort;
static int launch_lstopk_cohort(vispec_ctx* const* x, int n, hipStream_t s, int M, int V, int k) {
  if (k < 1 || k > TOPK_MAX) return fail("logsoftmax_topk: k must be in [1,16]");
  static const int chunk_min_v = getenv("VISPEC_LSTK_CHUNK_MIN_V") ? atoi(getenv("VISPEC_LSTK_CHUNK_MIN_V")) : 65536;
  static const int row_max_v = getenv("VISPEC_LSTK_ROW_MAX_V") ? atoi(getenv("VISPEC_LSTK_ROW_MAX_V")) : 1024 * 8 * 20;
  if (n > 1 && V % 8 == 0 && V > chunk_min_v && V <= 32768 * LSTK_CHUNKS) {
    int C = (V + 32767) / 32768;
    if (C < 8) C = 8;
    const int chunk = ((V / 8 + C - 1) / C) * 8, nv = (chunk / 8 + 1023) / 1024;
    auto p1 = [&](int t) { return make_pack((const bf16_t*)x[t]->dlogits, V, V, chunk, x[t]->lstk_stats); };
    auto p2 = [&](int t) { return make_pack((const bf16_t*)x[t]->dlogits, V, V, chunk, k, (const float*)x[t]->lstk_stats, x[t]->lstk2_cand); };
    auto p3 = [&](int t) { return make_pack((const unsigned long long*)x[t]->lstk2_cand, C, k, x[t]->top_idx, x[t]->top_logp); };
    decltype(p1(0)) a1[MAX_COHORT]; decltype(p2(0)) a2[MAX_COHORT]; decltype(p3(0)) a3[MAX_COHORT];
    for (int t = 0; t < n; ++t) { a1[t] = p1(t); a2[t] = p2(t); a3[t] = p3(t); }
#define LSTK2N(NV_)                                                                            \
  do {                                                                                         \
    launch_batch<lstk2_stats_fn<NV_>, 1024>(s, dim3(M, C), 0, a1, n);                          \
    launch_batch<lstk2_select_fn<NV_>, 1024>(s, dim3(M, C), 0, a2, n);                         \
  } while (0)
    if (nv <= 1) LSTK2N(1); else if (nv == 2) LSTK2N(2); else if (nv == 3) LSTK2N(3); else LSTK2N(4);
#undef LSTK2N
    KCHK();
    launch_batch<lstk2_merge_fn, 64>(s, dim3(M), 0, a3, n);
    KCHK();
    return 0;
  }
  if (n > 1 && V % 8 == 0 && V <= 1024 * 8 * 8 && V <= row_max_v && V <= chunk_min_v) {
    auto pr = [&](int t) { return make_pack((const bf16_t*)x[t]->dlogits, V, V, k, x[t]->top_idx, x[t]->top_logp); };
    decltype(pr(0)) a[MAX_COHORT];
    for (int t = 0; t < n; ++t) a[t] = pr(t);
    if (V <= 1024 * 8 * 4) launch_batch<lstk_row_fn<4>, 1024>(s, dim3(M), 0, a, n);
    else launch_batch<lstk_row_fn<8>, 1024>(s, dim3(M), 0, a, n);
    KCHK();
    return 0;
  }
  for (int t = 0; t < n; ++t)
    if (launch_lstopk(x[t], s, x[t]->dlogits, V, M, V, k, x[t]->top_idx, x[t]->top_logp)) return -1;
  return 0;
}

// Prefill-side GEMM (csrc/gemm_prefill.h): M rows against `n_tiles` 32-row blocks of a W32-packed weight starting at block n_tile0.
enum { PROF_GEMM_PREFILL = 7 };
static int launch_gemm_big(hipStream_t s, int epi, const BigA& a, int M, const void* P, int K, int n_tile0, int n_tiles, const BigEpi& e) {
  if (M < 1 || K % 64 || a.k_split % 64 || n_tiles < 1) return fail("gemm_prefill: K, k_split %% 64 == 0 and M, n_tiles >= 1 required");
  dim3 grid((n_tiles + 3) / 4, (M + 127) / 128), block(256);
  prof_begin(s, PROF_GEMM_PREFILL, 2.0 * M * (double)n_tiles * 32 * K);  // (flops, not bytes: this kernel is MFMA-bound)
  if (epi == BIG_PLAIN) PLAUNCH(gemm_w32_big_kernel<BIG_PLAIN>, grid, block, BIG_LDS_BYTES, s, a, M, (const bf16_t*)P, K, n_tile0, n_tiles, e);
  else if (epi == BIG_KV) PLAUNCH(gemm_w32_big_kernel<BIG_KV>, grid, block, BIG_LDS_BYTES, s, a, M, (const bf16_t*)P, K, n_tile0, n_tiles, e);
  else PLAUNCH(gemm_w32_big_kernel<BIG_ROPE_KV>, grid, block, BIG_LDS_BYTES, s, a, M, (const bf16_t*)P, K, n_tile0, n_tiles, e);
  KCHK();
  prof_end(s);
  return 0;
}
#define PREFILL_BIG_MIN_ROWS 64  // below this many rows a stage stays on the skinny kernel (one or two 32-row passes)

// ------------------------------------------------------------------------------------------------ unit-level C-ABI
extern "C" int vispec_pack_weight(vispec_ctx*, void* stream, const void* W, int N, int K, void* P) {
  return launch_pack((hipStream_t)stream, W, N, K, P);
}
extern "C" long long vispec_packed_elems(int N, int K) { return (long long)packed_elems(N, K); }
extern "C" int vispec_pack_weight_fp8(vispec_ctx*, void* stream, const void* Wq_rowmajor_u8, int N, int K, void* P) {
  if (K % 32) return fail("pack_fp8: K %% 32");
  hipLaunchKernelGGL(pack_w32_fp8_kernel, dim3(K / 32, (N + 31) / 32), dim3(64), 0, (hipStream_t)stream, (const unsigned char*)Wq_rowmajor_u8,
                     N, K, (unsigned char*)P);
  KCHK();
  return 0;
}
// vispec_gemm_skinny on an fp8 (e4m3) W32 image: Y = bf16(scale[n] * (X · Wq^T) + bias) (+ epilogue); scale fp32 [N] (2N for SwiGLU)
extern "C" int vispec_gemm_skinny_fp8(vispec_ctx* ctx, void* stream, const void* X, int ldx, const void* P8, const void* wscale,
                                      const void* bias, void* Y, int ldy, const void* R, int ldr, int M, int N, int K, int epilogue) {
  CTX_LIVE_OPT(ctx);
  if (epilogue < 0 || epilogue > 2 || !wscale) return fail("gemm_skinny_fp8: bad arguments");
  return launch_gemm(ctx, (hipStream_t)stream, X, ldx, P8, bias, Y, ldy, R, ldr, M, N, K, epilogue, wscale);
}
extern "C" int vispec_gemm_skinny(vispec_ctx* ctx, void* stream, const void* X, int ldx, const void* P, const void* bias, void* Y,
                                  int ldy, const void* R, int ldr, int M, int N, int K, int epilogue) {
  CTX_LIVE_OPT(ctx);
  if (epilogue < 0 || epilogue > 2) return fail("gemm_skinny: bad epilogue");
  return launch_gemm(ctx, (hipStream_t)stream, X, ldx, P, bias, Y, ldy, R, ldr, M, N, K, epilogue);
}
extern "C" int vispec_gemm_cohort(vispec_ctx* ctx, void* stream, const void* X, int ldx, const void* P, const void* wscale, const void* bias, void* Y,
                                  int ldy, const void* R, int ldr, int n_req, int m_tile, int N, int K, int epilogue) {
  CTX_LIVE_OPT(ctx);
  if (epilogue < 0 || epilogue > 2) return fail("gemm_cohort: bad epilogue");
  if (n_req < 2 || n_req > MAX_COHORT || m_tile == 0 || m_tile > 32 || m_tile < -8) return fail("gemm_cohort: 2..8 requests of 1..32 rows (slab mode: -8..-1)");
  if (m_tile < 0)  // slab mode: the requests' rows share one activation tile
    return launch_gemm(ctx, (hipStream_t)stream, X, ldx, P, bias, Y, ldy, R, ldr, 8 * (n_req - 1) - m_tile, N, K, epilogue, wscale, m_tile);
  return launch_gemm(ctx, (hipStream_t)stream, X, ldx, P, bias, Y, ldy, R, ldr, 32 * (n_req - 1) + m_tile, N, K, epilogue, wscale, m_tile);
}
// The W8A8 GEMM at unit level (tests): X is bf16 — it is quantised here exactly as target_forward does (quant_rows_e4m3_kernel into the ctx's
// scratch) — then multiplied by the e4m3 weight image on the fp8 MFMA.  n_req = 1: rows 0 .. M-1 (M <= 64); n_req = 2..4: the cohort layout of
// vispec_gemm_cohort (request t at rows 32 t .., m_tile live rows each; M ignored).  epilogue 0 none / 1 +R / 2 SwiGLU; norm_w != NULL adds the
// fused RMSNorm of the split-K reduce (the o_proj / down_proj form).
extern "C" int vispec_gemm_fp8a8(vispec_ctx* ctx, void* stream, const void* X, int ldx, const void* P8, const void* wscale, const void* bias, void* Y, int ldy,
                                 const void* R, int ldr, int n_req, int m_tile, int M, int N, int K, int epilogue, const void* norm_w, void* normed, float eps) {
  CTX_LIVE_OPT(ctx);
  if (!ctx || !wscale || epilogue < 0 || epilogue > 2 || K % 64 || ldx != K) return fail("gemm_fp8a8: needs a ctx, weight scales, K %% 64 == 0 and a dense X (ldx == K)");
  if (n_req < 1 || n_req > MAX_COHORT || (n_req > 1 && (m_tile < 1 || m_tile > 32)) || (n_req == 1 && (M < 1 || M > 64))) return fail("gemm_fp8a8: bad row layout");
  const size_t kmax = (size_t)std::max(std::max((int)ctx->c.hidden_size, (int)(ctx->c.num_heads * ctx->c.head_dim)), (int)ctx->c.intermediate_size);
  if ((size_t)K > kmax) return fail("gemm_fp8a8: K exceeds the ctx's quantisation scratch (max of hidden, heads x head_dim, intermediate size)");
  hipStream_t s = (hipStream_t)stream;
  const int rows = n_req == 1 ? M : 32 * (n_req - 1) + m_tile;
  const char* skip = getenv("VISPEC_A8_SKIP_QUANT");  // experiments (tools/fp8_k_sweep.py): time the GEMM without its quantisation pass
  if (!skip || skip[0] != '1') {
    hipLaunchKernelGGL(quant_rows_e4m3_kernel, dim3(rows), dim3(256), 0, s, (const bf16_t*)X, ldx, ctx->xq, K, ctx->sx, K);
    KCHK();
  }
  GemmOut o;
  o.wscale = (const float*)wscale; o.xscale = ctx->sx; o.m_tile = n_req == 1 ? 0 : m_tile;
  o.Y = Y; o.ldy = ldy; o.R = R; o.ldr = ldr; o.norm_w = norm_w; o.normed = normed; o.ldn = N; o.eps = eps;
  if (norm_w && (size_t)N <= kmax) { o.q8 = ctx->xq; o.sx8 = ctx->sx; }  // as target_forward: the normed rows leave the reduce quantised as well
  return launch_gemm_ex(ctx, s, ctx->xq, K / 2, P8, bias, rows, N, K, epilogue, o);
}
// Row-wise e4m3 quantisation of an activation matrix (the W8A8 arithmetic: sx[m] = max|x[m, :]| / 448, q = e4m3(x / sx)) into caller buffers:
// what the PyTorch prefill of an fp8a8 model feeds torch._scaled_mm (the library's fp8 x fp8 GEMM) with — the same kernel as the decode path's
extern "C" int vispec_quant_rows_e4m3(vispec_ctx*, void* stream, const void* X, int ldx, void* Q, int ldq, void* sx, int M, int K) {
  if (!X || !Q || !sx || M < 1 || K < 8 || K % 8 || ldx < K || ldq < K || ldx % 8 || ldq % 8) return fail("quant_rows_e4m3: K, ldx, ldq multiples of 8, ld >= K");
  hipLaunchKernelGGL(quant_rows_e4m3_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ldx, (unsigned char*)Q, ldq, (float*)sx, K);
  KCHK();
  return 0;
}
// test hook: the ctx's W8A8 scratch (e4m3 codes [rows][K] and per-row scales) as the last quantisation left it — copied on `stream`
extern "C" int vispec_a8_scratch_read(vispec_ctx* ctx, void* stream, void* codes_out, void* scales_out, int rows, int K) {
  CTX_LIVE(ctx);
  if (!ctx || !ctx->xq || rows < 1 || rows > ROWS || K < 1) return fail("a8_scratch_read: bad arguments");
  if (hipMemcpyAsync(codes_out, ctx->xq, (size_t)rows * K, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess ||
      hipMemcpyAsync(scales_out, ctx->sx, (size_t)rows * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
    return fail("a8_scratch_read: copy failed");
  return 0;
}
// skinny GEMM (+bias, +residual R) with the following RMSNorm fused: Y = bf16(R + bf16(X·W^T + b)), normed = norm_w * rms(Y)
extern "C" int vispec_gemm_skinny_norm(vispec_ctx* ctx, void* stream, const void* X, int ldx, const void* P, const void* bias, void* Y,
                                       int ldy, const void* R, int ldr, const void* norm_w, void* normed, int ldn, float eps, int M,
                                       int N, int K) {
  CTX_LIVE_OPT(ctx);
  GemmOut o;
  o.Y = Y; o.ldy = ldy; o.R = R; o.ldr = ldr; o.norm_w = norm_w; o.normed = normed; o.ldn = ldn; o.eps = eps;
  return launch_gemm_ex(ctx, (hipStream_t)stream, X, ldx, P, bias, M, N, K, R ? EPI_RESIDUAL : EPI_NONE, o);
}
// Tuning hook (tools/gemm_bench.py): packed-weight GEMM (no bias / epilogue) with explicit decomposition
//   variant = S*100 + NW_code*10 + UNROLL_code ; NW_code 0/1 = 4/8 waves ; UNROLL_code 0/1/2 = 4/8/16 ; S = split-K (>=1 -> partial path)
extern "C" int vispec_gemm_skinny_tune(vispec_ctx* ctx, int variant, void* stream, const void* X, int ldx, const void* P, void* Y, int ldy,
                                       int M, int N, int K) {
  CTX_LIVE_OPT(ctx);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t *x = (const bf16_t*)X, *w = (const bf16_t*)P;
  const int tiles = (N + 31) / 32;
  const bool no_reduce = variant >= 10000;  // time the GEMM kernel alone
  const int dbg = variant / 10000 >= 2 ? variant / 10000 - 1 : 0;  // 2xxxx: no activation loads, 3xxxx: no epilogue
  if (variant / 10000 == 7) {  // 7xxxx: the prefill kernel (128 x 128 workgroup tiles, activations shared through LDS), any M
    BigA a;
    a.src0 = x; a.ld0 = ldx; a.k_split = K;
    BigEpi e;
    e.Y = (bf16_t*)Y; e.ldy = ldy;
    return launch_gemm_big(s, BIG_PLAIN, a, M, P, K, 0, tiles, e);
  }
  variant %= 10000;
  const int S = variant / 100, nwc = (variant / 10) % 10, unc = variant % 10;
  if (S < 1 || S > 16 || !ctx || (size_t)S * 32 * N > ctx->gemm_part_elems) return fail("tune: bad split");
#define V(NWV, UN)                                                                                                              \
  hipLaunchKernelGGL((gemm_w32_kernel<1, EPI_PARTIAL, UN, NWV>), dim3(tiles, S), dim3(NWV * 64), (gemm_w32_lds_bytes<1, UN, NWV>()), s, x, \
                     ldx, w, 0, nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr)
  if (M > 128 && dbg == 8) {  // 9xxxx at M > 128: the cohort-8 kernel (eight requests, one accumulator chain per element), kernel alone
    if (M > ROWS || (size_t)S * C8_MPAD * N > ctx->gemm_part_elems) return fail("tune: c8 needs M <= 256 and a partial workspace of S*256*N");
    if (!c8_fast_ok(K, S, 0)) return fail("tune: c8 needs K %% 64 == 0 and a group per split");
    hipLaunchKernelGGL((gemm_w32_c8_kernel<EPI_PARTIAL, 0>), dim3((tiles + 7) / 8, S), dim3(512), C8_LDS_BYTES, s, x, ldx, w, nullptr, ctx->gemm_part, 0, nullptr,
                       0, 30, 8, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    KCHK();
    return 0;
  }
  if (M > 64 && dbg == 8) {  // 9xxxx: the wide-cohort kernel (16 waves = 4 row blocks x 4 K-quarters sharing staged activations), kernel alone
    if (M > 128 || (size_t)S * 128 * N > ctx->gemm_part_elems) return fail("tune: wide needs M <= 128 and a partial workspace of S*128*N");
    if (M > 96 && unc == 5) {  // 9xxx5: EIGHT row blocks per workgroup, K walked quarter by quarter (gemm_w32_wide8_kernel)
      if (!wide8_ok(K / 16, S, 4)) return fail("tune: wide8 needs a whole 64-k group in every K-quarter");
      hipLaunchKernelGGL((gemm_w32_wide8_kernel<EPI_PARTIAL, false, 4>), dim3((tiles + 7) / 8, S), dim3(512), wide8_lds_bytes<false>(), s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles);
    } else if (M > 96 && unc == 4)  // 9xxx4: three row blocks per workgroup
      hipLaunchKernelGGL((gemm_w32_wide_kernel<EPI_PARTIAL, false, 4, 0, 3>), dim3((tiles + 2) / 3, S), dim3(768), WIDE_LDS_BYTES, s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    else if (M > 96 && unc == 3)  // 9xxx3: two row blocks per workgroup (twice the workgroups, twice the X traffic per weight byte)
      hipLaunchKernelGGL((gemm_w32_wide_kernel<EPI_PARTIAL, false, 4, 0, 2>), dim3((tiles + 1) / 2, S), dim3(512), WIDE_LDS_BYTES, s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    else if (M > 96 && unc == 1)  // 9xxx1 / 9xxx2: the same without activation DMAs / without weight loads (wrong results; what each stream costs)
      hipLaunchKernelGGL((gemm_w32_wide_kernel<EPI_PARTIAL, false, 4, 1>), dim3((tiles + 3) / 4, S), dim3(1024), WIDE_LDS_BYTES, s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    else if (M > 96 && unc == 2)
      hipLaunchKernelGGL((gemm_w32_wide_kernel<EPI_PARTIAL, false, 4, 2>), dim3((tiles + 3) / 4, S), dim3(1024), WIDE_LDS_BYTES, s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    else if (M > 96)
      hipLaunchKernelGGL((gemm_w32_wide_kernel<EPI_PARTIAL, false, 4>), dim3((tiles + 3) / 4, S), dim3(1024), WIDE_LDS_BYTES, s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    else
      hipLaunchKernelGGL((gemm_w32_wide_kernel<EPI_PARTIAL, false, 3>), dim3((tiles + 3) / 4, S), dim3(1024), WIDE_LDS_BYTES, s, x, ldx, w, nullptr,
                         ctx->gemm_part, 0, nullptr, 0, 30, N, K, S, nullptr, RopeEpi{}, tiles, (const float*)nullptr);
    KCHK();
    return 0;
  }
  if (M > 64) {  // 8xxxx: FOUR activation tiles, one row block per workgroup (what a cohort of four would run) — measurement only
    if (dbg != 7 || M > 128 || (size_t)S * 128 * N > ctx->gemm_part_elems) return fail("tune: M > 64 needs variant 8xxxx, M <= 128 and a partial workspace of S*128*N");
    hipLaunchKernelGGL((gemm_w32_kernel<1, EPI_PARTIAL, 4, 4, 0, false, 4>), dim3(tiles, S), dim3(256), (gemm_w32_lds_bytes<1, 4, 4, 4>()), s, x, ldx, w, 0,
                       nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr);
    KCHK();
    return 0;
  }
  if (M > 32) {  // two activation tiles (the cohort / wide-tree instantiation): S x 4 waves x UNROLL 4 only
    if (M > 64 || (size_t)S * 64 * N > ctx->gemm_part_elems) return fail("tune: bad M / split for two tiles");
#define V2(DBG_)                                                                                                                   \
  hipLaunchKernelGGL((gemm_w32_kernel<1, EPI_PARTIAL, 4, 4, DBG_, false, 2>), dim3(tiles, S), dim3(256), (gemm_w32_lds_bytes<1, 4, 4, 2>()), s, x, \
                     ldx, w, 0, nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr)
    if (dbg == 5) {  // 6xxxx: the paired form with only HALF of the activations loaded (wrong results; what halving that traffic again would buy)
      hipLaunchKernelGGL((gemm_w32_kernel<2, EPI_PARTIAL, 4, 4, 3, false, 2>), dim3(tiles / 2, S), dim3(256), (gemm_w32_lds_bytes<2, 4, 4, 2>()), s, x,
                         ldx, w, tiles / 2, nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr);
    } else if (dbg == 4) {  // 5xxxx: two row blocks per workgroup (NT = 2) sharing every staged activation group: half the L2 -> CU activation traffic
      if (tiles & 1) return fail("tune: NT=2 needs an even tile count");
      hipLaunchKernelGGL((gemm_w32_kernel<2, EPI_PARTIAL, 4, 4, 0, false, 2>), dim3(tiles / 2, S), dim3(256), (gemm_w32_lds_bytes<2, 4, 4, 2>()), s, x,
                         ldx, w, tiles / 2, nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr);
    } else if (dbg == 1) V2(1); else if (dbg == 2) V2(2); else V2(0);
#undef V2
    KCHK();
    return 0;
  }
  if (dbg == 4) {  // 5xxxx at M <= 32: two row blocks per workgroup, one activation tile
    if (tiles & 1) return fail("tune: NT=2 needs an even tile count");
    hipLaunchKernelGGL((gemm_w32_kernel<2, EPI_PARTIAL, 4, 4, 0, false, 1>), dim3(tiles / 2, S), dim3(256), (gemm_w32_lds_bytes<2, 4, 4, 1>()), s, x,
                       ldx, w, tiles / 2, nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr);
    KCHK();
    return 0;
  }
  if (dbg == 1) {
    hipLaunchKernelGGL((gemm_w32_kernel<1, EPI_PARTIAL, 4, 4, 1>), dim3(tiles, S), dim3(256), (gemm_w32_lds_bytes<1, 4, 4>()), s, x, ldx, w, 0,
                       nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr);
    KCHK();
    return 0;
  }
  if (dbg == 2) {
    hipLaunchKernelGGL((gemm_w32_kernel<1, EPI_PARTIAL, 4, 4, 2>), dim3(tiles, S), dim3(256), (gemm_w32_lds_bytes<1, 4, 4>()), s, x, ldx, w, 0,
                       nullptr, ctx->gemm_part, 0, nullptr, 0, M, N, K, S, nullptr, RopeEpi{}, 0, (const float*)nullptr);
    KCHK();
    return 0;
  }
  switch (nwc * 10 + unc) {
    case 0: V(4, 4); break;
    case 1: V(4, 8); break;
    case 10: V(8, 4); break;
    case 11: V(8, 8); break;
    case 20: V(2, 4); break;
    case 21: V(2, 8); break;
    default: return fail("tune: unknown variant");
  }
#undef V
  KCHK();
  if (no_reduce) return 0;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(M), dim3(256), 0, s, ctx->gemm_part, S, 32, N, nullptr, nullptr, 0, (bf16_t*)Y, ldy, nullptr,
                     nullptr, 0, 0.f, 0);
  KCHK();
  return 0;
}

extern "C" int vispec_rmsnorm(vispec_ctx*, void* stream, const void* X, const void* w, void* Y, int M, int D, float eps) {
  return launch_rmsnorm((hipStream_t)stream, X, w, Y, M, D, eps);
}
extern "C" int vispec_add_rmsnorm(vispec_ctx*, void* stream, void* X, const void* R, const void* w, void* Y, int M, int D, float eps) {
  if (!X || !R || !w || !Y || M < 1 || D % 8) return fail("add_rmsnorm: bad arguments (D %% 8 == 0)");
  hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (bf16_t*)X, (const bf16_t*)R, (const bf16_t*)w, (bf16_t*)Y, D, eps);
  KCHK();
  return 0;
}
// Causal attention of a prompt's L rows over the K/V rows [0, L) its prefill has just written (csrc/kernels.h: prefill_attn_kernel).
extern "C" int vispec_prefill_attention(vispec_ctx*, void* stream, const void* q, int ldq, const void* k_cache, const void* v_cache, int s_max,
                                        int H, int H_kv, int L, void* out, int ldo, int eager_scores) {
  if (!q || !k_cache || !v_cache || !out) return fail("prefill_attention: null pointer");
  if (L < 1 || L > s_max || H < 1 || H_kv < 1 || H % H_kv || ldq % 8 || ldo % 4) return fail("prefill_attention: bad shape");
  {  // the dynamic-LDS limit is a per-DEVICE attribute and lanes call from several host threads: once per device, under a flag of its own
    static std::once_flag once[64];
    static bool ok[64];
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail("prefill_attention: device index out of range");
    std::call_once(once[dev], [dev]() {
      ok[dev] = hipFuncSetAttribute((const void*)prefill_attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES) == hipSuccess &&
                hipFuncSetAttribute((const void*)prefill_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES) == hipSuccess;
    });
    if (!ok[dev]) return fail("prefill_attention: cannot raise the dynamic LDS limit");
  }
  const int NB = (L + 127) / 128;
  const int paired = (NB / 2) * H >= 256 ? 1 : 0;  // a workgroup = one long + one short row block when that still fills the CUs
  const dim3 grid(paired ? (NB + 1) / 2 : NB, H), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (eager_scores)
    hipLaunchKernelGGL(prefill_attn_kernel<true>, grid, block, ATT2_LDS_BYTES, s, (const bf16_t*)q, ldq, (const bf16_t*)k_cache, (const bf16_t*)v_cache,
                       s_max, H, H_kv, L, (bf16_t*)out, ldo, paired);
  else
    hipLaunchKernelGGL(prefill_attn_kernel<false>, grid, block, ATT2_LDS_BYTES, s, (const bf16_t*)q, ldq, (const bf16_t*)k_cache, (const bf16_t*)v_cache,
                       s_max, H, H_kv, L, (bf16_t*)out, ldo, paired);
  KCHK();
  return 0;
}
extern "C" int vispec_silu_mul(vispec_ctx*, void* stream, const void* gate_up, int ld, void* out, int ldo, int M, int I) {
  if (M < 1 || I % 8 || ld % 8 || ldo % 8) return fail("silu_mul: I, ld, ldo must be multiples of 8");
  hipLaunchKernelGGL(silu_mul_kernel, dim3((I / 8 + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gate_up, ld, (bf16_t*)out,
                     ldo, I);
  KCHK();
  return 0;
}
// y[M, N] = bf16(acc[M, N] * scale[n] (+ bias[n])): the W8A16 epilogue applied to a library GEMM's fp32 output (fp8-weight prefill)
extern "C" int vispec_scale_bias_cast(vispec_ctx*, void* stream, const void* acc_f32, int ld, const void* scale_f32, const void* bias_bf16, void* out_bf16,
                                      int ldo, int M, int N) {
  if (!acc_f32 || !scale_f32 || !out_bf16) return fail("scale_bias_cast: null pointer");
  if (M < 1 || N < 8 || N % 8 || ld % 4 || ldo % 8) return fail("scale_bias_cast: N, ldo must be multiples of 8, ld of 4");
  hipLaunchKernelGGL(scale_bias_cast_kernel, dim3((N / 8 + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, (const float*)acc_f32, ld, (const float*)scale_f32,
                     (const bf16_t*)bias_bf16, (bf16_t*)out_bf16, ldo, N);
  KCHK();
  return 0;
}
extern "C" int vispec_rope_append(vispec_ctx*, void* stream, void* qkv, int M, int H, int H_kv, int hd, const void* cosT,
                                  const void* sinT, const int* pos_base_dev, const int* pos_off_dev, void* k_cache,
                                  void* v_cache, int s_max, const int* kv_base_dev) {
  if (hd != 128) return fail("rope_append: head_dim must be 128");
  if (M < 1 || (!kv_base_dev && M > s_max)) return fail("rope_append: the rows do not fit the cache (M > s_max)");
  PosSpec ps;
  ps.base = pos_base_dev;
  ps.off = pos_off_dev;
  ps.kv_base = kv_base_dev;
  return launch_rope((hipStream_t)stream, qkv, M, H, H_kv, cosT, sinT, ps, k_cache, v_cache, s_max, 1);
}
extern "C" int vispec_qkv_rope_fused(int n_qkv_rows) { return qkv_rope_fused(n_qkv_rows) ? 1 : 0; }
extern "C" int vispec_gemm_qkv_rope(vispec_ctx* ctx, void* stream, const void* X, int ldx, const void* W, const void* wscale,
                                    const void* bias, void* qkv, int M, int H, int H_kv, int hd, int K, const void* cosT,
                                    const void* sinT, const int* pos_base_dev, const int* pos_off_dev, void* k_cache, void* v_cache,
                                    int s_max, const int* kv_base_dev) {
  CTX_LIVE_OPT(ctx);
  if (hd != 128) return fail("gemm_qkv_rope: head_dim must be 128");
  PosSpec ps;
  ps.base = pos_base_dev;
  ps.off = pos_off_dev;
  ps.kv_base = kv_base_dev;
  return launch_qkv_rope1(ctx, (hipStream_t)stream, X, ldx, W, bias, wscale, qkv, M, H, H_kv, K, cosT, sinT, ps, k_cache, v_cache, s_max);
}
extern "C" int vispec_tree_attention(vispec_ctx* ctx, void* stream, const void* q, int ldq, const void* k_cache,
                                     const void* v_cache, int s_max, int H, int H_kv, int hd, int M, const int* prefix_dev,
                                     int tail, const uint64_t* mask_dev, void* out, int ldo, int eager_scores) {
  CTX_LIVE_OPT(ctx);
  CTX_LIVE(ctx);
  if (hd != 128) return fail("tree_attention: head_dim must be 128");
  return launch_attention(ctx, (hipStream_t)stream, q, ldq, k_cache, v_cache, s_max, H, H_kv, M, prefix_dev, tail,
                          (const unsigned long long*)mask_dev, out, ldo, eager_scores, s_max);
}
extern "C" int vispec_argmax_rows(vispec_ctx*, void* stream, const void* logits, int ld, int M, int V, int* out_idx) {
  if (V % 8 || ld % 8) return fail("argmax_rows: V and ld must be multiples of 8");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V, out_idx);
  KCHK();
  return 0;
}
extern "C" int vispec_logsoftmax_topk(vispec_ctx* ctx, void* stream, const void* logits, int ld, int M, int V, int k, int* out_idx,
                                      float* out_logp) {
  CTX_LIVE_OPT(ctx);
  return launch_lstopk(ctx, (hipStream_t)stream, logits, ld, M, V, k, out_idx, out_logp);
}

// ------------------------------------------------------------------------------------------------ hipGraph replay
// A round is ~390 dependent launches whose arguments never change (all variable quantities live in DevState): capture the
// sequence once per (ctx, key) with stream capture and replay it with ONE hipGraphLaunch.  This does not shorten the GPU work
// (the round is GPU-bound on one stream) but removes ~1.4 ms of host launch time per round, which is what limits several
// concurrent batch-1 lanes per GPU.  The legacy null stream cannot be captured: calls on it launch directly.
template <class F>
static int run_graphed(vispec_ctx* ctx, hipStream_t s, vispec_ctx::GraphSlot& slot, const vispec_ctx::GraphKey& key, F body) {
  if (!ctx->use_graphs || g_prof.on || s == nullptr) {
    ++ctx->direct_runs;
    return body();
  }
  int victim = 0;
  for (int i = 0; i < vispec_ctx::GraphSlot::CAP; ++i) {
    if (slot.exec[i] && slot.key[i] == key) {
      slot.stamp[i] = ++slot.clock;
      HIPCHK(hipGraphLaunch(slot.exec[i], s));
      ++ctx->graph_replays;
      return 0;
    }
    if (!slot.exec[i] ? slot.exec[victim] != nullptr : (slot.exec[victim] && slot.stamp[i] < slot.stamp[victim])) victim = i;  // an empty entry, else the oldest
  }
  if (slot.exec[victim]) {
    (void)hipGraphExecDestroy(slot.exec[victim]);
    slot.exec[victim] = nullptr;
    slot.key[victim] = vispec_ctx::GraphKey{};
  }
  if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    ++ctx->direct_runs;
    return body();  // stream not capturable: plain launches
  }
  const int rc = body();
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess || !graph) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  const hipError_t ei = hipGraphInstantiate(&slot.exec[victim], graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess) {
    slot.exec[victim] = nullptr;
    return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
  }
  slot.key[victim] = key;
  slot.stamp[victim] = ++slot.clock;
  ++ctx->graph_captures;
  HIPCHK(hipGraphLaunch(slot.exec[victim], s));
  return 0;
}
// sampling = the launch sequence reads temperature / seed / top_k (verify's sampling accept); the draft only switches the row order
// of its retrieve table on temperature > 0, the AR step uses none of them
static vispec_ctx::GraphKey graph_key_n(vispec_ctx* const* ctxs, int n, int forced_accept, bool sampling_args) {
  vispec_ctx::GraphKey k;
  k.n_req = n;
  k.forced_accept = forced_accept;
  k.total_token = ctxs[0]->c.total_token;
  k.wide_rb = ctxs[0]->wide_rb;
  for (int t = 0; t < n; ++t) {
    const vispec_ctx* ctx = ctxs[t];
    const bool sampling = ctx->temperature > 1e-5f;
    k.who[t] = ctx;
    k.u_over |= ctx->u_over_on ? 1 << t : 0;
    // the context hints reach the launches only as their maximum over the requests (the attention grids' split count): a stream of
    // requests of different lengths through the slots of a cohort replays one graph per maximum, not one per combination
    k.n_hint[0] = t == 0 ? ctx->n_hint : std::max(k.n_hint[0], ctx->n_hint);
    if (t > 0) k.n_hint[t] = 0;
    k.temperature[t] = sampling_args ? (sampling ? ctx->temperature : 0.f) : (sampling ? 1.f : 0.f);
    k.seed[t] = sampling_args && sampling ? ctx->seed : 0;
    k.sample_top_k[t] = sampling_args && sampling ? ctx->sample_top_k : 0;
  }
  return k;
}
static vispec_ctx::GraphKey graph_key(vispec_ctx* ctx, int forced_accept, bool sampling_args) {
  return graph_key_n(&ctx, 1, forced_accept, sampling_args);
}
// out[0..2] = {graph replays, graph captures, direct (un-graphed) runs} of the round functions since ctx creation
extern "C" int vispec_graph_stats(vispec_ctx* ctx, long long* out3) {
  CTX_LIVE(ctx);
  if (!ctx || !out3) return fail("null");
  out3[0] = ctx->graph_replays; out3[1] = ctx->graph_captures; out3[2] = ctx->direct_runs;
  return 0;
}
extern "C" int vispec_set_wide_row_blocks(vispec_ctx* ctx, int row_blocks) {
  CTX_LIVE(ctx);
  if (row_blocks != 0 && row_blocks != 8 && row_blocks != 84 && (row_blocks < 2 || row_blocks > 4))
    return fail("wide_row_blocks: 8, 4, 3, 2, 84 (= 8 for bf16 weights, 4 for fp8: several lanes) or 0 (= the smallest of 2..4 that still runs in one round of CUs: one lane)");
  ctx->wide_rb = row_blocks;  // (part of the graph key: cohort rounds captured with another value are not replayed)
  return 0;
}
// fp8 (e4m3) ACTIVATIONS for the target's q|k|v, gate|up and down GEMMs (o_proj stays W8A16) of the verify / AR forwards (BASELINE config 5, "CDNA4 fp8 MFMA"): each GEMM
// input is quantised per row (dynamic scale) and the product runs on v_mfma_scale_f32_32x32x64_f8f6f4.  Needs fp8 target weights; the lm_head
// and the draft keep bf16 activations (the PyTorch prefill of an "fp8a8" model quantises the same three GEMM inputs itself and runs the library's
// fp8 x fp8 GEMM: model/target.py).  Set it on every ctx of a cohort.  (Cached graphs are dropped.)
extern "C" int vispec_set_fp8_activations(vispec_ctx* ctx, int on) {
  CTX_LIVE(ctx);
  if (on && (ctx->c.hidden_size % 64 || ctx->c.intermediate_size % 64)) return fail("fp8 activations need hidden and intermediate sizes that are multiples of 64");
  ctx->a8 = on != 0;
  for (auto* g : {&ctx->g_verify, &ctx->g_draft, &ctx->g_ar, &ctx->g_cverify, &ctx->g_cdraft, &ctx->g_car}) g->clear();
  return 0;
}
extern "C" int vispec_set_graphs(vispec_ctx* ctx, int on) {
  CTX_LIVE(ctx);
  ctx->use_graphs = on != 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------ the path
__global__ void begin_request_kernel(DevState* st, int L, int max_new, int eos, int kv_cap, int draft_cap, int rope_rows, int round_rows) {
  if (threadIdx.x == 0) {
    DevState z{};
    z.n_ctx = L;
    z.n_prev = L;
    z.max_new_tokens = max_new;
    z.eos_token_id = eos;
    z.tree_T = 0;
    z.kv_cap = kv_cap;
    z.stop2 = -1;
    z.draft_cap = draft_cap;
    z.draft_rope_rows = rope_rows;
    z.draft_round_rows = round_rows;
    *st = z;
  }
}
__global__ void set_first_token_kernel(DevState* st, const int* tok, int draft_len, int real_len) {
  if (threadIdx.x == 0) {
    st->next_token = *tok;
    st->draft_len = draft_len;
    st->draft_real_len = real_len;
  }
}

extern "C" int vispec_begin_request(vispec_ctx* ctx, void* stream, const int* prompt_ids_host, int L, int max_new_tokens) {
  CTX_LIVE(ctx);
  hipStream_t s = (hipStream_t)stream;
  const vispec_config& c = ctx->c;
  if (L < 1 || L + c.total_token + 8 > c.max_pos) return fail("prompt does not fit the KV cache");
  if (prompt_ids_host) HIPCHK(hipMemcpyAsync(ctx->tokens, prompt_ids_host, sizeof(int) * L, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(begin_request_kernel, dim3(1), dim3(64), 0, s, ctx->st, L, max_new_tokens, c.eos_token_id, c.max_pos,
                     c.draft_max_pos, ctx->rope_rows, c.top_k * c.depth + c.depth + 2);
  KCHK();
  long hint = (long)L + max_new_tokens + 2 * (c.depth + 2) + c.total_token + 64;
  hint = (hint + 511) / 512 * 512;  // in steps of the attention kernels' 512 keys per workgroup: the split count is all a launch takes from it
  ctx->n_hint = (int)(hint < c.max_pos ? hint : c.max_pos);
  return 0;
}

// fc(cat(emb, img_fc(cat(h, g))))  for `rows` rows already gathered:  dx1 = [h | g], dx2[:, :D] = emb   (cnets_ours.py:918-922,982-988)
// ---- cohorts ----------------------------------------------------------------------------------------------------------------------
// A round is HBM-bound on the weights, so two requests that run their rounds in lockstep can share ONE weight pass: every GEMM of the
// round is launched once on 64 activation rows — tile t (rows 32t ..) belongs to request t (gemm_w32_kernel's m_tile mode) — while
// everything that is per request (tree, accept, KV caches, attention, round state) stays per request: each request keeps its own
// vispec_ctx, and the "member" ctx's activation buffers are simply the second 32-row tiles of the "leader's" (vispec_ctx_create_member).
// Every request keeps the reference's batch-1 semantics; its tokens are bit-identical to a run on its own (the 64-row GEMM computes
// each row with the same tiles, in the same order).  n == 1 is the ordinary single-request round.
// A/B switch (VISPEC_DRAFT_SLAB=0): the draft GEMMs of a cohort in the tile-per-request form of the target's
static const bool g_draft_slab = !(getenv("VISPEC_DRAFT_SLAB") && atoi(getenv("VISPEC_DRAFT_SLAB")) == 0);
struct Cohort {
  int n;
  vispec_ctx* c[MAX_COHORT];
  int stride = 32;  // target-side activation rows per request: 32, or 64 for trees of 33..64 nodes (two tiles per request, n <= 4; retarget_views)
  vispec_ctx* lead() const { return c[0]; }
  // TARGET-side GEMMs (rows = the tree size T).  stride 64: every request spans two full tiles — 2 n tiles of 32 "live" rows; rows T .. 63 of
  // a request are computed on whatever its spare rows hold and land in its own spare rows (the q|k|v epilogue, which writes the KV cache, takes
  // the exact row counts: RopeEpi::rows)
  // (rows <= 32 at stride 64 — the AR step's single row: request t's rows sit in tile 2t, the odd tiles ride along dead)
  int mt(int rows) const { return n >= 2 ? (stride == 64 && rows > 32 ? 32 : rows) : 0; }  // m_tile argument
  int M(int rows) const {                                                                   // M argument of a shared GEMM
    if (n < 2) return rows;
    if (stride == 64) return rows > 32 ? 64 * n : 32 * (2 * n - 2) + rows;
    return 32 * (n - 1) + rows;
  }
  // The draft's GEMMs see at most top_k / depth + 2 / 1 live rows per request: with <= 8 of them the requests share ONE activation tile
  // (slab mode: GemmOut::m_tile < 0) and the weight pass costs what a single request's costs instead of the 128-row wide form.
  bool slab(int rows) const { return n >= 2 && rows <= 8 && g_draft_slab; }
  // (the draft-side views stay at 32 rows per request whatever the tree size)
  int dmt(int rows) const { return slab(rows) ? -rows : (n >= 2 ? rows : 0); }
  int dM(int rows) const { return slab(rows) ? 8 * (n - 1) + rows : (n >= 2 ? 32 * (n - 1) + rows : rows); }
};
static Cohort solo_cohort(vispec_ctx* ctx) { Cohort co{}; co.n = 1; co.c[0] = ctx; return co; }

// bcast_g: (re)write the right half of dx1 with the current global image feature g.  g changes only inside the draft prefill (one new
// g per image run); vispec_draft_prefill leaves dx1[:, D:2D] = final g for all rows, so the decode rounds never touch it.
static int draft_fuse(const Cohort& co, hipStream_t s, int rows, void* out, int ld_out, bool bcast_g) {
  vispec_ctx* ctx = co.lead();
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size;
  if (bcast_g) {
    if (co.n > 1) {
      auto pk = [&](int t) { return make_pack((const bf16_t*)co.c[t]->dg, co.c[t]->dx1 + D, 2 * D, D); };
      decltype(pk(0)) a[MAX_COHORT];
      for (int t = 0; t < co.n; ++t) a[t] = pk(t);
      launch_batch<bcast_row_fn, 256>(s, dim3(rows), 0, a, co.n);
      KCHK();
    } else if (launch_bcast(s, ctx->dg, ctx->dx1 + D, 2 * D, rows, D)) {
      return -1;
    }
  }
  if (launch_gemm(ctx, s, ctx->dx1, 2 * D, ctx->dw.imgfc_w, ctx->dw.imgfc_b, ctx->dx2 + D, 2 * D, nullptr, 0, co.dM(rows), D, 2 * D, EPI_NONE, nullptr,
                  co.dmt(rows)))
    return -1;
  return launch_gemm(ctx, s, ctx->dx2, 2 * D, ctx->dw.fc_w, ctx->dw.fc_b, out, ld_out, nullptr, 0, co.dM(rows), D, 2 * D, EPI_NONE, nullptr, co.dmt(rows));
}

// o_proj + residual -> post_attention_layernorm -> SwiGLU MLP -> + residual for `rows` rows per request (cnets_ours.py:575-600):
// attention output in dattn, residual rows `resid` (ld D), result rows in `out` (ld D) — pointers of the leader's 64-row buffers.
static int draft_layer_tail(const Cohort& co, hipStream_t s, int rows, const bf16_t* resid, bf16_t* out) {
  vispec_ctx* ctx = co.lead();
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size, Id = c.draft_intermediate;
  {
    GemmOut o;  // o_proj + residual, post_attention_layernorm fused into the split-K reduce
    o.Y = ctx->dh; o.ldy = D; o.R = resid; o.ldr = D; o.norm_w = ctx->dw.ln2; o.normed = ctx->dn; o.ldn = D; o.eps = c.draft_rms_eps;
    o.m_tile = co.dmt(rows);
    if (launch_gemm_ex(ctx, s, ctx->dattn, D, ctx->dw.wo, nullptr, co.dM(rows), D, D, EPI_RESIDUAL, o)) return -1;
  }
  if (launch_gemm(ctx, s, ctx->dn, D, ctx->dw.wgu, nullptr, ctx->dact, Id, nullptr, 0, co.dM(rows), Id, D, EPI_SWIGLU, nullptr, co.dmt(rows))) return -1;
  return launch_gemm(ctx, s, ctx->dact, Id, ctx->dw.wdown, nullptr, out, D, ctx->dh, D, co.dM(rows), D, Id, EPI_RESIDUAL, nullptr, co.dmt(rows));
}

// the draft's single decoder layer on `rows` rows per request of dx (cnets_ours.py:545-600); result in dout.
// level < 0: the catch-up forward (positions continue from the real length, causal mask); level >= 0: tree level (same position for
// all rows, level mask).
static int draft_layer(const Cohort& co, hipStream_t s, int rows, int level) {
  vispec_ctx* ctx = co.lead();
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size, Hd = c.draft_heads, k = c.top_k;
  QkvReq rq[MAX_COHORT];
  for (int t = 0; t < co.n; ++t) {
    vispec_ctx* x = co.c[t];
    PosSpec& ps = rq[t].ps;
    if (level < 0) {  // positions continue from the REAL length, KV rows from the compressed length (cnets_ours.py:845-868)
      ps.base = &x->st->draft_real_len;
      ps.kv_base = &x->st->draft_len;
    } else {  // position_ids = len_posi + i for all k rows (cnets_ours.py:1128,1137); KV rows appended after the stable KV
      ps.base = &x->st->n_ctx;
      ps.add = level;
      ps.row = 0;
      ps.kv_base = &x->st->draft_len;
      ps.kv_add = level * k;
    }
    rq[t].kc = x->draft_kv;
    rq[t].vc = x->draft_kv + (size_t)Hd * c.draft_max_pos * 128;
  }
  if (launch_qkv_rope(ctx, s, ctx->dx, D, ctx->dw.wqkv, ctx->dw.bqkv, nullptr, ctx->dqkv, rows, Hd, Hd, D, ctx->dw.rope_cos, ctx->dw.rope_sin, rq,
                      co.n, c.draft_max_pos, co.slab(rows)))
    return -1;
  {
    AttnCall calls[MAX_COHORT];
    int max_keys = 1;
    for (int t = 0; t < co.n; ++t) {
      vispec_ctx* x = co.c[t];
      calls[t] = AttnCall{x, x->dqkv, rq[t].kc, rq[t].vc, &x->st->draft_len, level < 0 ? x->causal_mask : x->tb.lvl_mask, x->dattn};
      max_keys = std::max(max_keys, x->n_hint < c.draft_max_pos ? x->n_hint : c.draft_max_pos);
    }
    const int tail = level < 0 ? rows : k * (level + 1);
    if (launch_attention_n(s, calls, co.n, 3 * D, c.draft_max_pos, Hd, Hd, rows, tail, D, 0, max_keys)) return -1;
  }
  return draft_layer_tail(co, s, rows, ctx->dx, ctx->dout);
}

// head(last) -> log-softmax -> top-k, then `depth` tree levels and the final re-rank (cnets_ours.py:1109-1238).
// Expects dlast = last hidden row, draft_len/real_len already advanced.
static int draft_grow_tree(const Cohort& co, hipStream_t s) {
  vispec_ctx* ctx = co.lead();
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size, V = c.vocab_size, k = c.top_k;
  if (launch_gemm(ctx, s, ctx->dlast, D, ctx->tm.lm_head, nullptr, ctx->dlogits, V, nullptr, 0, co.dM(1), V, D, EPI_NONE, ctx->tm.lm_head_scale, co.dmt(1)))
    return -1;
  // (a cohort's per-request kernels go out as ONE launch each: launch_batch)
  if (launch_lstopk_cohort(co.c, co.n, s, 1, V, k)) return -1;
  // the tree kernels stage the next level's inputs themselves: dx1[:, :D] = input_hidden, dx2[:, :D] = embed(input_ids)
  if (co.n > 1) {
    auto pk = [&](int t) {
      vispec_ctx* x = co.c[t];
      return make_pack(x->tb, (const int*)x->top_idx, (const float*)x->top_logp, k, (const bf16_t*)x->dlast, (const bf16_t*)ctx->dw.embed, x->dx1, x->dx2, D);
    };
    decltype(pk(0)) a[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) a[t] = pk(t);
    launch_batch<tree_init_fn, 1024>(s, dim3(1), 0, a, co.n);
  } else {
    hipLaunchKernelGGL(tree_init_kernel, dim3(1), dim3(1024), 0, s, ctx->tb, ctx->top_idx, ctx->top_logp, k, ctx->dlast, (const bf16_t*)ctx->dw.embed,
                       ctx->dx1, ctx->dx2, D);
  }
  KCHK();
  for (int lvl = 0; lvl < c.depth; ++lvl) {
    if (draft_fuse(co, s, k, ctx->dx, D, false)) return -1;
    if (draft_layer(co, s, k, lvl)) return -1;
    if (launch_gemm(ctx, s, ctx->dout, D, ctx->tm.lm_head, nullptr, ctx->dlogits, V, nullptr, 0, co.dM(k), V, D, EPI_NONE, ctx->tm.lm_head_scale, co.dmt(k)))
      return -1;
    if (launch_lstopk_cohort(co.c, co.n, s, k, V, k)) return -1;
    if (co.n > 1) {
      auto pk = [&](int t) {
        vispec_ctx* x = co.c[t];
        return make_pack(x->tb, lvl, k, (const int*)x->top_idx, (const float*)x->top_logp, (const bf16_t*)x->dout, (const bf16_t*)ctx->dw.embed, x->dx1, x->dx2, D);
      };
      decltype(pk(0)) a[MAX_COHORT];
      for (int t = 0; t < co.n; ++t) a[t] = pk(t);
      launch_batch<tree_level_fn, 1024>(s, dim3(1), 0, a, co.n);
    } else {
      hipLaunchKernelGGL(tree_level_kernel, dim3(1), dim3(1024), 0, s, ctx->tb, lvl, k, ctx->top_idx, ctx->top_logp, ctx->dout, (const bf16_t*)ctx->dw.embed,
                         ctx->dx1, ctx->dx2, D);
    }
    KCHK();
  }
  // sampling: retrieve rows sorted (cnets_ours.py:1215-1224)
  if (co.n > 1) {
    auto pk = [&](int t) { vispec_ctx* x = co.c[t]; return make_pack(x->tb, x->st, k, c.depth, c.total_token - 1, x->temperature > 1e-5f ? 1 : 0); };
    decltype(pk(0)) a[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) a[t] = pk(t);
    launch_batch<tree_finalize_fn, 256>(s, dim3(1), 0, a, co.n);
  } else {
    hipLaunchKernelGGL(tree_finalize_kernel, dim3(1), dim3(256), 0, s, ctx->tb, ctx->st, k, c.depth, c.total_token - 1, ctx->temperature > 1e-5f ? 1 : 0);
  }
  KCHK();
  return 0;
}

static int draft_round_body(const Cohort& co, hipStream_t s) {
  vispec_ctx* ctx = co.lead();
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size, MC = c.depth + 2;  // a+1 <= depth+2 catch-up rows; rows beyond a are scratch
  // catch-up forward on the accepted hidden states (cnets_ours.py:1090-1097); its inputs (dx1[:, :D] = accepted hidden rows,
  // dx2[:, :D] = embeddings of the ids they pair with) were staged by the accept step (post_accept_kernel)
  if (draft_fuse(co, s, MC, ctx->dx, D, false)) return -1;
  if (draft_layer(co, s, MC, -1)) return -1;
  if (co.n > 1) {  // + dlast = out_hidden[:, -1]
    auto pk = [&](int t) { vispec_ctx* x = co.c[t]; return make_pack(x->st, (const bf16_t*)x->dout, x->dlast, D); };
    decltype(pk(0)) a[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) a[t] = pk(t);
    launch_batch<draft_advance_fn, 256>(s, dim3(1), 0, a, co.n);
  } else {
    hipLaunchKernelGGL(draft_advance_kernel, dim3(1), dim3(256), 0, s, ctx->st, ctx->dout, ctx->dlast, D);
  }
  KCHK();
  return draft_grow_tree(co, s);
}
extern "C" int vispec_draft_round(vispec_ctx* ctx, void* stream) {
  CTX_LIVE(ctx);
  hipStream_t s = (hipStream_t)stream;
  const Cohort co = solo_cohort(ctx);
  return run_graphed(ctx, s, ctx->g_draft, graph_key(ctx, 0, false), [&]() { return draft_round_body(co, s); });
}

extern "C" int vispec_draft_prefill(vispec_ctx* ctx, void* stream, const void* hidden, const void* embeds,
                                    const uint8_t* image_mask_host, int L, const int* first_token_dev) {
  CTX_LIVE(ctx);
  hipStream_t s = (hipStream_t)stream;
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size, q = c.num_q, Hd = c.draft_heads;
  const Cohort solo = solo_cohort(ctx);  // the prefill of a request always runs on its own (cohorts form in the decode rounds)
  // the prompt may be as long as the target cache allows (the reference's draft handles it, cnets_ours.py:879-975); what must fit the
  // draft's own cache is the COMPRESSED sequence (checked below), and every row is rotated at its real position < rope_rows
  if (L < 1 || L > ctx->scr_rows) return fail("draft_prefill: prompt does not fit the prefill scratch (max(max_pos, draft_max_pos) rows)");
  if (L + c.depth + 2 > ctx->rope_rows) return fail("draft_prefill: prompt does not fit the draft rotary tables (draft_rope_rows)");
  // inputs_embeds shifted by one, last row = embed(sampled token)  (cnets_ours.py:1081-1082)
  if (L > 1)
    HIPCHK(hipMemcpyAsync(ctx->emb_shift, (const bf16_t*)embeds + D, (size_t)(L - 1) * D * sizeof(bf16_t),
                          hipMemcpyDeviceToDevice, s));
  if (launch_gather(s, ctx->dw.embed, D, first_token_dev, 0, nullptr, ctx->emb_shift + (size_t)(L - 1) * D, D, 1, D)) return -1;
  HIPCHK(hipMemsetAsync(ctx->dg, 0, sizeof(bf16_t) * D, s));  // last_img_hidden = 0   (:914,978)
  // ---- host-side segmentation of the shifted mask (cnets_ours.py:880-884, 915-950) ----
  // Everything index-like is laid out in one pinned host block and uploaded once:
  //   [0,L) source row of every compressed text row | [L,2L) image rows, run after run | [2L,3L) position ids
  struct Op { int is_adapt, c_row, n, off; };
  std::vector<Op> plan;
  int* h_src = ctx->h_pin;
  int* h_img = ctx->h_pin + ctx->scr_rows;
  int* h_pos = ctx->h_pin + 2 * ctx->scr_rows;
  int c_row = 0, img_off = 0;
  {
    int start = 0;
    auto text_rows = [&](int lo, int hi_excl, const uint8_t* m1) {
      const int c0 = c_row;
      for (int r = lo; r < hi_excl; ++r)
        if (!m1 || !m1[r]) { h_src[c_row] = r; h_pos[c_row] = r; ++c_row; }
      if (c_row > c0) plan.push_back({0, c0, c_row - c0, 0});
    };
    if (image_mask_host && L > 1) {
      const uint8_t* m1 = image_mask_host + 1;  // length L-1
      for (int e = 0; e < L - 1; ++e) {
        const bool is_end = m1[e] && (e == L - 2 || !m1[e + 1]);
        if (!is_end) continue;
        text_rows(start, e + 1, m1);  // text rows before the run use the previous g   (:918-922)
        const int i0 = img_off;
        for (int r = start; r <= e; ++r)
          if (m1[r]) h_img[img_off++] = r;
        plan.push_back({1, c_row, img_off - i0, i0});
        for (int t = 0; t < q - 1; ++t) { h_src[c_row] = 0; h_pos[c_row] = (e + 1) - (q - 1) + t; ++c_row; }  // :932-937
        start = e + 1;
      }
    }
    text_rows(start, L, nullptr);  // :944-950 (the trailing segment is all text by construction)
  }
  const int Lc = c_row;
  if (Lc < 1) return fail("draft_prefill: bad compressed length");
  // the first round appends a catch-up (<= depth+2 rows) and top_k rows per tree level behind the compressed prompt
  if (Lc + c.top_k * c.depth + c.depth + 2 > c.draft_max_pos)
    return fail("draft_prefill: the compressed prompt does not fit the draft KV cache (draft_max_pos)");
  HIPCHK(hipMemcpyAsync(ctx->idx_tmp, h_src, sizeof(int) * Lc, hipMemcpyHostToDevice, s));
  if (img_off) HIPCHK(hipMemcpyAsync(ctx->idx_img, h_img, sizeof(int) * img_off, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(ctx->pos_c, h_pos, sizeof(int) * Lc, hipMemcpyHostToDevice, s));
  bf16_t* akc = ctx->ad_kv;
  const int ad_cap = ctx->scr_rows;  // row capacity of the adaptor's scratch K/V (one image run can be as long as the prompt)
  bf16_t* avc = ctx->ad_kv + (size_t)Hd * ad_cap * 128;
  for (const Op& op : plan) {
    if (!op.is_adapt && op.n >= PREFILL_BIG_MIN_ROWS) {
      // fc(cat(emb, img_fc(cat(h, g)))) over the segment's rows in two prefill GEMMs: the row gathers (hidden / shifted embeddings by
      // source row) and both concatenations happen while the operands are staged
      BigA a1;
      a1.src0 = (const bf16_t*)hidden; a1.idx0 = ctx->idx_tmp + op.c_row; a1.ld0 = D;
      a1.src1 = ctx->dg; a1.bcast1 = 1; a1.ld1 = D; a1.k_split = D;
      BigEpi e1;
      e1.bias = (const bf16_t*)ctx->dw.imgfc_b; e1.Y = ctx->pf_t1; e1.ldy = D;
      if (launch_gemm_big(s, BIG_PLAIN, a1, op.n, ctx->dw.imgfc_w, 2 * D, 0, D / 32, e1)) return -1;
      BigA a2;
      a2.src0 = ctx->emb_shift; a2.idx0 = ctx->idx_tmp + op.c_row; a2.ld0 = D;
      a2.src1 = ctx->pf_t1; a2.ld1 = D; a2.k_split = D;
      BigEpi e2;
      e2.bias = (const bf16_t*)ctx->dw.fc_b; e2.Y = ctx->xc + (size_t)op.c_row * D; e2.ldy = D;
      if (launch_gemm_big(s, BIG_PLAIN, a2, op.n, ctx->dw.fc_w, 2 * D, 0, D / 32, e2)) return -1;
      continue;
    }
    if (!op.is_adapt) {
      for (int o = 0; o < op.n; o += CHUNK) {
        const int rows = std::min(CHUNK, op.n - o), r0 = op.c_row + o;
        if (launch_gather(s, hidden, D, ctx->idx_tmp, r0, nullptr, ctx->dx1, 2 * D, rows, D)) return -1;
        if (launch_gather(s, ctx->emb_shift, D, ctx->idx_tmp, r0, nullptr, ctx->dx2, 2 * D, rows, D)) return -1;
        if (draft_fuse(solo, s, rows, ctx->xc + (size_t)r0 * D, D, true)) return -1;
      }
      continue;
    }
    // ImgAdaptor (cnets_ours.py:630-661): K/V projection of the image rows into a [2][H][cap][hd] scratch cache ...
    const int N = op.n;
    if (N >= PREFILL_BIG_MIN_ROWS) {  // K|V projection of all N image rows in one prefill GEMM, image-row gather fused into its operand load
      BigA a;
      a.src0 = ctx->emb_shift; a.idx0 = ctx->idx_img + op.off; a.ld0 = D; a.k_split = D;
      BigEpi e;
      e.bias = (const bf16_t*)ctx->dw.ad_bkv; e.kc = akc; e.vc = avc; e.cap = ad_cap; e.H = Hd; e.D = D;
      if (launch_gemm_big(s, BIG_KV, a, N, ctx->dw.ad_wkv, D, 0, 2 * D / 32, e)) return -1;
    }
    for (int o = 0; o < (N >= PREFILL_BIG_MIN_ROWS ? 0 : N); o += CHUNK) {
      const int rows = std::min(CHUNK, N - o);
      if (launch_gather(s, ctx->emb_shift, D, ctx->idx_img, op.off + o, nullptr, ctx->dx, D, rows, D)) return -1;
      if (launch_gemm(ctx, s, ctx->dx, D, ctx->dw.ad_wkv, ctx->dw.ad_bkv, ctx->ad_tmp, 2 * D, nullptr, 0, rows, 2 * D, D, EPI_NONE))
        return -1;
      PosSpec ps;
      ps.kv_add = o;
      if (launch_rope(s, ctx->ad_tmp, rows, 0, Hd, nullptr, nullptr, ps, akc, avc, ad_cap, 0)) return -1;
    }
    // ... then num_q learned queries attend over all N rows (non-causal), o_proj
    hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, s, (const int*)nullptr, N, ctx->scratch_int);
    KCHK();
    if (launch_attention(ctx, s, ctx->dw.ad_q, D, akc, avc, ad_cap, Hd, Hd, q, ctx->scratch_int, 0, nullptr, ctx->dattn, D,
                         0, N))
      return -1;
    if (launch_gemm(ctx, s, ctx->dattn, D, ctx->dw.ad_wo, nullptr, ctx->ad_out, D, nullptr, 0, q, D, D, EPI_NONE)) return -1;
    // first q-1 outputs are the compressed tokens, the last is the new global feature g   (:928-930)
    if (q > 1)
      HIPCHK(hipMemcpyAsync(ctx->xc + (size_t)op.c_row * D, ctx->ad_out, (size_t)(q - 1) * D * sizeof(bf16_t),
                            hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->dg, ctx->ad_out + (size_t)(q - 1) * D, (size_t)D * sizeof(bf16_t), hipMemcpyDeviceToDevice, s));
  }
  // ---- decoder layer over the compressed sequence: K,V for every row, the rest only for the last row (:1109) ----
  bf16_t* kc = ctx->draft_kv;
  bf16_t* vc = ctx->draft_kv + (size_t)Hd * c.draft_max_pos * 128;
  int last_chunk_rows = 0;
  int o_first = 0;
  if (Lc >= PREFILL_BIG_MIN_ROWS && qkv_rope_fused(3 * D)) {
    // k|v (+ rotary at each row's ORIGINAL position, + append) of every compressed row in one prefill GEMM over the k and v row blocks of
    // the rope-ordered q|k|v weight; q is only needed for the last row (:1109) — the last 32-row pass below provides it
    BigA a;
    a.src0 = ctx->xc; a.ld0 = D; a.k_split = D;
    BigEpi e;
    e.bias = (const bf16_t*)ctx->dw.bqkv; e.kc = kc; e.vc = vc; e.cap = c.draft_max_pos; e.H = Hd; e.D = D;
    e.cosT = (const bf16_t*)ctx->dw.rope_cos; e.sinT = (const bf16_t*)ctx->dw.rope_sin; e.pos = ctx->pos_c;
    if (launch_gemm_big(s, BIG_ROPE_KV, a, Lc, ctx->dw.wqkv, D, D / 32, 2 * D / 32, e)) return -1;
    o_first = ((Lc - 1) / CHUNK) * CHUNK;  // only the last chunk goes through the skinny q|k|v (it rewrites the same k, v rows)
  }
  for (int o = o_first; o < Lc; o += CHUNK) {
    const int rows = std::min(CHUNK, Lc - o);
    PosSpec ps;
    ps.off = ctx->pos_c + o;
    ps.kv_add = o;
    if (launch_qkv_rope1(ctx, s, ctx->xc + (size_t)o * D, D, ctx->dw.wqkv, ctx->dw.bqkv, nullptr, ctx->dqkv, rows, Hd, Hd, D,
                         ctx->dw.rope_cos, ctx->dw.rope_sin, ps, kc, vc, c.draft_max_pos))
      return -1;
    last_chunk_rows = rows;
  }
  const bf16_t* qlast = ctx->dqkv + (size_t)(last_chunk_rows - 1) * 3 * D;
  const bf16_t* xlast = ctx->xc + (size_t)(Lc - 1) * D;
  hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, s, (const int*)nullptr, Lc, ctx->scratch_int);
  KCHK();
  if (launch_attention(ctx, s, qlast, 3 * D, kc, vc, c.draft_max_pos, Hd, Hd, 1, ctx->scratch_int, 0, nullptr, ctx->dattn, D, 0, Lc))
    return -1;
  if (draft_layer_tail(solo, s, 1, xlast, ctx->dlast)) return -1;
  hipLaunchKernelGGL(set_first_token_kernel, dim3(1), dim3(64), 0, s, ctx->st, first_token_dev, Lc, L);
  KCHK();
  // the request's final g, for every later draft_fuse (they use <= max(depth + 2, top_k) <= 16 rows; a cohort member's dx1 is the
  // second 32-row tile of its leader's: never write past 32 rows of a ctx's own view)
  if (launch_bcast(s, ctx->dg, ctx->dx1 + D, 2 * D, 32, D)) return -1;
  return draft_grow_tree(solo, s);
}

static int target_forward(const Cohort& co, hipStream_t s, int T) {
  vispec_ctx* ctx = co.lead();
  const vispec_config& c = ctx->c;
  const int D = c.hidden_size, H = c.num_heads, Hk = c.num_kv_heads, V = c.vocab_size, I = c.intermediate_size;
  const int QKV = (H + 2 * Hk) * 128;
  const size_t slab = (size_t)Hk * c.max_pos * 128;
  QkvReq rq[MAX_COHORT];
  for (int t = 0; t < co.n; ++t) {
    vispec_ctx* x = co.c[t];
    if (!x->target_kv) return fail("target KV not set");
    // embed the tree tokens (modeling_llama_kv.py:985) + the first layer's input_layernorm (a cohort's: one launch, below)
    if (co.n == 1) {
      hipLaunchKernelGGL(embed_rmsnorm_kernel, dim3(T), dim3(256), 0, s, (const bf16_t*)ctx->tm.embed, x->tb.tree_tokens, x->xa,
                         (const bf16_t*)ctx->layers[0].ln1, x->xn, D, c.rms_eps);
      KCHK();
    }
    PosSpec& ps = rq[t].ps;  // position_ids = tree_position_ids + n (utils.py:397) ; KV rows [n, n+T)  (KVCache.cat)
    ps.base = &x->st->n_ctx;
    ps.base2 = &x->st->rope_delta;
    ps.off = x->tb.tree_pos;
    ps.kv_base = &x->st->n_ctx;
  }
  if (co.n > 1) {
    auto pk = [&](int t) {
      vispec_ctx* x = co.c[t];
      return make_pack((const bf16_t*)ctx->tm.embed, (const int*)x->tb.tree_tokens, x->xa, (const bf16_t*)ctx->layers[0].ln1, x->xn, D, c.rms_eps);
    };
    decltype(pk(0)) a[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) a[t] = pk(t);
    launch_batch<embed_rmsnorm_fn, 256>(s, dim3(T), 0, a, co.n);
    KCHK();
  }
  for (int l = 0; l < c.num_layers; ++l) {
    const vispec_layer_weights& w = ctx->layers[l];
    for (int t = 0; t < co.n; ++t) {
      rq[t].kc = co.c[t]->target_kv + (size_t)(2 * l) * slab;
      rq[t].vc = co.c[t]->target_kv + (size_t)(2 * l + 1) * slab;
    }
    // fp8 activations (vispec_set_fp8_activations, fp8 weights only): the inputs of q|k|v, gate|up and down_proj are quantised row by row to
    // e4m3 (by the reduce + norm that produces them; quant_rows_e4m3_kernel for the first layer's input and the SwiGLU output) and multiplied on
    // the fp8 MFMA; X then points at the codes (pitch in 2-byte units) and `xscale` at the row scales
    const bool a8 = ctx->a8 && w.sqkv != nullptr;
    const int MR = co.M(T);  // activation rows of the launch (a cohort: 32 (n - 1) + T, dead rows included)
    auto quant = [&](const bf16_t* X, int K_) {
      hipLaunchKernelGGL(quant_rows_e4m3_kernel, dim3(MR), dim3(256), 0, s, X, K_, ctx->xq, K_, ctx->sx, K_);
      return hipGetLastError() == hipSuccess ? 0 : fail("quant_rows launch failed");
    };
    if (a8) {
      if (l == 0 && quant(ctx->xn, D)) return -1;  // (later layers: the codes come out of the previous layer's down_proj reduce + norm)
      if (launch_qkv_rope(ctx, s, ctx->xq, D / 2, w.wqkv, w.bqkv, w.sqkv, ctx->qkv, T, H, Hk, D, ctx->tm.rope_cos, ctx->tm.rope_sin, rq, co.n, c.max_pos, false,
                          ctx->sx, co.stride == 64))
        return -1;
    } else if (launch_qkv_rope(ctx, s, ctx->xn, D, w.wqkv, w.bqkv, w.sqkv, ctx->qkv, T, H, Hk, D, ctx->tm.rope_cos, ctx->tm.rope_sin, rq, co.n, c.max_pos, false,
                               nullptr, co.stride == 64))
      return -1;
    {
      AttnCall calls[MAX_COHORT];
      int max_keys = 1;
      for (int t = 0; t < co.n; ++t) {
        vispec_ctx* x = co.c[t];
        calls[t] = AttnCall{x, x->qkv, rq[t].kc, rq[t].vc, &x->st->n_ctx, x->tb.tree_mask, x->attn_o};
        max_keys = std::max(max_keys, x->n_hint);
      }
      if (launch_attention_n(s, calls, co.n, QKV, c.max_pos, H, Hk, T, T, H * 128, c.eager_scores, max_keys)) return -1;
    }
    {
      GemmOut o;  // x += o_proj(attn) ; xn = post_attention_layernorm(x)   — one split-K GEMM + one reduce
      o.Y = ctx->xa; o.ldy = D; o.R = ctx->xa; o.ldr = D; o.norm_w = w.ln2; o.normed = ctx->xn; o.ldn = D; o.eps = c.rms_eps;
      o.wscale = (const float*)w.so;
      o.m_tile = co.mt(T);
      // W8A8: o_proj itself stays on bf16 activations (W8A16: the attention output is merged per head — no kernel sees a whole row to
      // quantise it — and o_proj is 5 % of the layer's weight bytes); its reduce + norm leaves gate|up's input quantised
      if (a8) { o.q8 = ctx->xq; o.sx8 = ctx->sx; }
      if (launch_gemm_ex(ctx, s, ctx->attn_o, H * 128, w.wo, nullptr, co.M(T), D, H * 128, EPI_RESIDUAL, o)) return -1;
    }
    if (a8) {
      GemmOut o;
      o.wscale = (const float*)w.sgu; o.xscale = ctx->sx; o.m_tile = co.mt(T); o.Y = ctx->act; o.ldy = I;
      if (launch_gemm_ex(ctx, s, ctx->xq, D / 2, w.wgu, nullptr, co.M(T), I, D, EPI_SWIGLU, o)) return -1;
    } else if (launch_gemm(ctx, s, ctx->xn, D, w.wgu, nullptr, ctx->act, I, nullptr, 0, co.M(T), I, D, EPI_SWIGLU, w.sgu, co.mt(T))) return -1;
    {
      GemmOut o;  // x += down(act) ; then the NEXT layer's input_layernorm, or the final model.norm (hidden_states[-1] is post-norm)
      const bool last = l + 1 == c.num_layers;
      o.Y = ctx->xa; o.ldy = D; o.R = ctx->xa; o.ldr = D; o.eps = c.rms_eps; o.ldn = D;
      o.norm_w = last ? ctx->tm.norm : ctx->layers[l + 1].ln1;
      o.normed = last ? ctx->hidden_new : ctx->xn;
      o.wscale = (const float*)w.sdown;
      o.m_tile = co.mt(T);
      if (a8) {
        if (quant(ctx->act, I)) return -1;
        o.xscale = ctx->sx;
        if (!last) { o.q8 = ctx->xq; o.sx8 = ctx->sx; }  // the next layer's q|k|v input
        if (launch_gemm_ex(ctx, s, ctx->xq, I / 2, w.wdown, nullptr, co.M(T), D, I, EPI_RESIDUAL, o)) return -1;
      } else if (launch_gemm_ex(ctx, s, ctx->act, I, w.wdown, nullptr, co.M(T), D, I, EPI_RESIDUAL, o)) return -1;
    }
  }
  if (launch_gemm(ctx, s, ctx->hidden_new, D, ctx->tm.lm_head, nullptr, ctx->logits, V, nullptr, 0, co.M(T), V, D, EPI_NONE, ctx->tm.lm_head_scale,
                  co.mt(T)))
    return -1;
  if (co.n > 1) {
    auto pk = [&](int t) { return make_pack((const bf16_t*)co.c[t]->logits, V, V, co.c[t]->am); };
    decltype(pk(0)) a[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) a[t] = pk(t);
    launch_batch<argmax_rows_fn, 1024>(s, dim3(T), 0, a, co.n);
  } else {
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(T), dim3(1024), 0, s, ctx->logits, V, V, ctx->am);
  }
  KCHK();
  return 0;
}

static int target_accept(const Cohort& co, hipStream_t s, int T, int forced_accept) {
  const vispec_config& c = co.lead()->c;
  const int D = c.hidden_size, Hk = c.num_kv_heads;
  // KV compaction (T > 1) + accept_hidden_state_new = hidden_state_new[:, retrieve_indices][:, best, :a+1] (utils.py:529-546), staged
  // for the draft's catch-up forward
  const int n_kv = T > 1 ? 2 * c.num_layers * Hk : 0;
  const bf16_t* dembed = (const bf16_t*)co.lead()->dw.embed;
  bool all_sample = true, none_sample = true;
  for (int t = 0; t < co.n; ++t) {
    const bool smp = co.c[t]->temperature > 1e-5f && T > 1 && forced_accept < 0;
    all_sample &= smp;
    none_sample &= !smp;
  }
  if (co.n > 1 && (all_sample || none_sample)) {  // one launch per step for the whole cohort
    if (all_sample) {
      auto pk = [&](int t) {
        vispec_ctx* x = co.c[t];
        return make_pack(x->tb, x->st, (const bf16_t*)x->logits, c.vocab_size, x->temperature, x->sample_top_k, x->seed, x->tokens, x->tokens_cap, x->sel,
                         x->accept_log, x->log_cap, x->draft_ids, (const float*)(x->u_over_on ? x->u_over : nullptr), 1);
      };
      decltype(pk(0)) a[MAX_COHORT];
      for (int t = 0; t < co.n; ++t) a[t] = pk(t);
      launch_batch<verify_accept_sample_fn, 1024>(s, dim3(1), 0, a, co.n);
    } else {
      auto pk = [&](int t) {
        vispec_ctx* x = co.c[t];
        return make_pack(x->tb, x->st, (const int*)x->am, x->tokens, x->tokens_cap, x->sel, x->accept_log, x->log_cap, forced_accept, x->draft_ids, 1);
      };
      decltype(pk(0)) a[MAX_COHORT];
      for (int t = 0; t < co.n; ++t) a[t] = pk(t);
      launch_batch<verify_accept_fn, 64>(s, dim3(1), 0, a, co.n);
    }
    KCHK();
    auto pp = [&](int t) {
      vispec_ctx* x = co.c[t];
      return make_pack(x->target_kv, c.max_pos, n_kv, (const DevState*)x->st, (const int*)x->sel, (const bf16_t*)x->hidden_new, x->accept_hidden,
                       (const int*)x->draft_ids, dembed, x->dx1, x->dx2, D);
    };
    decltype(pp(0)) b[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) b[t] = pp(t);
    launch_batch<post_accept_fn, 256>(s, dim3(n_kv + TREE_RET_W), 0, b, co.n);
    KCHK();
    return 0;
  }
  for (int t = 0; t < co.n; ++t) {
    vispec_ctx* ctx = co.c[t];
    if (ctx->temperature > 1e-5f && T > 1 && forced_accept < 0)
      hipLaunchKernelGGL(verify_accept_sample_kernel, dim3(1), dim3(1024), 0, s, ctx->tb, ctx->st, ctx->logits, c.vocab_size, ctx->temperature,
                         ctx->sample_top_k, ctx->seed, ctx->tokens, ctx->tokens_cap, ctx->sel, ctx->accept_log, ctx->log_cap, ctx->draft_ids,
                         (const float*)(ctx->u_over_on ? ctx->u_over : nullptr), co.n > 1 ? 1 : 0);
    else
      hipLaunchKernelGGL(verify_accept_kernel, dim3(1), dim3(64), 0, s, ctx->tb, ctx->st, ctx->am, ctx->tokens, ctx->tokens_cap,
                         ctx->sel, ctx->accept_log, ctx->log_cap, forced_accept, ctx->draft_ids, co.n > 1 ? 1 : 0);
    KCHK();
    hipLaunchKernelGGL(post_accept_kernel, dim3(n_kv + TREE_RET_W), dim3(256), 0, s, ctx->target_kv, c.max_pos, n_kv, ctx->st, ctx->sel,
                       ctx->hidden_new, ctx->accept_hidden, ctx->draft_ids, dembed, ctx->dx1, ctx->dx2, D);
    KCHK();
  }
  return 0;
}

extern "C" int vispec_verify_accept(vispec_ctx* ctx, void* stream, int forced_accept) {
  CTX_LIVE(ctx);
  hipStream_t s = (hipStream_t)stream;
  const Cohort co = solo_cohort(ctx);
  return run_graphed(ctx, s, ctx->g_verify, graph_key(ctx, forced_accept, true), [&]() {
    if (target_forward(co, s, ctx->c.total_token)) return -1;
    return target_accept(co, s, ctx->c.total_token, forced_accept);
  });
}
extern "C" int vispec_target_forward(vispec_ctx* ctx, void* stream) {
  CTX_LIVE(ctx);
  return target_forward(solo_cohort(ctx), (hipStream_t)stream, ctx->c.total_token);
}
extern "C" int vispec_accept(vispec_ctx* ctx, void* stream, int forced_accept) {
  CTX_LIVE(ctx);
  return target_accept(solo_cohort(ctx), (hipStream_t)stream, ctx->c.total_token, forced_accept);
}

// ---- cohort rounds: two to four requests (leader + member ctxs), one weight pass ------------------------------------------------------
static int cohort_check(vispec_ctx* const* ctxs, int n, Cohort* co) {
  if (!ctxs || n < 2 || n > MAX_COHORT) return fail("cohort: 2..8 requests");
  for (int t = 0; t < n; ++t)
    if (!ctxs[t]) return fail("null ctx");
  vispec_ctx* a = ctxs[0];
  if (a->leader) return fail("cohort: the first ctx must be the leader (an ordinary ctx)");
  if (a->zombie) return fail("cohort: the leader was destroyed (vispec_ctx_destroy); its members can only run as single requests");
  if (a->c.total_token > 32 && n > 4) return fail("cohort: trees of more than 32 nodes take two activation tiles per request: at most four requests per round");
  co->n = n;
  co->stride = a->c.total_token > 32 ? 64 : 32;
  for (int t = 0; t < MAX_COHORT; ++t) co->c[t] = nullptr;
  co->c[0] = a;
  // request t of the round sits in activation tile t: the members must own tiles 1 .. n-1 (any order of creation, no tile twice)
  for (int t = 1; t < n; ++t) {
    vispec_ctx* b = ctxs[t];
    if (b->leader != a) return fail("cohort: every other ctx must have been created as a member of the first (vispec_ctx_create_member)");
    if (b->c.total_token != a->c.total_token) return fail("cohort: all requests need the same tree size");
    if (b->slot < 1 || b->slot >= n) return fail("cohort: the members of an n-request round must own activation tiles 1 .. n-1 (create them in order)");
    if (co->c[b->slot]) return fail("cohort: the same member twice");
    co->c[b->slot] = b;
  }
  return 0;
}
extern "C" int vispec_cohortn_verify_accept(vispec_ctx* const* ctxs, int n, void* stream, int forced_accept) {
  Cohort co;
  if (cohort_check(ctxs, n, &co)) return -1;
  hipStream_t s = (hipStream_t)stream;
  vispec_ctx* a = co.c[0];
  return run_graphed(a, s, a->g_cverify, graph_key_n(co.c, n, forced_accept, true), [&]() {
    if (target_forward(co, s, a->c.total_token)) return -1;
    return target_accept(co, s, a->c.total_token, forced_accept);
  });
}
extern "C" int vispec_cohortn_draft_round(vispec_ctx* const* ctxs, int n, void* stream) {
  Cohort co;
  if (cohort_check(ctxs, n, &co)) return -1;
  hipStream_t s = (hipStream_t)stream;
  vispec_ctx* a = co.c[0];
  return run_graphed(a, s, a->g_cdraft, graph_key_n(co.c, n, 0, false), [&]() { return draft_round_body(co, s); });
}
extern "C" int vispec_cohort_verify_accept(vispec_ctx* a, vispec_ctx* b, void* stream, int forced_accept) {
  vispec_ctx* two[2] = {a, b};
  return vispec_cohortn_verify_accept(two, 2, stream, forced_accept);
}
extern "C" int vispec_cohort_draft_round(vispec_ctx* a, vispec_ctx* b, void* stream) {
  vispec_ctx* two[2] = {a, b};
  return vispec_cohortn_draft_round(two, 2, stream);
}
__global__ void set_tree_meta_kernel(DevState* st, int n_leaf, int max_depth, int T) {
  if (threadIdx.x == 0) { st->n_leaf = n_leaf; st->max_depth = max_depth; st->tree_T = T; }
}
static int upload_retrieve(vispec_ctx* ctx, hipStream_t s, const int* retrieve, int n_leaf, int max_depth, int T) {
  if (n_leaf < 1 || n_leaf > TREE_MAX_T || max_depth < 1 || max_depth > TREE_RET_W) return fail("bad tree shape");
  std::vector<int> ret(TREE_MAX_T * TREE_RET_W, -1);
  for (int r = 0; r < n_leaf; ++r)
    for (int j = 0; j < max_depth; ++j) {
      const int v = retrieve[r * max_depth + j];
      if (v < -1 || v >= T) return fail("retrieve index out of range");
      ret[r * TREE_RET_W + j] = v;
    }
  HIPCHK(hipMemcpyAsync(ctx->tb.retrieve, ret.data(), sizeof(int) * ret.size(), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // `ret` is a local
  hipLaunchKernelGGL(set_tree_meta_kernel, dim3(1), dim3(64), 0, s, ctx->st, n_leaf, max_depth, T);
  KCHK();
  return 0;
}
extern "C" int vispec_set_tree_host(vispec_ctx* ctx, void* stream, const int* tokens_T, const int* pos_T, const uint64_t* mask_T,
                                    const int* retrieve, int n_leaf, int max_depth) {
  if (!ctx || !tokens_T || !pos_T || !mask_T) return fail("null");
  hipStream_t s = (hipStream_t)stream;
  const int T = ctx->c.total_token;
  HIPCHK(hipMemcpyAsync(ctx->tb.tree_tokens, tokens_T, sizeof(int) * T, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(ctx->tb.tree_pos, pos_T, sizeof(int) * T, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(ctx->tb.tree_mask, mask_T, sizeof(uint64_t) * T, hipMemcpyHostToDevice, s));
  const int root_only = 0;  // retrieve == NULL: a single path holding the root (the forward needs no retrieve table; see vispec_set_retrieve_host)
  return retrieve ? upload_retrieve(ctx, s, retrieve, n_leaf, max_depth, T) : upload_retrieve(ctx, s, &root_only, 1, 1, T);
}
extern "C" int vispec_set_retrieve_host(vispec_ctx* ctx, void* stream, const int* retrieve, int n_leaf, int max_depth) {
  if (!ctx || !retrieve) return fail("null");
  return upload_retrieve(ctx, (hipStream_t)stream, retrieve, n_leaf, max_depth, ctx->c.total_token);
}

// Tests only: the uniforms of the sampling accept (utils.py:453-493: torch.rand_like per candidate row and level; one more for the final
// multinomial) from a host table instead of the counter-based generator — the way the reference's RECORDED draws (tests/golden g7) are fed
// to verify_accept_sample_kernel.  u = [n_leaf, max_depth] row-major; u == NULL switches the override off.  Blocking.
extern "C" int vispec_set_uniform_override_host(vispec_ctx* ctx, void* stream, const float* u, int n_leaf, int max_depth, float u_final) {
  CTX_LIVE(ctx);
  if (!u) { ctx->u_over_on = false; return 0; }
  if (n_leaf < 1 || n_leaf > TREE_MAX_T || max_depth < 1 || max_depth > TREE_RET_W) return fail("uniform_override: bad shape");
  std::vector<float> tab(TREE_MAX_T * TREE_RET_W + 1, 2.0f);  // (2 = never accepted)
  for (int j = 0; j < n_leaf; ++j)
    for (int i = 0; i < max_depth; ++i) tab[j * TREE_RET_W + i] = u[j * max_depth + i];
  tab[TREE_MAX_T * TREE_RET_W] = u_final;
  HIPCHK(hipMemcpyAsync(ctx->u_over, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  ctx->u_over_on = true;
  return 0;
}

__global__ void set_stop2_kernel(DevState* st, int tok) {
  if (threadIdx.x == 0) st->stop2 = tok;
}
// is_llama3: "<|eot_id|>" among the generated ids also ends the request (spec_model_ours.py:268-269, 540-542); after begin_request
extern "C" int vispec_set_stop_token(vispec_ctx* ctx, void* stream, int token_id) {
  CTX_LIVE(ctx);
  hipLaunchKernelGGL(set_stop2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctx->st, token_id);
  KCHK();
  return 0;
}
__global__ void set_rope_delta_kernel(DevState* st, int delta) {
  if (threadIdx.x == 0) st->rope_delta = delta;
}
// spec_model_ours.py:179-201 changes the tree size after construction (`model.spec_layer.total_tokens = total_token - 1`)
extern "C" int vispec_set_total_token(vispec_ctx* ctx, int total_token) {
  CTX_LIVE(ctx);
  const vispec_config& c = ctx->c;
  if (total_token < 1 || total_token > TREE_MAX_T) return fail("total_token must be in [1,64]");
  if (total_token - 1 > c.top_k + c.depth * c.top_k * c.top_k) return fail("total_token larger than the candidate pool");
  if (ctx->leader && total_token > 32 && ctx->slot > 3)
    return fail("trees of more than 32 nodes take two activation tiles per request: only the first four request slots of a cohort can hold them");
  ctx->c.total_token = total_token;
  retarget_views(ctx);  // a member's target-side views: 32 or 64 rows per request slot (every ctx of a cohort round must agree: cohort_check)
  // (the tree size is part of every graph key: the captured launch sequences — and the views above — depend on it)
  return 0;
}
extern "C" int vispec_set_rope_delta(vispec_ctx* ctx, void* stream, int delta) {
  CTX_LIVE(ctx);
  hipLaunchKernelGGL(set_rope_delta_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctx->st, delta);
  KCHK();
  return 0;
}

// temperature <= 1e-5: greedy (utils.py:438-451); > 1e-5: sampling (utils.py:453-493) with the counter-based uniforms of `seed`.
extern "C" int vispec_set_sampling(vispec_ctx* ctx, float temperature, unsigned long long seed) {
  CTX_LIVE(ctx);
  ctx->temperature = temperature;
  ctx->seed = seed;
  ctx->sample_top_k = 0;
  return 0;
}
// token = multinomial(softmax(row / T)) of one bf16 logits row (first token of a sampled request, utils.py:284-288)
// TopKLogitsWarper of prepare_logits_processor (utils.py:52-53); call after vispec_set_sampling (which resets it to 0 = off)
extern "C" int vispec_set_top_k(vispec_ctx* ctx, int top_k) {
  if (!ctx || top_k < 0) return fail("set_top_k: bad arguments");
  ctx->sample_top_k = top_k;
  return 0;
}
extern "C" int vispec_sample_row(vispec_ctx* ctx, void* stream, const void* logits_row, int V, int* out_token_dev) {
  if (!ctx || ctx->temperature <= 1e-5f) return fail("sample_row: sampling not enabled (vispec_set_sampling)");
  hipLaunchKernelGGL(sample_row_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)logits_row, V, ctx->temperature, ctx->sample_top_k,
                     ctx->seed,
                     out_token_dev);
  KCHK();
  return 0;
}

extern "C" int vispec_set_next_token(vispec_ctx* ctx, void* stream, const int* token_dev) {
  if (!ctx || !token_dev) return fail("null");
  hipLaunchKernelGGL(set_first_token_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctx->st, token_dev, 0, 0);
  KCHK();
  return 0;
}

extern "C" int vispec_ar_step(vispec_ctx* ctx, void* stream) {
  CTX_LIVE(ctx);
  hipStream_t s = (hipStream_t)stream;
  return run_graphed(ctx, s, ctx->g_ar, graph_key(ctx, -1, false), [&]() {
    hipLaunchKernelGGL(tree_single_kernel, dim3(1), dim3(64), 0, s, ctx->tb, ctx->st);
    KCHK();
    const Cohort co = solo_cohort(ctx);
    if (target_forward(co, s, 1)) return -1;
    return target_accept(co, s, 1, -1);
  });
}

// The AR baseline at the cohort's batching (gen_baseline_answer_coco_caption.py:34-133 is batch 1 per request, like everything in the
// reference; the speed-up bench.py prints divides like by like: n speculative requests per weight pass by n AR requests per weight pass):
// one greedy token for each of the n requests, the target's GEMMs launched ONCE for all of them (m_tile = 1).  Row for row the
// arithmetic of vispec_ar_step, so a request's AR tokens do not depend on its cohort; a finished request freezes like in a cohort round.
extern "C" int vispec_cohortn_ar_step(vispec_ctx* const* ctxs, int n, void* stream) {
  Cohort co;
  if (cohort_check(ctxs, n, &co)) return -1;
  hipStream_t s = (hipStream_t)stream;
  vispec_ctx* a = co.c[0];
  return run_graphed(a, s, a->g_car, graph_key_n(co.c, n, -1, false), [&]() {
    auto pk = [&](int t) { return make_pack(co.c[t]->tb, co.c[t]->st); };
    decltype(pk(0)) b[MAX_COHORT];
    for (int t = 0; t < co.n; ++t) b[t] = pk(t);
    launch_batch<tree_single_fn, 64>(s, dim3(1), 0, b, co.n);
    KCHK();
    if (target_forward(co, s, 1)) return -1;
    return target_accept(co, s, 1, -1);
  });
}

// ------------------------------------------------------------------------------------------------ read-back
extern "C" int vispec_get_state_host(vispec_ctx* ctx, void* stream, int* out) {
  CTX_LIVE(ctx);
  if (!ctx || !out) return fail("null");
  DevState h;
  HIPCHK(hipMemcpyAsync(&h, ctx->st, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  out[0] = h.n_ctx; out[1] = h.new_token; out[2] = h.rounds; out[3] = h.done; out[4] = h.accept_len;
  out[5] = h.next_token; out[6] = h.draft_len; out[7] = h.n_leaf;
  return 0;
}
// The same for every request of a cohort with ONE stream synchronisation (the per-request form costs a blocking round trip each: four per
// lockstep round): the states travel through each ctx's pinned staging buffer.  out = n x 8 ints, laid out as above.
extern "C" int vispec_cohort_get_state_host(vispec_ctx* const* ctxs, int n, void* stream, int* out) {
  if (ctxs && n >= 1 && n <= MAX_COHORT) for (int t_ = 0; t_ < n; ++t_) CTX_LIVE(ctxs[t_]);
  if (!ctxs || !out || n < 1 || n > MAX_COHORT) return fail("cohort_get_state: 1..8 contexts");
  for (int t = 0; t < n; ++t) {
    if (!ctxs[t] || !ctxs[t]->h_pin) return fail("null ctx");
    HIPCHK(hipMemcpyAsync(ctxs[t]->h_pin, ctxs[t]->st, sizeof(DevState), hipMemcpyDeviceToHost, (hipStream_t)stream));
  }
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  for (int t = 0; t < n; ++t) {
    const DevState& h = *reinterpret_cast<const DevState*>(ctxs[t]->h_pin);
    int* o = out + 8 * t;
    o[0] = h.n_ctx; o[1] = h.new_token; o[2] = h.rounds; o[3] = h.done; o[4] = h.accept_len;
    o[5] = h.next_token; o[6] = h.draft_len; o[7] = h.n_leaf;
  }
  return 0;
}
// The same in two halves, for a host loop that keeps one round of lookahead (model/spec_model_ours.py: specgenerate_stream): `enqueue` copies
// every request's DevState into pinned snapshot slot `slot` (0 / 1) in stream order and records the slot's event — the next round can be
// launched right behind it; `wait` blocks on that event only (not on the stream, which is already running the next round) and unpacks.
extern "C" int vispec_cohort_state_enqueue(vispec_ctx* const* ctxs, int n, void* stream, int slot) {
  if (ctxs && n >= 1 && n <= MAX_COHORT) for (int t_ = 0; t_ < n; ++t_) CTX_LIVE(ctxs[t_]);
  if (!ctxs || n < 1 || n > MAX_COHORT || slot < 0 || slot > 1) return fail("cohort_state_enqueue: 1..8 contexts, slot 0 or 1");
  for (int t = 0; t < n; ++t) {
    if (!ctxs[t]) return fail("null ctx");
    if (!ctxs[t]->h_state && hipHostMalloc(&ctxs[t]->h_state, 2 * sizeof(DevState)) != hipSuccess) return fail("hipHostMalloc failed");
    HIPCHK(hipMemcpyAsync(static_cast<DevState*>(ctxs[t]->h_state) + slot, ctxs[t]->st, sizeof(DevState), hipMemcpyDeviceToHost, (hipStream_t)stream));
  }
  vispec_ctx* lead = ctxs[0];
  if (!lead->st_ev[slot]) HIPCHK(hipEventCreateWithFlags(&lead->st_ev[slot], hipEventDisableTiming));
  HIPCHK(hipEventRecord(lead->st_ev[slot], (hipStream_t)stream));
  return 0;
}
extern "C" int vispec_cohort_state_wait(vispec_ctx* const* ctxs, int n, int slot, int* out) {
  if (ctxs && n >= 1 && n <= MAX_COHORT) for (int t_ = 0; t_ < n; ++t_) CTX_LIVE(ctxs[t_]);
  if (!ctxs || !out || n < 1 || n > MAX_COHORT || slot < 0 || slot > 1 || !ctxs[0] || !ctxs[0]->st_ev[slot]) return fail("cohort_state_wait: nothing enqueued in this slot");
  HIPCHK(hipEventSynchronize(ctxs[0]->st_ev[slot]));
  for (int t = 0; t < n; ++t) {
    if (!ctxs[t] || !ctxs[t]->h_state) return fail("null ctx");
    const DevState& h = static_cast<const DevState*>(ctxs[t]->h_state)[slot];
    int* o = out + 8 * t;
    o[0] = h.n_ctx; o[1] = h.new_token; o[2] = h.rounds; o[3] = h.done; o[4] = h.accept_len;
    o[5] = h.next_token; o[6] = h.draft_len; o[7] = h.n_leaf;
  }
  return 0;
}
extern "C" int vispec_get_last_accept_host(vispec_ctx* ctx, void* stream, int* out2) {
  CTX_LIVE(ctx);
  if (!ctx || !out2) return fail("null");
  DevState h;
  HIPCHK(hipMemcpyAsync(&h, ctx->st, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  out2[0] = h.best;
  out2[1] = h.accept_len;
  return 0;
}
extern "C" int vispec_get_tokens_host(vispec_ctx* ctx, void* stream, int* out, int n) {
  CTX_LIVE(ctx);
  if (!ctx || !out || n < 0 || n > ctx->tokens_cap) return fail("bad args");
  HIPCHK(hipMemcpyAsync(out, ctx->tokens, sizeof(int) * n, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int vispec_get_accept_log_host(vispec_ctx* ctx, void* stream, int* out, int n) {
  CTX_LIVE(ctx);
  if (!ctx || !out || n < 0 || n > ctx->log_cap) return fail("bad args");
  HIPCHK(hipMemcpyAsync(out, ctx->accept_log, sizeof(int) * n, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int vispec_get_tree_host(vispec_ctx* ctx, void* stream, int* tokens_T, int* pos_T, uint64_t* mask_T, int* retrieve,
                                    int* n_leaf, int* max_depth) {
  CTX_LIVE(ctx);
  if (!ctx) return fail("null");
  hipStream_t s = (hipStream_t)stream;
  DevState h;
  HIPCHK(hipMemcpyAsync(&h, ctx->st, sizeof(h), hipMemcpyDeviceToHost, s));
  if (tokens_T) HIPCHK(hipMemcpyAsync(tokens_T, ctx->tb.tree_tokens, sizeof(int) * TREE_MAX_T, hipMemcpyDeviceToHost, s));
  if (pos_T) HIPCHK(hipMemcpyAsync(pos_T, ctx->tb.tree_pos, sizeof(int) * TREE_MAX_T, hipMemcpyDeviceToHost, s));
  if (mask_T) HIPCHK(hipMemcpyAsync(mask_T, ctx->tb.tree_mask, sizeof(uint64_t) * TREE_MAX_T, hipMemcpyDeviceToHost, s));
  if (retrieve) HIPCHK(hipMemcpyAsync(retrieve, ctx->tb.retrieve, sizeof(int) * TREE_MAX_T * TREE_RET_W, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (n_leaf) *n_leaf = h.n_leaf;
  if (max_depth) *max_depth = h.max_depth;
  return 0;
}

extern "C" void* vispec_buffer(vispec_ctx* ctx, const char* name) {
  if (!ctx || ctx->zombie || !name) return nullptr;
  struct E { const char* n; void* p; };
  const E tab[] = {{"state", ctx->st}, {"tokens", ctx->tokens}, {"hidden_new", ctx->hidden_new}, {"logits", ctx->logits},
                   {"am", ctx->am}, {"sel", ctx->sel}, {"draft_ids", ctx->draft_ids}, {"accept_hidden", ctx->accept_hidden},
                   {"draft_last", ctx->dlast}, {"draft_out", ctx->dout}, {"draft_logits", ctx->dlogits}, {"draft_g", ctx->dg},
                   {"draft_xc", ctx->xc}, {"tree_tokens", ctx->tb.tree_tokens}, {"tree_pos", ctx->tb.tree_pos},
                   {"tree_mask", ctx->tb.tree_mask}, {"retrieve", ctx->tb.retrieve}, {"scores_all", ctx->tb.scores_all},
                   {"tokens_all", ctx->tb.tokens_all}, {"parents_all", ctx->tb.parents_all}, {"accept_log", ctx->accept_log},
                   {"lvl_mask", ctx->tb.lvl_mask}, {"in_ids", ctx->tb.in_ids}};
  for (const E& e : tab)
    if (!strcmp(e.n, name)) return e.p;
  return nullptr;
}
