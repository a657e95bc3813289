// Token-tree bookkeeping on the device (integer logic, exact restatement of the reference's host code):
//   draft side  : cnets_ours.py:1109-1238  (level expansion, global re-rank, parents, mask, positions, retrieve table)
//   target side : utils.py:415-451 (greedy evaluate_posterior) and utils.py:496-556 (update_inference_inputs)
// Every kernel here is a single small workgroup; they exist so that a round never returns to the host.
#pragma once
#include "kernels.h"

#define TREE_MAX_T 64      // nodes in the verify tree (total_token <= 64 -> one 64-bit mask word per row)
#define TREE_MAX_K 16      // top_k
#define TREE_MAX_DEPTH 8   // depth
#define TREE_MAX_SCORES (TREE_MAX_K + TREE_MAX_DEPTH * TREE_MAX_K * TREE_MAX_K)
#define TREE_RET_W (TREE_MAX_DEPTH + 2)  // width of a retrieve row (root + depth+1 levels)

struct TreeBufs {
  // growing lists (cnets_ours.py:1062-1064)
  float* scores_all;   // [k + depth*k*k]   bf16-rounded cumulative log-probs
  int* tokens_all;     // [k + depth*k*k]
  int* parents_all;    // [1 + depth*k]
  // frontier
  float* cur_scores;   // [k]
  int* cs_idx;         // [k]   topk_cs_index of the previous level
  int* in_ids;         // [k]   tokens fed to the next level forward
  unsigned long long* lvl_mask;  // [k]  visibility of the tail keys for the next level forward
  // finished tree (what verify consumes)
  int* tree_tokens;    // [T]
  int* tree_pos;       // [T]   depth of each node (tree_position_ids)
  unsigned long long* tree_mask;  // [T]
  int* retrieve;       // [T][TREE_RET_W]  padded with -1
  int* mask_index;     // [T]  parent row of node i+1 (scratch, kept for the tests)
};

// The input rows of a level forward are staged where the fused input GEMMs read them: dx1[r, 0:D] = input hidden of frontier node r
// (row stride 2D; the right half holds the global image feature g, constant during a request) and dx2[r, 0:D] = embed(in_ids[r]).
__device__ __forceinline__ void tree_stage_row(const bf16_t* __restrict__ hid_src, const bf16_t* __restrict__ emb_src, bf16_t* __restrict__ dx1,
                                               bf16_t* __restrict__ dx2, int r, int D, int tid, int nthreads) {
  for (int d = tid * 8; d < D; d += nthreads * 8) {
    *reinterpret_cast<uint4*>(dx1 + (size_t)r * 2 * D + d) = *reinterpret_cast<const uint4*>(hid_src + d);
    *reinterpret_cast<uint4*>(dx2 + (size_t)r * 2 * D + d) = *reinterpret_cast<const uint4*>(emb_src + d);
  }
}

// Level 1 (cnets_ours.py:1114-1123): the k children of the root come from the last hidden state's top-k.
__device__ __forceinline__ void tree_init_body(TreeBufs tb, const int* __restrict__ top_idx, const float* __restrict__ top_logp, int k,
                                                         const bf16_t* __restrict__ last_hidden, const bf16_t* __restrict__ embed,
                                                         bf16_t* __restrict__ dx1, bf16_t* __restrict__ dx2, int D) {
  const int tid = threadIdx.x;
  if (tid < k) {
    tb.scores_all[tid] = top_logp[tid];
    tb.tokens_all[tid] = top_idx[tid];
    tb.cur_scores[tid] = top_logp[tid];
    tb.cs_idx[tid] = tid;
    tb.in_ids[tid] = top_idx[tid];
    tb.lvl_mask[tid] = 1ull << tid;  // tree_mask_init = eye(k)
  }
  if (tid == 0) tb.parents_all[0] = 0;
  // input_hidden = last_hidden.repeat(k) ; input_ids = topk_index   (:1120-1121)
  const int per = 1024 / k;  // threads per row
  const int r = tid / per;
  if (r < k) tree_stage_row(last_hidden, embed + (size_t)top_idx[r] * D, dx1, dx2, r, D, tid % per, per);
}
__global__ __launch_bounds__(1024) void tree_init_kernel(TreeBufs tb, const int* __restrict__ top_idx, const float* __restrict__ top_logp, int k,
                                                         const bf16_t* __restrict__ last_hidden, const bf16_t* __restrict__ embed,
                                                         bf16_t* __restrict__ dx1, bf16_t* __restrict__ dx2, int D) { tree_init_body(tb, top_idx, top_logp, k, last_hidden, embed, dx1, dx2, D); }
struct tree_init_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { tree_init_body(a...); }
};

// One tree level (cnets_ours.py:1139-1165) after the level forward + LM head + per-row top-k.
__device__ __forceinline__ void tree_level_body(TreeBufs tb, int level, int k, const int* __restrict__ top_idx,
                                                          const float* __restrict__ top_logp, const bf16_t* __restrict__ out_hidden,
                                                          const bf16_t* __restrict__ embed, bf16_t* __restrict__ dx1,
                                                          bf16_t* __restrict__ dx2, int D) {
  __shared__ float cu[TREE_MAX_K * TREE_MAX_K];
  __shared__ int sel[TREE_MAX_K];
  __shared__ unsigned long long newmask[TREE_MAX_K];
  const int tid = threadIdx.x, kk = k * k;
  const int base = k + level * kk;
  if (tid < k) {
    const int bias = 1 + kk * max(0, level - 1) + (level > 0 ? k : 0);
    tb.parents_all[1 + level * k + tid] = tb.cs_idx[tid] + bias;
  }
  if (tid < kk) {
    const float c = rdbf(top_logp[tid] + tb.cur_scores[tid / k]);  // cu_scores = topk_p + scores[:, None] (bf16 add)
    cu[tid] = c;
    tb.scores_all[base + tid] = c;
    tb.tokens_all[base + tid] = top_idx[tid];
  }
  __syncthreads();
  if (tid < kk) {  // rank by (value desc, index asc)
    const float c = cu[tid];
    int rank = 0;
    for (int o = 0; o < kk; ++o) rank += (cu[o] > c) || (cu[o] == c && o < tid);
    if (rank < k) sel[rank] = tid;
  }
  __syncthreads();
  if (tid < k) {
    const int s = sel[tid], parent_row = s / k;
    newmask[tid] = tb.lvl_mask[parent_row] | (1ull << (k * (level + 1) + tid));
  }
  __syncthreads();
  if (tid < k) {
    const int s = sel[tid];
    tb.cur_scores[tid] = cu[s];
    tb.cs_idx[tid] = s;
    tb.in_ids[tid] = top_idx[s];
    tb.lvl_mask[tid] = newmask[tid];
  }
  // input_hidden = out_hidden[:, out_ids] ; input_ids = topk_index.view(-1)[topk_cs_index]   (:1157-1159)
  const int per = 1024 / k;
  const int r = tid / per;
  if (r < k) tree_stage_row(out_hidden + (size_t)(sel[r] / k) * D, embed + (size_t)top_idx[sel[r]] * D, dx1, dx2, r, D, tid % per, per);
}
__global__ __launch_bounds__(1024) void tree_level_kernel(TreeBufs tb, int level, int k, const int* __restrict__ top_idx,
                                                          const float* __restrict__ top_logp, const bf16_t* __restrict__ out_hidden,
                                                          const bf16_t* __restrict__ embed, bf16_t* __restrict__ dx1,
                                                          bf16_t* __restrict__ dx2, int D) { tree_level_body(tb, level, k, top_idx, top_logp, out_hidden, embed, dx1, dx2, D); }
struct tree_level_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { tree_level_body(a...); }
};

// Global re-rank and tree construction (cnets_ours.py:1167-1213).  total = total_token-1.
__device__ __forceinline__ void tree_finalize_body(TreeBufs tb, DevState* st, int k, int depth, int total,
                                                            int sampling) {
  __shared__ float sc[TREE_MAX_SCORES];
  __shared__ unsigned char keep[TREE_MAX_SCORES];
  __shared__ int top_idx[TREE_MAX_T];
  __shared__ int midx[TREE_MAX_T];
  __shared__ unsigned long long rows[TREE_MAX_T];
  __shared__ unsigned char nonleaf[TREE_MAX_T];
  const int tid = threadIdx.x;
  const int n_all = k + depth * k * k, T = total + 1;
  for (int e = tid; e < n_all; e += 256) sc[e] = tb.scores_all[e];
  __syncthreads();
  for (int e = tid; e < n_all; e += 256) {  // top-`total` by (value desc, index asc)   :1169
    const float c = sc[e];
    int rank = 0;
    for (int o = 0; o < n_all; ++o) rank += (sc[o] > c) || (sc[o] == c && o < e);
    keep[e] = rank < total;
  }
  __syncthreads();
  for (int e = tid; e < n_all; e += 256)  // ascending flat index = torch.sort(top_scores_index)   :1171
    if (keep[e]) {
      int p = 0;
      for (int o = 0; o < e; ++o) p += keep[o];
      top_idx[p] = e;
    }
  __syncthreads();
  if (tid < total) {
    const int e = top_idx[tid];
    tb.tree_tokens[1 + tid] = tb.tokens_all[e];  // :1173-1174
    const int par = tb.parents_all[e / k];       // :1176
    int mi;
    if (par == 0) mi = 0;                        // :1180-1181
    else {                                       // searchsorted(top_idx, par-1, right=False) + 1   :1177-1181
      const int v = par - 1;
      int lo = 0, hi2 = total;
      while (lo < hi2) { int mid = (lo + hi2) >> 1; if (top_idx[mid] < v) lo = mid + 1; else hi2 = mid; }
      mi = lo + 1;
    }
    midx[tid] = mi;
    tb.mask_index[tid] = mi;
  }
  if (tid == 0) tb.tree_tokens[0] = st->next_token;  // sample_token is the root   :1060,1174
  if (tid < T) nonleaf[tid] = 0;
  __syncthreads();
  if (tid == 0) {  // ancestors-or-self, sequential like the reference loop   :1183-1186
    rows[0] = 1ull;
    for (int i = 0; i < total; ++i) {
      const int mi = midx[i];
      unsigned long long r = (1ull << (i + 1)) | 1ull;
      if (mi <= i) r |= rows[mi];  // (mi == i+1 would be the row itself: no-op)
      rows[i + 1] = r;
    }
  }
  if (tid < total) {
    const int mi = midx[tid];
    if (mi < T) nonleaf[mi] = 1;  // noleaf_index = unique(mask_index)   :1196
  }
  __syncthreads();
  if (tid < T) {
    tb.tree_mask[tid] = rows[tid];
    tb.tree_pos[tid] = __popcll(rows[tid]) - 1;  // :1188
  }
  // leaf paths (:1195-1213), one thread per leaf, built in LDS; with sampling the rows are then sorted lexicographically with -1 -> large
  // (:1215-1224) by a rank computed in parallel (the paths are distinct, so the order is total) — the first form of this tail was a
  // single thread insertion-sorting rows in global memory: 250 us per round on the sampling path, 21 us greedy
  __shared__ int s_ret[TREE_MAX_T][TREE_RET_W];
  __shared__ int s_nleaf, s_maxd;
  if (tid == 0) {
    int maxd = 0, nl = 0;
    for (int i = 0; i < T; ++i) {
      maxd = max(maxd, __popcll(rows[i]) - 1);
      nl += nonleaf[i] ? 0 : 1;
    }
    s_maxd = maxd;
    s_nleaf = nl;
  }
  for (int e = tid; e < TREE_MAX_T * TREE_RET_W; e += 256) s_ret[e / TREE_RET_W][e % TREE_RET_W] = -1;
  __syncthreads();
  if (tid < T && !nonleaf[tid]) {
    int rid = 0;
    for (int i = 0; i < tid; ++i) rid += nonleaf[i] ? 0 : 1;  // leaves keep their node order
    int cid = tid;
    for (int jj = __popcll(rows[tid]) - 1; jj >= 0; --jj) {
      s_ret[rid][jj] = cid;
      cid = (cid > 0) ? midx[cid - 1] : 0;
    }
  }
  __syncthreads();
  const int n_leaf = s_nleaf, maxd = s_maxd;
  for (int e = tid; e < TREE_MAX_T * TREE_RET_W; e += 256) {
    const int a = e / TREE_RET_W, c = e % TREE_RET_W;
    int dest = a;
    if (sampling && a < n_leaf && n_leaf > 1) {
      const int big = total + 5;
      dest = 0;
      for (int b = 0; b < n_leaf; ++b) {
        if (b == a) continue;
        bool lt = b < a;  // equal rows (cannot happen: paths are distinct) would keep their order
        for (int cc = 0; cc < maxd + 1; ++cc) {
          int x = s_ret[b][cc], y = s_ret[a][cc];
          x = x >= 0 ? x : big; y = y >= 0 ? y : big;
          if (x != y) { lt = x < y; break; }
        }
        dest += lt ? 1 : 0;
      }
    }
    tb.retrieve[dest * TREE_RET_W + c] = s_ret[a][c];  // rows >= n_leaf are all -1 and map to themselves
  }
  if (tid == 0) {
    st->n_leaf = n_leaf;
    st->max_depth = maxd + 1;
    st->tree_T = T;
  }
}
__global__ __launch_bounds__(256) void tree_finalize_kernel(TreeBufs tb, DevState* st, int k, int depth, int total,
                                                            int sampling) { tree_finalize_body(tb, st, k, depth, total, sampling); }
struct tree_finalize_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { tree_finalize_body(a...); }
};

// A one-node "tree" = plain autoregressive decoding with the same verify kernels (baseline_forward).
__device__ __forceinline__ void tree_single_body(TreeBufs tb, DevState* st) {
  if (threadIdx.x == 0) {
    tb.tree_tokens[0] = st->next_token;
    tb.tree_pos[0] = 0;
    tb.tree_mask[0] = 1ull;
    for (int c = 0; c < TREE_RET_W; ++c) tb.retrieve[c] = c == 0 ? 0 : -1;
    st->n_leaf = 1;
    st->max_depth = 1;
    st->tree_T = 1;
  }
}
__global__ void tree_single_kernel(TreeBufs tb, DevState* st) { tree_single_body(tb, st); }
struct tree_single_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { tree_single_body(a...); }
};

// Greedy evaluate_posterior (utils.py:438-451) + the integer part of update_inference_inputs (utils.py:514-526,541,554,582)
// am[i] = argmax of the target logits at tree node i.  sel[j] = tree node accepted at depth j (j = 0..a).
__device__ __forceinline__ void verify_accept_body(TreeBufs tb, DevState* st, const int* __restrict__ am, int* __restrict__ tokens,
                                     int tokens_cap, int* __restrict__ sel, int* __restrict__ accept_log, int log_cap,
                                     int forced_accept, int* __restrict__ draft_ids, int cohort) {
  __shared__ int acc[TREE_MAX_T];
  const int tid = threadIdx.x;
  // cohort rounds only: a finished request (its partners are still running) freezes.  The single-request entry points keep
  // stepping past the EOS / budget flags — a caller driving the step API with its own stop rule gets fresh results every step — but
  // NOT past bit 2 (the KV cache is full): the next tree's rows would be written beyond the cache slab, so that flag freezes always.
  if ((cohort && st->done) || (st->done & 4)) {
    __syncthreads();
    if (tid == 0) st->frozen = 1;
    return;
  }
  const int n_leaf = st->n_leaf, md = st->max_depth;
  if (tid < n_leaf) {
    const int* row = tb.retrieve + tid * TREE_RET_W;
    int a = 0;
    for (int jj = 0; jj + 1 < md; ++jj) {
      const int nxt = row[jj + 1];
      const int cand = nxt >= 0 ? tb.tree_tokens[nxt] : -1;  // draft_tokens padded with -1   spec_model_ours.py:503-504
      if (cand == am[row[jj]]) ++a; else break;             // cumprod of the posterior mask
    }
    if (forced_accept >= 0) {  // bench-only scripted acceptance: pretend the first `forced` draft tokens of the path matched
      int len = 0;
      for (int jj = 0; jj < md; ++jj) len += row[jj] >= 0;
      a = min(forced_accept, len - 1);
    }
    acc[tid] = a;
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0, a = 0;
    for (int r = 0; r < n_leaf; ++r)
      if (acc[r] > a) { a = acc[r]; best = r; }  // first max; 0 if none accepted
    const int* row = tb.retrieve + best * TREE_RET_W;
    const int n = st->n_ctx;
    int done = st->done;
    for (int jj = 0; jj <= a; ++jj) {
      const int node = row[jj];
      sel[jj] = node;
      const int tok = tb.tree_tokens[node];
      if (n + jj < tokens_cap) tokens[n + jj] = tok;
      if (tok == st->eos_token_id || tok == st->stop2) done |= 1;
    }
    for (int jj = a + 1; jj < TREE_RET_W; ++jj) sel[jj] = row[a];
    const int next = am[row[a]];  // token = argmax(sample_p), sample_p = logits[best, accept_length]
    // ids the draft's catch-up rows are paired with: accepted tokens 1..a then the new sample (cnets_ours.py:1084,1093)
    for (int jj = 0; jj < TREE_RET_W; ++jj)
      draft_ids[jj] = jj < a ? tb.tree_tokens[row[jj + 1]] : next;
    st->n_prev = n;
    st->n_ctx = n + a + 1;
    st->accept_len = a;
    st->best = best;
    st->next_token = next;
    st->new_token += a + 1;
    if (st->new_token > st->max_new_tokens) done |= 2;
    if (st->n_ctx + TREE_MAX_T + KV_GUARD_ROWS > st->kv_cap) done |= 4;
    st->done = done;
    if (st->rounds < log_cap) accept_log[st->rounds] = a;
    st->rounds += 1;
  }
}
__global__ void verify_accept_kernel(TreeBufs tb, DevState* st, const int* __restrict__ am, int* __restrict__ tokens,
                                     int tokens_cap, int* __restrict__ sel, int* __restrict__ accept_log, int log_cap,
                                     int forced_accept, int* __restrict__ draft_ids, int cohort) { verify_accept_body(tb, st, am, tokens, tokens_cap, sel, accept_log, log_cap, forced_accept, draft_ids, cohort); }
struct verify_accept_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { verify_accept_body(a...); }
};

// ------------------------------------------------------------------------------------------------
// Sampling path (temperature > 0): utils.py:453-493 (evaluate_posterior, sequential rejection over the tree's children),
// :284-288 / :549-552 (multinomial for the first / next token).  Randomness is explicit and counter-based so that the oracle
// draws the very same numbers (oracle/vispec_oracle.py: uniform_hash); parity with the reference's torch RNG is distributional.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float vs_uniform(unsigned long long seed, unsigned a, unsigned b, unsigned c) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + a * 0xBF58476D1CE4E5B9ull + b * 0x94D049BB133111EBull + c * 0xD6E8FEB86659FD93ull +
                         0x2545F4914F6CDD1Dull;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// TopKLogitsWarper threshold (utils.py:52-53): the k-th largest logit of the row; scores < thr are filtered, ties with thr survive.
// Logits are bf16, so the k-th value is found exactly with two 256-bin histogram passes over an order-preserving 16-bit key.
__device__ __forceinline__ unsigned vs_key(bf16_t b) { return (b & 0x8000u) ? (~(unsigned)b & 0xFFFFu) : ((unsigned)b | 0x8000u); }
__device__ __forceinline__ float vs_topk_threshold(const bf16_t* __restrict__ row, int V, int top_k, int* s_hist) {
  const int tid = threadIdx.x;
  if (top_k <= 0 || top_k >= V) return NEG_INF;
  int want = top_k, hb = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    for (int v = tid; v < V; v += 1024) {
      const unsigned k = vs_key(row[v]);
      if (pass == 0) atomicAdd(&s_hist[k >> 8], 1);
      else if ((int)(k >> 8) == hb) atomicAdd(&s_hist[k & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int b = 255, acc = 0;
      while (b > 0 && acc + s_hist[b] < want) { acc += s_hist[b]; --b; }
      s_hist[256] = b;
      s_hist[257] = want - acc;
    }
    __syncthreads();
    const int b = s_hist[256];
    want = s_hist[257];
    __syncthreads();
    if (pass == 0) hb = b;
    else {
      const unsigned key = ((unsigned)hb << 8) | (unsigned)b;
      const bf16_t bits = (key & 0x8000u) ? (bf16_t)(key & 0x7FFFu) : (bf16_t)(~key & 0xFFFFu);
      return bf2f(bits);
    }
  }
  return NEG_INF;
}

// block-wide (1024 threads) max and sum-exp of logits_processor(row) = row/T restricted to row >= thr  (softmax in fp32)
__device__ __forceinline__ void vs_row_stats(const bf16_t* __restrict__ row, int V, float T, float thr, float* s_f, float& m, float& Z) {
  const int tid = threadIdx.x;
  float mx = NEG_INF;
  for (int v = tid; v < V; v += 1024) mx = fmaxf(mx, bf2f(row[v]) / T);
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_f[tid >> 6] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_f[w]);
  __syncthreads();
  float se = 0.f;
  for (int v = tid; v < V; v += 1024) {
    const float x = bf2f(row[v]);
    if (x >= thr) se += expf(x / T - mx);
  }
  se = wave_sum(se);
  if ((tid & 63) == 0) s_f[tid >> 6] = se;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < 16; ++w) tot += s_f[w];
  __syncthreads();
  m = mx;
  Z = tot;
}

// one multinomial draw by inverse CDF over weights exp(row/T - m) with `nrem` removed tokens (torch.multinomial stand-in)
__device__ __forceinline__ int vs_multinomial(const bf16_t* __restrict__ row, int V, float T, float m, float thr, const int* removed, int nrem,
                                              float u, double* s_d, int* s_i) {
  const int tid = threadIdx.x;
  const int chunk = (V + 1023) / 1024;
  const int lo = tid * chunk, hi = min(V, lo + chunk);
  double loc = 0.0;
  for (int v = lo; v < hi; ++v) {
    bool rem = bf2f(row[v]) < thr;
    for (int r = 0; r < nrem; ++r) rem |= removed[r] == v;
    if (!rem) loc += (double)expf(bf2f(row[v]) / T - m);
  }
  // exclusive scan over the 1024 partial sums + locate the chunk holding the target mass: wave-level shuffles, 16 wave totals through LDS
  // (the first form did this on one thread: 1024 dependent LDS round trips, ~27 us per draw)
  double inc = loc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double up = __shfl_up(inc, d, 64);
    if ((tid & 63) >= d) inc += up;
  }
  double* s_w = s_d + 1026;  // 16 wave totals (s_d has 1025 + 17 entries, see the callers)
  if ((tid & 63) == 63) s_w[tid >> 6] = inc;
  __syncthreads();
  double off = 0.0, run = 0.0;
  for (int w = 0; w < 16; ++w) {  // fixed order
    const double x = s_w[w];
    if (w < (tid >> 6)) off += x;
    run += x;
  }
  const double target = (double)u * run;
  // exclusive prefix of a chunk = the INCLUSIVE prefix of its left neighbour (never inc - loc: a prefix built by subtraction is not
  // guaranteed non-decreasing in floating point); the winner is the LAST chunk whose start is <= target, resolved with atomicMax so
  // that equal starts (empty chunks) cannot race
  const double left = __shfl_up(inc, 1, 64);
  s_d[tid] = off + ((tid & 63) ? left : 0.0);
  if (tid == 0) { s_d[1024] = target; s_i[0] = 0; }
  __syncthreads();
  if (s_d[tid] <= target) atomicMax(&s_i[0], tid);
  __syncthreads();
  const int tsel = s_i[0];
  if (tid == tsel) {
    const double target = s_d[1024];
    double run = s_d[tid];
    int pick = min(V, hi) - 1;
    for (int v = lo; v < hi; ++v) {
      bool rem = bf2f(row[v]) < thr;
      for (int r = 0; r < nrem; ++r) rem |= removed[r] == v;
      if (!rem) run += (double)expf(bf2f(row[v]) / T - m);
      if (run > target) { pick = v; break; }
    }
    s_i[1] = max(pick, 0);
  }
  __syncthreads();
  return s_i[1];
}

// first token: token = multinomial(softmax(lp(orig[:, -1])))   (utils.py:284-288)
__global__ __launch_bounds__(1024) void sample_row_kernel(const bf16_t* __restrict__ row, int V, float T, int top_k, unsigned long long seed,
                                                          int* out) {
  __shared__ float s_f[16];
  __shared__ double s_d[1025 + 17];
  __shared__ int s_i[2];
  __shared__ int s_hist[258];
  float m, Z;
  const float thr = vs_topk_threshold(row, V, top_k, s_hist);
  vs_row_stats(row, V, T, thr, s_f, m, Z);
  const int tok = vs_multinomial(row, V, T, m, thr, nullptr, 0, vs_uniform(seed, 0xFFFFu, 0u, 0u), s_d, s_i);
  if (threadIdx.x == 0) out[0] = tok;
}

// evaluate_posterior (sampling) + the integer half of update_inference_inputs; logits [T, V] bf16 of the verify forward
__device__ __forceinline__ void verify_accept_sample_body(TreeBufs tb, DevState* st, const bf16_t* __restrict__ logits, int V, float T,
                                                                    int top_k, unsigned long long seed, int* __restrict__ tokens, int tokens_cap,
                                                                    int* __restrict__ sel, int* __restrict__ accept_log, int log_cap,
                                                                    int* __restrict__ draft_ids, const float* __restrict__ u_over, int cohort) {
  // u_over (tests only, vispec_set_uniform_override_host): the uniforms of the rejection steps come from this table — [row j][level i] at
  // j * TREE_RET_W + i, the final multinomial's at TREE_MAX_T * TREE_RET_W — instead of the counter-based generator: how the reference's
  // recorded torch.rand_like draws (tests/golden g7) reach this kernel
  __shared__ int cand[TREE_MAX_T][TREE_RET_W];
  __shared__ int s_eq[TREE_MAX_T];
  __shared__ int accept_cand[TREE_RET_W];
  __shared__ int removed[TREE_MAX_T];
  __shared__ int s_seen[TREE_MAX_T];
  __shared__ int sh[8];  // 0 accept_length, 1 best, 2 adjust, 3 nrem, 4 fi, 5 accepted-this-level
  __shared__ float s_f[16];
  __shared__ double s_d[1025 + 17];
  __shared__ int s_i[2];
  __shared__ int s_hist[258];
  const int tid = threadIdx.x;
  if ((cohort && st->done) || (st->done & 4)) {  // finished request of a cohort (its partners are still running), or a full KV cache: freeze
    __syncthreads();
    if (tid == 0) st->frozen = 1;
    return;
  }
  const int nl = st->n_leaf, md = st->max_depth, round = st->rounds;
  if (tid < nl)
    for (int c = 0; c < TREE_RET_W; ++c) {
      const int node = tb.retrieve[tid * TREE_RET_W + c];
      cand[tid][c] = (c < md && node >= 0) ? tb.tree_tokens[node] : -1;
    }
  if (tid == 0) { sh[0] = 1; sh[1] = 0; sh[2] = 0; sh[3] = 0; }
  __syncthreads();
  if (tid == 0) accept_cand[0] = cand[0][0];
  __syncthreads();
  float last_m = 0.f, last_thr = NEG_INF;
  int last_node = 0;
  for (int i = 1; i < md; ++i) {
    const int al = sh[0];
    if (i != al) break;
    if (tid < nl) {
      bool eq = true;
      for (int c = 0; c < al; ++c) eq &= cand[tid][c] == accept_cand[c];
      s_eq[tid] = eq;
    }
    __syncthreads();
    if (tid == 0) {
      int fi = 0;
      while (fi < nl && !s_eq[fi]) ++fi;
      sh[4] = fi;
      sh[2] = 0;
      sh[3] = 0;
    }
    __syncthreads();
    const int node = tb.retrieve[sh[4] * TREE_RET_W + (i - 1)];
    const bf16_t* row = logits + (size_t)node * V;
    float m, Z;
    const float thr = vs_topk_threshold(row, V, top_k, s_hist);
    vs_row_stats(row, V, T, thr, s_f, m, Z);
    last_m = m;
    last_thr = thr;
    last_node = node;
    if (tid == 0) {
      float rm = 0.f;
      int nrem = 0, nseen = 0;
      int* seen = s_seen;  // (thread 0 only; a dynamically indexed local array is 256 B of scratch per lane for all 1024 threads)
      for (int j = 0; j < nl; ++j) {
        if (!s_eq[j]) continue;
        const int x = cand[j][i];
        if (x == -1) continue;
        bool dup = false;
        for (int q = 0; q < nseen; ++q) dup |= seen[q] == x;
        if (dup) continue;
        seen[nseen++] = x;
        const float p = bf2f(row[x]) >= thr ? expf(bf2f(row[x]) / T - m) / Z : 0.f;
        const float px = p / (1.0f - rm);
        if ((u_over ? u_over[j * TREE_RET_W + i] : vs_uniform(seed, (unsigned)round, (unsigned)j, (unsigned)i)) <= px) {
          accept_cand[al] = x;
          sh[0] = al + 1;
          sh[1] = j;
          break;
        }
        rm += p;
        removed[nrem++] = x;
        sh[2] = 1;
      }
      sh[3] = nrem;
    }
    __syncthreads();
  }
  const int accept_length = sh[0], best = sh[1];
  const bool use_gtp = sh[2] && accept_length != md;
  const int a = accept_length - 1;
  int node, nrem;
  float m, thr;
  if (use_gtp) {
    node = last_node; m = last_m; thr = last_thr; nrem = sh[3];
  } else {
    node = tb.retrieve[best * TREE_RET_W + a];
    float Z;
    thr = vs_topk_threshold(logits + (size_t)node * V, V, top_k, s_hist);
    vs_row_stats(logits + (size_t)node * V, V, T, thr, s_f, m, Z);
    nrem = 0;
  }
  const int next = vs_multinomial(logits + (size_t)node * V, V, T, m, thr, removed, nrem,
                                  u_over ? u_over[TREE_MAX_T * TREE_RET_W] : vs_uniform(seed, (unsigned)round, 255u, 255u), s_d, s_i);
  if (tid == 0) {
    const int* rowp = tb.retrieve + best * TREE_RET_W;
    const int n = st->n_ctx;
    int done = st->done;
    for (int jj = 0; jj <= a; ++jj) {
      const int nd = rowp[jj];
      sel[jj] = nd;
      const int tok = tb.tree_tokens[nd];
      if (n + jj < tokens_cap) tokens[n + jj] = tok;
      if (tok == st->eos_token_id || tok == st->stop2) done |= 1;
    }
    for (int jj = a + 1; jj < TREE_RET_W; ++jj) sel[jj] = rowp[a];
    for (int jj = 0; jj < TREE_RET_W; ++jj) draft_ids[jj] = jj < a ? tb.tree_tokens[rowp[jj + 1]] : next;
    st->n_prev = n;
    st->n_ctx = n + a + 1;
    st->accept_len = a;
    st->best = best;
    st->next_token = next;
    st->new_token += a + 1;
    if (st->new_token > st->max_new_tokens) done |= 2;
    if (st->n_ctx + TREE_MAX_T + KV_GUARD_ROWS > st->kv_cap) done |= 4;
    st->done = done;
    if (st->rounds < log_cap) accept_log[st->rounds] = a;
    st->rounds += 1;
  }
}
__global__ __launch_bounds__(1024) void verify_accept_sample_kernel(TreeBufs tb, DevState* st, const bf16_t* __restrict__ logits, int V, float T,
                                                                    int top_k, unsigned long long seed, int* __restrict__ tokens, int tokens_cap,
                                                                    int* __restrict__ sel, int* __restrict__ accept_log, int log_cap,
                                                                    int* __restrict__ draft_ids, const float* __restrict__ u_over, int cohort) { verify_accept_sample_body(tb, st, logits, V, T, top_k, seed, tokens, tokens_cap, sel, accept_log, log_cap, draft_ids, u_over, cohort); }
struct verify_accept_sample_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { verify_accept_sample_body(a...); }
};

// After the accept decision (one launch):
//   blocks [0, n_kv)   KV compaction (utils.py:529-538): rows n+sel[j] -> n+j for j=1..a, for every (layer, k|v, head); one wave per
//                      (slab, head): all sources are read into registers before anything is written (rows may overlap);
//   blocks n_kv + j    accept_hidden_state_new[j] = hidden_state_new[sel[j]] (utils.py:543-546), staged where the draft's catch-up forward
//                      reads it (dx1[j, 0:D]) together with the embedding of the id it is paired with (dx2[j, 0:D] = embed(draft_ids[j]),
//                      cnets_ours.py:1084,1093).
__device__ __forceinline__ void post_accept_body(bf16_t* __restrict__ kv, int s_max, int n_kv, const DevState* __restrict__ st,
                                                          const int* __restrict__ sel, const bf16_t* __restrict__ hidden_new,
                                                          bf16_t* __restrict__ accept_hidden, const int* __restrict__ draft_ids,
                                                          const bf16_t* __restrict__ draft_embed, bf16_t* __restrict__ dx1,
                                                          bf16_t* __restrict__ dx2, int D) {
  constexpr int HD = 128;
  if (st->frozen) return;
  if ((int)blockIdx.x < n_kv) {
    const int a = st->accept_len, n = st->n_prev;
    if (a == 0 || threadIdx.x >= 64) return;
    bf16_t* base = kv + ((size_t)blockIdx.x * s_max + n) * HD;  // blockIdx.x = slab*H_kv + head
    unsigned v[TREE_RET_W];
#pragma unroll
    for (int jj = 1; jj < TREE_RET_W; ++jj)
      if (jj <= a) v[jj] = *reinterpret_cast<const unsigned*>(base + (size_t)sel[jj] * HD + threadIdx.x * 2);
#pragma unroll
    for (int jj = 1; jj < TREE_RET_W; ++jj)
      if (jj <= a) *reinterpret_cast<unsigned*>(base + (size_t)jj * HD + threadIdx.x * 2) = v[jj];
    return;
  }
  const int j = blockIdx.x - n_kv;
  const bf16_t* hs = hidden_new + (size_t)sel[j] * D;
  const bf16_t* es = draft_embed ? draft_embed + (size_t)draft_ids[j] * D : nullptr;
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
    const uint4 h = *reinterpret_cast<const uint4*>(hs + d);
    *reinterpret_cast<uint4*>(accept_hidden + (size_t)j * D + d) = h;
    *reinterpret_cast<uint4*>(dx1 + (size_t)j * 2 * D + d) = h;
    if (es) *reinterpret_cast<uint4*>(dx2 + (size_t)j * 2 * D + d) = *reinterpret_cast<const uint4*>(es + d);
  }
}
__global__ __launch_bounds__(256) void post_accept_kernel(bf16_t* __restrict__ kv, int s_max, int n_kv, const DevState* __restrict__ st,
                                                          const int* __restrict__ sel, const bf16_t* __restrict__ hidden_new,
                                                          bf16_t* __restrict__ accept_hidden, const int* __restrict__ draft_ids,
                                                          const bf16_t* __restrict__ draft_embed, bf16_t* __restrict__ dx1,
                                                          bf16_t* __restrict__ dx2, int D) { post_accept_body(kv, s_max, n_kv, st, sel, hidden_new, accept_hidden, draft_ids, draft_embed, dx1, dx2, D); }
struct post_accept_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { post_accept_body(a...); }
};

// Draft-side bookkeeping after the catch-up forward: the a+1 new rows become part of stable_kv (cnets_ours.py:1108) and the last
// of them (out_hidden[:, -1], :1109) is what the tree grows from.
__device__ __forceinline__ void draft_advance_body(DevState* st, const bf16_t* __restrict__ dout, bf16_t* __restrict__ dlast, int D) {
  if (st->frozen) return;
  const int a = st->accept_len;
  for (int d = threadIdx.x * 8; d < D; d += 256 * 8)
    *reinterpret_cast<uint4*>(dlast + d) = *reinterpret_cast<const uint4*>(dout + (size_t)a * D + d);
  if (threadIdx.x == 0) {
    st->draft_len += a + 1;
    st->draft_real_len += a + 1;
    // the next round appends a catch-up (<= depth+2 rows) and top_k rows per tree level behind the stable KV
    if (st->draft_len + st->draft_round_rows + KV_GUARD_ROWS > st->draft_cap) st->done |= 4;
    // ... and rotates its rows at the real (uncompressed) positions draft_real_len .. + depth + 1: they must stay inside the tables
    if (st->draft_real_len + TREE_MAX_DEPTH + 2 + KV_GUARD_ROWS > st->draft_rope_rows) st->done |= 4;
  }
}
__global__ __launch_bounds__(256) void draft_advance_kernel(DevState* st, const bf16_t* __restrict__ dout, bf16_t* __restrict__ dlast, int D) { draft_advance_body(st, dout, dlast, D); }
struct draft_advance_fn {
  template <class... A> __device__ __forceinline__ void operator()(A... a) const { draft_advance_body(a...); }
};
