/*
 * libvispec_hip — C-ABI of the MI355X (gfx950) ViSpec draft-and-verify hot path.
 *
 * The reference (KangJialiang/ViSpec) is pure Python and has no FFI; the boundary it exposes for this path
 * is its Python object API.  This header is the C-ABI that sits *under* the Python mirror
 * of that API (vispec_amd/model/ *.py) — each entry point names the reference function(s) it replaces
 * (paths relative to /root/reference/vispec/model/).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; tensors are bf16 (uint16 storage) unless noted;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and never synchronises,
 *    except the *_host getters documented as blocking;
 *  - return value: 0 = ok, negative = error (vispec_last_error() gives the text);
 *  - no ownership transfer: the caller owns weights, KV caches and I/O buffers; the library owns only the
 *    opaque vispec_ctx and the workspace it allocates at create time;
 *  - one ctx per (process, device); calls on one ctx are not re-entrant; different ctxs are independent.
 */
#ifndef VISPEC_HIP_H
#define VISPEC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vispec_ctx vispec_ctx;

typedef struct {
  /* target language model (modeling_llama_kv.py) */
  int hidden_size, num_heads, num_kv_heads, head_dim, intermediate_size, vocab_size, num_layers;
  int max_pos;            /* KV-cache capacity per layer (kv_cache.py:109: config.max_position_embeddings) */
  float rms_eps;
  int qkv_bias;           /* Qwen-style q/k/v bias */
  /* draft (cnets_ours.py): one decoder layer, MHA, same hidden size */
  int draft_heads, draft_intermediate, draft_max_pos, draft_qkv_bias, draft_fc_bias;
  float draft_rms_eps;
  /* tree shape (cnets_ours.py:732-735) */
  int total_token, depth, top_k, num_q;
  int eos_token_id;
  int eager_scores;       /* 1: target attention rounds scores to bf16 like modeling_llama_kv.py:602-604 */
  int draft_rope_rows;    /* rows of the draft's rope_cos / rope_sin tables; 0 = draft_max_pos.  Draft rows are rotated at their
                             UNCOMPRESSED position (cnets_ours.py:845-868), which runs up to the target's context length, so the loader
                             should build max(max_pos, draft_max_pos) rows (the reference regrows its cache on demand, cnets_ours.py:157-162) */
} vispec_config;

/* per-layer target weights; wqkv = rows [q | k | v] fused, wgu = rows [gate | up] fused, all W32-packed (done once at load) */
typedef struct {
  const void *wqkv, *bqkv, *wo, *wgu, *wdown, *ln1, *ln2;
  /* fp8 (OCP e4m3) weights, BASELINE config 5: when a scale pointer is non-NULL the matching weight is an fp8 W32 image
     (vispec_pack_weight_fp8) and the pointer holds its per-output-channel fp32 dequantisation scales; NULL = bf16 weight */
  const void *sqkv, *so, *sgu, *sdown;
} vispec_layer_weights;

typedef struct {
  const void *embed;      /* [V, D]   model.embed_tokens.weight */
  const void *norm;       /* [D]      model.norm.weight */
  const void *lm_head;    /* [V, D]   lm_head.weight (also the draft's head, utils.py:300) */
  const void *lm_head_scale; /* fp32 [V] when lm_head is an fp8 image, else NULL */
  const void *rope_cos, *rope_sin; /* [max_pos, head_dim] bf16, built as modeling_llama_kv.py:147-181 */
} vispec_target_misc;

typedef struct {            /* SURVEY §8 A0 state-dict contract */
  const void *embed;        /* embed_tokens.weight [V, D] */
  const void *fc_w, *fc_b;  /* fc [D, 2D] (+[D]) */
  const void *imgfc_w, *imgfc_b;
  const void *wqkv, *bqkv, *wo, *wgu, *wdown, *ln2;   /* layers.0.* (no input norm, cnets_ours.py:537-540) */
  const void *ad_q;         /* imadpt.q [num_q, H, hd] */
  const void *ad_wkv, *ad_bkv; /* imadpt.{k,v}_proj fused [2D, D] */
  const void *ad_wo;        /* imadpt.o_proj [D, D] */
  const void *rope_cos, *rope_sin; /* [draft_rope_rows, head_dim] */
} vispec_draft_weights;

const char* vispec_last_error(void);
int  vispec_version(void);

int  vispec_ctx_create(const vispec_config* cfg, vispec_ctx** out);
/* Cohort member: another request context whose activation workspaces are one 32-row tile of `leader`'s 256-row workspaces (the first
   member owns tile 1, the second tile 2, ... the seventh tile 7; an eighth is refused), so that the cohort round functions below can launch
   every GEMM once for all requests (same config as the leader).  Lifetime: destroy the members before the leader; a leader destroyed
   while members are alive keeps its allocations until its last member is destroyed (the members stay usable as single requests) and its
   HANDLE IS INVALID from that call on — every entry point refuses it while a member is alive, and it must not be passed anywhere (not
   even to vispec_ctx_destroy again) once the last member is gone.
   A member is otherwise an ordinary ctx: own round state, tree, KV caches (vispec_set_kv), prefill calls.
   Tree size: with trees of up to 32 nodes a request owns ONE tile (eight requests per cohort); with 33..64 nodes (what the reference's
   total_token = -1 autotune picks, spec_model_ours.py:179-201) it owns TWO tiles of the target-side workspaces — rows 64 slot .. of the
   leader's — so only request slots 0..3 can hold such trees and a cohort round then serves at most four requests (2 requests: 4 tiles on
   the wide kernel, 3 / 4 requests: 6 / 8 tiles on the cohort-8 kernel).  All contexts of one round must have the same tree size. */
int  vispec_ctx_create_member(const vispec_config* cfg, vispec_ctx* leader, vispec_ctx** out);
void vispec_ctx_destroy(vispec_ctx* ctx);
int  vispec_set_target_layer(vispec_ctx*, int layer, const vispec_layer_weights*);
int  vispec_set_target_misc(vispec_ctx*, const vispec_target_misc*);
int  vispec_set_draft_weights(vispec_ctx*, const vispec_draft_weights*);
/* KV buffers owned by the caller: target [2*layers, 1, H_kv, max_pos, hd] (kv_cache.py:105-126),
   draft [2, H, draft_max_pos, hd] (replaces the torch.cat growth of cnets_ours.py:393-396) */
int  vispec_set_kv(vispec_ctx*, void* target_kv, void* draft_kv);

/* ---- single kernels (unit-testable building blocks) ------------------------------------------------ */
/* Every GEMM weight the library streams (all pointers in the three weight structs above except embed / norm vectors /
   ad_q / rope tables) is in the "W32" layout: 32-row x 16-k tiles stored as the 1 KiB A-operand image of
   v_mfma_f32_32x32x16_bf16, tiles of a row block contiguous along k (csrc/kernels.h).  vispec_pack_weight converts a
   row-major nn.Linear weight [N, K] once at load; P must hold vispec_packed_elems(N, K) bf16 elements.    A gate|up weight used with epilogue 2 (SwiGLU) must be handed over in "SwiGLU order": packed row 32t + c is natural gate row
   16t + c for c < 16 and natural up row I + 16t + (c - 16) otherwise (I = rows of one half, I % 16 == 0), so that the gate and up
   values of an output meet in one lane of the epilogue and the GEMM is I/16 equal single-tile workgroups. */
int vispec_pack_weight(vispec_ctx*, void* stream, const void* W_rowmajor, int N, int K, void* P);
long long vispec_packed_elems(int N, int K);
/* fp8 weights (W8A16): Wq is a row-major uint8 matrix of OCP e4m3fn codes; the GEMM computes bf16(scale[n]·(X·Wqᵀ)+bias) with bf16
   activations (the tile is up-converted in registers, exact) — the kernel is HBM-bound, the gain is the halved weight stream. */
int vispec_pack_weight_fp8(vispec_ctx*, void* stream, const void* Wq_rowmajor_u8, int N, int K, void* P8);
int vispec_gemm_skinny_fp8(vispec_ctx*, void* stream, const void* X, int ldx, const void* P8, const void* wscale_f32, const void* bias,
                           void* Y, int ldy, const void* R, int ldr, int M, int N, int K, int epilogue);
/* Y[M,N] = X[M,K] · W[N,K]^T (+bias) ; epilogue: 0 none, 1 += residual R (bf16 add of two bf16 tensors),
   2 SwiGLU: W holds [gate rows | up rows] (2N rows), Y = silu(g)*u.  M <= 32.   nn.Linear in every module above.
   W is W32-packed.  Small N is split over K across workgroups (needs ctx for the partial workspace). */
int vispec_gemm_skinny(vispec_ctx*, void* stream, const void* X, int ldx, const void* W, const void* bias,
                       void* Y, int ldy, const void* R, int ldr, int M, int N, int K, int epilogue);
/* same with the RMSNorm that follows fused: Y = bf16(R + bf16(X·W^T + b)) (R may be NULL), normed = norm_w * rms_norm(Y)
   (modeling_llama_kv.py: o_proj -> +residual -> post_attention_layernorm ; down_proj -> +residual -> next input_layernorm) */
int vispec_gemm_skinny_norm(vispec_ctx*, void* stream, const void* X, int ldx, const void* W, const void* bias, void* Y, int ldy,
                            const void* R, int ldr, const void* norm_w, void* normed, int ldn, float eps, int M, int N, int K);
/* W8A8 (vispec_set_fp8_activations) at unit level (tests): Y = bf16((q_x . q_w^T) * wscale[n] * sx[m] + b) [+ epilogue], X bf16 quantised per
   row inside (sx = max|x| / 448, q = e4m3(x / sx)).  n_req = 1: M <= 64 rows; n_req = 2..8: vispec_gemm_cohort's row layout.  norm_w != NULL:
   + residual and the fused RMSNorm of the split-K reduce (o_proj / down_proj form), written to `normed` (ld N).  No reference counterpart. */
int vispec_gemm_fp8a8(vispec_ctx*, void* stream, const void* X, int ldx, const void* P8, const void* wscale_f32, const void* bias, void* Y, int ldy,
                      const void* R, int ldr, int n_req, int m_tile, int M, int N, int K, int epilogue, const void* norm_w, void* normed, float eps);
/* Row-wise e4m3 quantisation of bf16 activations X[M, K] (W8A8: sx[m] = max|x[m, :]| / 448, Q = e4m3(x / sx), uint8 codes, fp32 scales):
   the PyTorch prefill of an fp8a8 model quantises its q|k|v, gate|up and down inputs with it and multiplies on the library's fp8 x fp8 GEMM
   (torch._scaled_mm with row-wise scales).  No reference counterpart.  ctx may be NULL. */
int vispec_quant_rows_e4m3(vispec_ctx*, void* stream, const void* X, int ldx, void* Q_u8, int ldq, void* sx_f32, int M, int K);
/* test hook: the W8A8 scratch of the ctx as the last quantisation left it — e4m3 codes [rows][K] (uint8) and per-row scales (fp32), device to
   device on `stream`.  After vispec_gemm_fp8a8 with norm_w: the quantised `normed` rows, written by the split-K reduce itself (the fused form of
   the quantisation pass that target_forward uses for the q|k|v and gate|up inputs). */
int vispec_a8_scratch_read(vispec_ctx*, void* stream, void* codes_out, void* scales_out, int rows, int K);
/* The GEMM of a cohort round at unit level (tests): n_req = 2..8 requests of m_tile <= 32 rows each (2..4: rows bit-identical to the
   single-request kernel's; 5..8: csrc/gemm_c8.h's kernel — one accumulator chain per output element, rows independent of what shares the
   launch but not bit-identical to the single-request order; m_tile < 0, the slab form, keeps the single-request order at every n_req); request t's rows are rows
   32t .. 32t + m_tile - 1 of X / Y / R (which therefore span 32 n_req rows; rows beyond m_tile of a tile are neither read for results
   nor written).  Same epilogues as vispec_gemm_skinny (0 none, 1 +R, 2 SwiGLU).  Row for row bit-identical to vispec_gemm_skinny on the
   request's own rows.  m_tile in [-8, -1] = the draft's slab form: the requests have -m_tile <= 8 live rows each (top_k rows of a tree
   level, the catch-up rows, the root row: cnets_ours.py:1090-1165) and share ONE activation tile of the launch — same row placement in
   X / Y / R, same results, the weight pass at a single request's cost. */
int vispec_gemm_cohort(vispec_ctx*, void* stream, const void* X, int ldx, const void* W, const void* wscale /* fp8 image: per-row scales, else NULL */,
                       const void* bias, void* Y, int ldy, const void* R, int ldr, int n_req, int m_tile, int N, int K, int epilogue);
/* tuning hook for tools/gemm_bench.py: explicit decomposition, variant = S*100 + {0:4,1:8 waves}*10 + {0:4,1:8,2:16 unroll} */
int vispec_gemm_skinny_tune(vispec_ctx*, int variant, void* stream, const void* X, int ldx, const void* W, void* Y, int ldy,
                            int M, int N, int K);
/* LlamaRMSNorm (cnets_ours.py:513-527, modeling_llama_kv.py:104-133) */
int vispec_rmsnorm(vispec_ctx*, void* stream, const void* X, const void* w, void* Y, int M, int D, float eps);
/* Residual add + the norm that follows it in one pass (the prefill side of LlamaDecoderLayer, modeling_llama_kv.py:742-756):
   X <- bf16(X + R) in place, Y = RMSNorm(X) * w.  any M */
int vispec_add_rmsnorm(vispec_ctx*, void* stream, void* X, const void* R, const void* w, void* Y, int M, int D, float eps);
/* SwiGLU activation of a gate|up block [M, 2I] (row stride ld): out[M, I] = bf16(bf16(silu(gate)) * up) — the prefill side of
   LlamaMLP (modeling_llama_kv.py:240-262), where the projections themselves are library GEMMs; any M */
int vispec_silu_mul(vispec_ctx*, void* stream, const void* gate_up, int ld, void* out, int ldo, int M, int I);
/* Epilogue of a PREFILL GEMM on fp8 (e4m3) weights whose accumulation a library GEMM did in fp32 (bf16 activations x codes):
   out[m, n] = bf16(acc[m, n] * scale[n] (+ bias[n])) — the W8A16 rounding points of the decode GEMMs (no reference counterpart: the reference has
   no fp8 path; BASELINE config 5) — one pass instead of three torch element-wise passes over the fp32 tensor. */
int vispec_scale_bias_cast(vispec_ctx*, void* stream, const void* acc_f32, int ld, const void* scale_f32, const void* bias_bf16, void* out_bf16,
                           int ldo, int M, int N);
/* Causal self-attention of a prompt's L rows — the PREFILL side of LlamaAttention / Qwen2_5_VLSdpaAttention (modeling_llama_kv.py:595-640:
   scores bf16, fp32 softmax, eager_scores = 1; modeling_qwen2_5_vl_kv.py:1073-1170: SDPA, eager_scores = 0) without the [H, L, L]
   score tensor: q rows [L, ldq] (head h at column 128 h, rotary already applied by vispec_rope_append), K/V rows [0, L) of one layer's
   cache slabs [H_kv, s_max, 128] as that call wrote them, out [L, ldo] bf16 (head h at column 128 h).  head_dim 128; any L <= s_max. */
int vispec_prefill_attention(vispec_ctx*, void* stream, const void* q, int ldq, const void* k_cache, const void* v_cache, int s_max,
                             int H, int H_kv, int L, void* out, int ldo, int eager_scores);
/* rotary (cnets_ours.py:104-119) on fused qkv rows + append K,V to a [H_kv, S_max, hd] cache at rows
   *kv_base_dev + i ; positions = *pos_base_dev + pos_off_dev[i] (pos_off_dev may be NULL = i). Q is rotated in place. */
int vispec_rope_append(vispec_ctx*, void* stream, void* qkv, int M, int H, int H_kv, int hd, const void* cos, const void* sin,
                       const int* pos_base_dev, const int* pos_off_dev, void* k_cache, void* v_cache, int s_max,
                       const int* kv_base_dev);
/* q|k|v projection + rotary + KV append in one pass (modeling_llama_kv.py:560-594; cnets_ours.py:362-396): qkv[:, :H*hd] receives
   the rotated q, k (rotated) and v go straight to cache rows *kv_base_dev + i.  W is the fused [ (H+2H_kv)*hd, K ] weight packed
   by vispec_pack_weight(_fp8); when vispec_qkv_rope_fused((H+2H_kv)*hd) is 1 the loader must first reorder its q and k rows into
   rope order — within each head, row 32t + c takes natural row 16t + c (c < 16) or 64 + 16t + (c - 16) — so that a rotate_half
   pair meets in one lane of the GEMM epilogue (the launch is then ONE kernel); when it is 0 the natural order is kept and the
   library runs the split-K GEMM followed by the rotary/append kernel.  wscale NULL = bf16 weight, else fp8 + per-row scales
   (natural order).  Positions as vispec_rope_append. */
int vispec_qkv_rope_fused(int n_qkv_rows);
int vispec_gemm_qkv_rope(vispec_ctx*, void* stream, const void* X, int ldx, const void* W, const void* wscale, const void* bias,
                         void* qkv, int M, int H, int H_kv, int hd, int K, const void* cos, const void* sin,
                         const int* pos_base_dev, const int* pos_off_dev, void* k_cache, void* v_cache, int s_max,
                         const int* kv_base_dev);
/* tree-masked attention of M query rows against cache rows [0, *prefix_dev) (all visible) + `tail` rows after
   them, of which row m sees tail key t iff bit t of mask_dev[m] is set  (cnets_ours.py:781-815 + 428-433;
   modeling_llama_kv.py:890-924 + 602-623).  q [M, H*hd] row stride ldq ; out [M, H*hd]. */
int vispec_tree_attention(vispec_ctx*, void* stream, const void* q, int ldq, const void* k_cache, const void* v_cache,
                          int s_max, int H, int H_kv, int hd, int M, const int* prefix_dev, int tail,
                          const uint64_t* mask_dev, void* out, int ldo, int eager_scores);
/* row-wise argmax over bf16 logits (first max wins) -> int32 [M]   (utils.py:290,441,554) */
int vispec_argmax_rows(vispec_ctx*, void* stream, const void* logits, int ld, int M, int V, int* out_idx);
/* row-wise log-softmax (bf16 out) + top-k, value desc / index asc (cnets_ours.py:1113-1115,1146-1149) */
int vispec_logsoftmax_topk(vispec_ctx*, void* stream, const void* logits, int ld, int M, int V, int k,
                           int* out_idx, float* out_logp);

/* ---- the path ------------------------------------------------------------------------------------- */
/* Request start: context length L already prefetched into the target KV (by the caller's prefill), first sampled
   token; resets the round state.  (spec_model_ours.py:283-307,476) */
int vispec_begin_request(vispec_ctx*, void* stream, const int* prompt_ids_host, int L, int max_new_tokens);

/* Draft prefill with fused vision adaptor (cnets_ours.py:879-975 + 1099-1123) followed by the tree growth of
   topK_genrate (cnets_ours.py:1126-1238).  hidden/embeds [L, D] (embeds = the target's inputs_embeds, un-shifted),
   image_mask_host [L] (NULL = LLaVA-1.5 / text semantics), first_token_dev = argmax of the prefill's last logits row. */
int vispec_draft_prefill(vispec_ctx*, void* stream, const void* hidden, const void* embeds, const uint8_t* image_mask_host,
                         int L, const int* first_token_dev);

/* Target verify forward of the current tree (utils.py:389-412 tree_decoding -> SpecModel.forward ->
   modeling_llama_kv.py:927-1080) + greedy accept (utils.py:415-451) + KV compaction / bookkeeping
   (utils.py:496-556).  n_tokens = 0 means "the current tree"; forced_accept >= 0 overrides the measured accept
   length (bench-only scripted-acceptance mode, never used by the parity tests). */
int vispec_verify_accept(vispec_ctx*, void* stream, int forced_accept);

/* The two halves of vispec_verify_accept, for the API mirror (utils.tree_decoding / evaluate_posterior +
   update_inference_inputs are separate calls in the reference): forward leaves logits [T,V], hidden_state_new [T,D]
   and the per-node argmax in ctx buffers; accept consumes them. */
int vispec_target_forward(vispec_ctx*, void* stream);
int vispec_accept(vispec_ctx*, void* stream, int forced_accept);
/* Install a caller-built tree (host arrays; retrieve is [n_leaf, max_depth] row-major, -1 padded) instead of the one the
   draft produced — what utils.tree_decoding does with arbitrary tree_candidates / tree_mask.  Blocking. */
int vispec_set_tree_host(vispec_ctx*, void* stream, const int* tokens_T, const int* pos_T, const uint64_t* mask_T,
                         const int* retrieve, int n_leaf, int max_depth);
/* The retrieve table alone (`retrieve` may be NULL in vispec_set_tree_host): SpecModel.forward's verify form
   (spec_model_ours.py:226-245 called by utils.tree_decoding, utils.py:404-409) receives tokens / positions / mask only; the
   table reaches the library when evaluate_posterior / update_inference_inputs are given it (utils.py:415,496).  Blocking. */
int vispec_set_retrieve_host(vispec_ctx*, void* stream, const int* retrieve, int n_leaf, int max_depth);

/* One decode call of topK_genrate (cnets_ours.py:1043-1238, stable_kv branch) on the hidden states accepted by the
   last vispec_verify_accept. */
int vispec_draft_round(vispec_ctx*, void* stream);

/* The same two round functions for a COHORT of two independent requests (leader + member ctx) running their rounds in lockstep:
   a round is bound by the HBM stream of the weights, so every GEMM is launched once on 64 activation rows (tile t = request t) and the
   weights are read once for both; trees, accept decisions, KV caches, attention and round state stay per request.  Each request keeps
   the reference's batch-1 semantics (spec_model_ours.py:247-582 per request) and produces the tokens it would produce alone, bit for
   bit.  A request that has finished (done != 0) is frozen on the device while its partner completes. */
int vispec_cohort_verify_accept(vispec_ctx* leader, vispec_ctx* member, void* stream, int forced_accept);
int vispec_cohort_draft_round(vispec_ctx* leader, vispec_ctx* member, void* stream);
/* The same for n = 2..8 requests: ctxs[0] = the leader, the others its members owning activation tiles 1 .. n-1 (any order).  Five to
   eight requests (round 5): the target's GEMMs run on csrc/gemm_c8.h's kernel (8 weight row blocks x 8 request tiles per workgroup, ONE
   accumulator chain per output element: a request's tokens do not depend on what shares its weight pass, its logits differ from the
   single-request kernel's in fp32 summation order only — tests/test_c8_gpu.py, tests/test_full_size_gpu.py), the draft's GEMMs in the two-tile
   slab form (the single-request summation order, bit for bit), attention / per-request kernels as one launch for all eight; BOTH attentions
   of such a cohort — the target's tree attention and the draft's — split the keys 768 per workgroup instead of 512, so the draft path as a
   whole is composition-independent but, like the target path, not bit-identical to a solo request's.  With
   three or four requests the GEMMs run on csrc/gemm_wide.h's kernel (16 waves = 4 weight row blocks x 4 K-quarters sharing each staged
   activation group; per-row arithmetic identical to the single-request kernel), attention takes all requests in one partial + one
   reduce launch.  Guarantee for n <= 4: every request's tokens are those of the same request alone, bit for bit; for n = 5..8: every
   request's tokens are those it produces in ANY cohort of 5..8 (tile, neighbours, live-row count do not matter). */
int vispec_cohortn_verify_accept(vispec_ctx* const* ctxs, int n, void* stream, int forced_accept);
int vispec_cohortn_draft_round(vispec_ctx* const* ctxs, int n, void* stream);

/* second stop token of the current request (`is_llama3`: "<|eot_id|>", spec_model_ours.py:268-269,540-542); call after
   vispec_begin_request, which clears it; -1 = none */
int vispec_set_stop_token(vispec_ctx*, void* stream, int token_id);
/* change the tree size (nodes incl. the root, 1..64) of later rounds: what `model.spec_layer.total_tokens = total_token - 1`
   does after the total_token=-1 autotune of SpecModel.from_pretrained (spec_model_ours.py:179-201).  Trees of more than 32 nodes
   run every verify GEMM on two 32-row activation tiles; a cohort member then needs request slot 1..3 (see vispec_ctx_create_member),
   and its target-side buffer views (vispec_buffer: logits, hidden_new ...) move to rows 64 slot .. of the leader's workspaces. */
int vispec_set_total_token(vispec_ctx*, int total_token);
/* Qwen2.5-VL: the rope_deltas cached by the prefill (modeling_qwen2_5_vl_kv.py) shift every decode position: tree verify and AR
   steps rotate at n + tree_pos + delta (utils.py:397-402; the three M-RoPE components are equal there, i.e. ordinary 1-D rotary).
   Call after vispec_begin_request (which resets it to 0). */
int vispec_set_rope_delta(vispec_ctx*, void* stream, int delta);
/* TopKLogitsWarper (utils.py:52-53) applied after the temperature on every sampled / verified distribution; 0 = off.  Call after
   vispec_set_sampling, which resets it.  (TopPLogitsWarper is not offered: HF's implementation raises on the 3-D tree logits the
   reference hands it in evaluate_posterior, so the reference cannot run it either.) */
int vispec_set_top_k(vispec_ctx*, int top_k);
/* Sampling (temperature > 0, utils.py:453-493 + multinomial at :288,:551): enable with a temperature and a seed; verify_accept then
   runs the sequential-rejection accept and draws the next token on the device with counter-based uniforms (distributionally
   equivalent to the reference's torch RNG, bit-reproducible against the oracle).  temperature <= 1e-5 restores greedy. */
int vispec_set_sampling(vispec_ctx*, float temperature, unsigned long long seed);
int vispec_sample_row(vispec_ctx*, void* stream, const void* logits_row_bf16, int V, int* out_token_dev);
/* Tests only: take the uniforms of the sampling accept — one per (candidate row, level): the reference's torch.rand_like (utils.py:461),
   and one for the final multinomial (utils.py:488-493 / :551) — from a host table [n_leaf, max_depth] instead of the counter-based generator,
   so that the reference's RECORDED draws (tests/golden g7, g11) can be replayed through vispec_accept.  u = NULL: off.  Blocking. */
int vispec_set_uniform_override_host(vispec_ctx*, void* stream, const float* u, int n_leaf, int max_depth, float u_final);
/* Seed the next token to decode (the prefill's argmax) when no draft is used (AR baseline). */
int vispec_set_next_token(vispec_ctx*, void* stream, const int* token_dev);
/* Plain autoregressive step of the target with the same kernels (gen_baseline_answer_coco_caption.py:111-129). */
int vispec_ar_step(vispec_ctx*, void* stream);
/* The same for the n = 2..8 requests of a cohort (ctxs as for vispec_cohortn_verify_accept) on ONE weight pass: the AR baseline at the
   batching of the speculative run it is compared with (speed.py:56-97 divides like by like).  Row for row vispec_ar_step's arithmetic:
   a request's AR tokens do not depend on its cohort; a finished request (done != 0) freezes while the others go on. */
int vispec_cohortn_ar_step(vispec_ctx* const* ctxs, int n, void* stream);

/* Blocking read-back of the round state: out[0]=n_ctx, [1]=new_token, [2]=rounds, [3]=done(eos|max_new), [4]=last accept_len,
   [5]=next_token, [6]=draft kv len, [7]=n_leaf of current tree. */
int vispec_get_state_host(vispec_ctx*, void* stream, int* out8_host);
/* ... of every request of a cohort (leader first) with one synchronisation: out = n x 8 ints */
int vispec_cohort_get_state_host(vispec_ctx* const* ctxs, int n, void* stream, int* out8n_host);
/* vispec_cohort_get_state_host in two halves for a host loop with one round of lookahead (no reference counterpart: the reference's loop
   synchronises ~10 times per round, spec_model_ours.py:478-582): `enqueue` snapshots every request's state into pinned slot 0 / 1 in stream
   order and records an event — the next round is launched right behind it; `wait` blocks on that event only and unpacks n x 8 ints as above.
   A request that finished in the snapshot's round is frozen on the device, so the round already in flight leaves it untouched. */
int vispec_cohort_state_enqueue(vispec_ctx* const* ctxs, int n, void* stream, int slot);
int vispec_cohort_state_wait(vispec_ctx* const* ctxs, int n, int slot, int* out8n_host);
/* (best_candidate, accept_length) of the last accept — what evaluate_posterior returns (utils.py:451,493); blocking */
int vispec_get_last_accept_host(vispec_ctx*, void* stream, int* out2_host);
/* Blocking copies of device-side logs/buffers, for the API mirror and the tests. */
int vispec_get_tokens_host(vispec_ctx*, void* stream, int* out_host, int n);             /* committed token ids */
int vispec_get_accept_log_host(vispec_ctx*, void* stream, int* out_host, int n_rounds);
int vispec_get_tree_host(vispec_ctx*, void* stream, int* tokens_T, int* pos_T, uint64_t* mask_T, int* retrieve_Txd2,
                         int* n_leaf, int* max_depth);
/* hipGraph replay of the round functions (verify_accept / draft_round / ar_step) on capturable (non-null) streams: on by default. */
int vispec_set_graphs(vispec_ctx*, int on);
/* BASELINE config 5 ("fp8 weights (CDNA4 fp8 MFMA)"; the reference has no fp8 path): with fp8 target weights, also take the ACTIVATIONS of the
   target's q|k|v, gate|up and down GEMMs (modeling_qwen2_5_vl_kv.py:1065-1170; o_proj keeps bf16 activations) in e4m3 — one dynamic scale per row —
   and multiply on v_mfma_scale_f32_32x32x64_f8f6f4 (W8A8) instead of up-converting the weight codes for the bf16 MFMA (W8A16, the default).
   A different arithmetic ("same accepted tokens or documented divergence"); lm_head and draft unchanged.  This switch
   covers the library's verify / AR forwards; the Python prefill of a target_weight_dtype="fp8a8" model (vispec_amd/model/target.py) quantises
   the same three GEMM inputs with vispec_quant_rows_e4m3 and runs them on the library's fp8 x fp8 GEMM (torch._scaled_mm) as well. */
int vispec_set_fp8_activations(vispec_ctx*, int on);
int vispec_graph_stats(vispec_ctx*, long long* out3);  /* {replays, captures, direct runs} */
/* Launch shape of the GEMMs of a three- or four-request cohort round (five to eight requests always run gemm_c8.h's one shape) (no reference counterpart: the reference is batch-1 only,
   spec_model_ours.py:247-582): weight row blocks per workgroup — 4 (default: one byte of activations per weight byte; for a GPU that
   several request lanes keep busy), 3 or 2 (more, smaller workgroups), 0 (the smallest of {2, 3, 4} whose grid still runs in one round
   of CUs: a single lane), 8 (round 4: eight row blocks per workgroup, the K range walked quarter by quarter by every wave — half the
   activation traffic per weight byte, half the workgroups: gemm_w32_wide8_kernel), 84 (eight for bf16 weights and for W8A8, four for fp8 weights with bf16 activations: what bench.py runs
   with several lanes per GPU).
   Results are bit-identical in every setting. */
int vispec_set_wide_row_blocks(vispec_ctx* leader, int row_blocks);
/* In-library profiling used by bench.py's roofline object: when on, every skinny-GEMM / attention launch is bracketed by
   HIP events on its own stream.  kinds 0..4 = skinny GEMM {none, residual, swiglu, split-K partial, split-K reduce(+norm)}, 9 = attention
   partial, 10 = attention reduce.  report: out[kind*4+{0,1,2,3}] = {launches, total ms, total algorithmic bytes, total workgroups}. Blocking. */
int vispec_prof_enable(vispec_ctx*, int on);
int vispec_prof_report_host(vispec_ctx*, void* stream, double* out, int n_kinds);
/* device pointers of internal buffers (hidden_state_new [T,D], verify logits [T,V] bf16, draft last hidden ...) */
void* vispec_buffer(vispec_ctx*, const char* name);

#ifdef __cplusplus
}
#endif
#endif
