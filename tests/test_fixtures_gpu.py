"""The reference's STAGE fixtures (tests/golden g4, g6, g7, g11, g13, g14 — captured from /root/reference by tests/golden/gen_golden.py) fed to
the HIP kernels DIRECTLY, through the C-ABI: until round 4 those fixtures pinned the oracle (tests/test_oracle_golden.py) and the kernels
met the oracle on other inputs; here the kernels meet the reference's own inputs and outputs — its edge cases included (ties, zero accept,
-1 padded paths, recorded torch.rand_like draws, top-k warper, KV gather-compaction of every cache tensor).

Bars, written where they are used: integer outputs (accept decisions, token ids, tree tables, KV rows — pure copies) exact; floats within 2^-6 of
the tensor's largest magnitude (the fixtures are the reference's fp32 CPU run, the kernels compute in bf16 like the reference on a GPU).  Where
a fixture's decision hangs on an fp32 quantity closer to its threshold than bf16 can resolve, the case is checked against the oracle evaluated
on the very bf16 numbers the kernel saw, and the test says how many cases that was."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402

from test_kernels_gpu import assert_bf16_close, dev, lib, p, packed, stream, tb  # noqa: E402,F401
from test_loop_gpu import build  # noqa: E402

PAD = -30000.0  # logit of the vocabulary columns a (narrower) fixture does not have: exp() == 0, never an arg-max


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def path_tree(cand):
    """A fixture gives `candidates` [n_leaf, m] (= tree tokens gathered along each root-to-leaf path, -1 padded: utils.py:412) and logits
    gathered the same way.  The equivalent tree for vispec_set_tree_host: one node per (path, depth) — node r*m + j — so that
    tree_logits[retrieve] is the fixture's tensor again.  -> tokens [T], pos [T], ancestor mask bits [T], retrieve [n_leaf, m]."""
    nl, m = cand.shape
    tokens = np.where(cand >= 0, cand, 0).reshape(-1).astype(np.int32)
    retrieve = np.arange(nl * m, dtype=np.int32).reshape(nl, m)
    retrieve[cand < 0] = -1
    pos = np.tile(np.arange(m, dtype=np.int32), nl)
    bits = np.array([sum(1 << (r * m + jj) for jj in range(j + 1)) for r in range(nl) for j in range(m)], np.uint64)
    return tokens, pos, bits, retrieve


def install(eng, cand, logits_rows, prompt):
    """New request + the fixture's tree + its logits in the ctx's verify-logits buffer (bf16 [T, V]); -> the T x V view."""
    nl, m = cand.shape
    Tn, V = nl * m, T["V"]
    eng.set_total_token(Tn)
    eng.begin_request(prompt, 200)
    eng.set_tree(*path_tree(cand))
    lb = eng.buffer("logits", (64, V))
    rows = np.full((Tn, V), PAD, np.float32)
    rows[:, : logits_rows.shape[-1]] = logits_rows.reshape(Tn, -1)
    lb[:Tn] = torch.from_numpy(rows).to(torch.bfloat16).to(lb.device)
    return lb[:Tn]


def dense_ranks(x):
    """Rank transform along the last axis (ties keep equal ranks): small integers — exact in bf16 — with the arg-max structure of x."""
    out = np.empty_like(x, dtype=np.float32)
    flat, o = x.reshape(-1, x.shape[-1]), out.reshape(-1, x.shape[-1])
    for i, row in enumerate(flat):
        _, inv = np.unique(row, return_inverse=True)
        o[i] = inv
    return out


def test_g6_greedy_evaluate_posterior_fixture_through_vispec_accept(lib, golden_dir):
    """evaluate_posterior, greedy branch (utils.py:438-451) — the reference's cases incl. ties between paths (first max wins), zero accept and
    -1 padded paths — through vispec_set_tree_host + logits upload + vispec_argmax_rows + vispec_accept: (best_candidate, accept_length), the
    accepted ids appended to the request, and the next token = arg-max of the returned distribution, all exact."""
    g = load(golden_dir, "g6_posterior.npz")
    sm, _, _ = build(50, 60, True)
    eng = sm.engine
    prompt = np.arange(3, 15, dtype=np.int32)
    seen = set()
    for i in range(int(g["n"])):
        lg, cand = g[f"logits{i}"], g[f"cand{i}"]
        nl, m, _ = lg.shape
        rows = install(eng, cand, dense_ranks(lg), prompt)  # greedy decisions depend on the arg-max only: ranks keep it, ties included, exactly
        am = eng.buffer("am", (64,), torch.int32)
        L.check(lib.vispec_argmax_rows(eng.h, stream(), p(rows), T["V"], nl * m, T["V"], p(am)))
        eng.accept()
        best, acc = eng.last_accept()
        st = eng.state()
        assert (best, acc) == (int(g[f"best{i}"]), int(g[f"acc{i}"])), f"case {i}"
        assert st["n_ctx"] == len(prompt) + acc + 1 and st["new_token"] == acc + 1
        np.testing.assert_array_equal(eng.tokens(st["n_ctx"])[len(prompt):], cand[best, : acc + 1], err_msg=f"case {i}: accepted ids")
        assert st["next_token"] == int(vo.argmax_first(g[f"p{i}"])), f"case {i}: next token"  # utils.py:554 on the returned sample_p
        seen.add(acc)
    assert 0 in seen and max(seen) >= 2


def test_g7_sampling_evaluate_posterior_fixture_with_the_references_recorded_uniforms(lib, golden_dir):
    """evaluate_posterior, sampling branch (utils.py:453-493) with the processor list of --temperature T [+ TopKLogitsWarper]: the reference's
    RECORDED torch.rand_like draws are replayed through verify_accept_sample_kernel (vispec_set_uniform_override_host).  The kernel sees the
    fixture's logits rounded to bf16 (verify logits are bf16 tensors); a case whose accept decision survives that rounding must reproduce the
    fixture's (best, accept_length) exactly — and every case must reproduce the oracle evaluated on the rounded logits; the drawn next token
    must carry probability mass in the returned (residual, renormalised) distribution and be the inverse-CDF draw of the override uniform."""
    g = load(golden_dir, "g7_posterior_sampling.npz")
    sm, _, _ = build(50, 60, True)
    eng = sm.engine
    prompt = np.arange(3, 15, dtype=np.int32)
    same_as_fixture, accs = 0, []
    n = int(g["n"])
    for i in range(n):
        lg, cand, u = g[f"logits{i}"], g[f"cand{i}"], np.ascontiguousarray(g[f"u{i}"], np.float32)
        Tq, K = float(g[f"T{i}"]), int(g[f"K{i}"])
        nl, m, _ = lg.shape
        lg16 = synth.bf16_grid(lg)
        install(eng, cand, lg16, prompt)
        eng.set_sampling(Tq, seed=1, top_k=K)
        u_final = 0.37 + 0.01 * (i % 7)
        L.check(lib.vispec_set_uniform_override_host(eng.h, stream(), u.ctypes.data_as(C.c_void_p), nl, m, C.c_float(u_final)))
        eng.accept()
        best, acc = eng.last_accept()
        st = eng.state()
        ob, oa, op_ = vo.evaluate_posterior_sampling(lg16, cand, Tq, lambda j, c: u[j, c], top_k=K)
        assert (best, acc) == (ob, oa), f"case {i}: kernel {(best, acc)} vs the oracle on the same bf16 logits {(ob, oa)}"
        if (ob, oa) == (int(g[f"best{i}"]), int(g[f"acc{i}"])):
            same_as_fixture += 1
        np.testing.assert_array_equal(eng.tokens(st["n_ctx"])[len(prompt):], cand[best, : acc + 1], err_msg=f"case {i}: accepted ids")
        nxt = st["next_token"]
        assert nxt < lg.shape[-1] and op_[nxt] > 0 and g[f"p{i}"][nxt] > 0, f"case {i}: next token {nxt} has no mass in the returned distribution"
        want = vo.multinomial_inverse_cdf(op_, u_final)
        if nxt != want:  # only a draw that sits on a CDF step (fp32 exp / sums in another order) may differ
            cdf = np.cumsum(op_.astype(np.float64))
            assert np.abs(cdf - u_final * cdf[-1]).min() < 1e-4, f"case {i}: next token {nxt}, oracle draw {want}"
        if K:
            assert (op_ > 0).sum() <= K
        accs.append(acc)
    L.check(lib.vispec_set_uniform_override_host(eng.h, stream(), None, 0, 0, C.c_float(0)))
    eng.set_sampling(0.0, 0)
    assert n == 36 and same_as_fixture >= n - 3, f"only {same_as_fixture} of {n} fixture decisions survive the bf16 rounding of the logits"
    assert 0 in accs and max(accs) >= 2
    print(f"g7 through the HIP accept: {same_as_fixture}/{n} cases equal to the reference's own (best, accept_length); all equal to the oracle on bf16 logits")


def test_g11_update_inference_inputs_fixture_through_post_accept(lib, golden_dir):
    """utils.update_inference_inputs in isolation (utils.py:496-593) on the reference's own tensors: accepted ids appended, KV rows gathered from
    the tree slots into [n, n + a + 1) of EVERY cache tensor (both of the fixture's device tensors), length, the hidden rows and ids staged for
    the draft, next token (arg-max, and the multinomial draw with the recorded uniform), new_token.  The fixture's (best, accept_length) is
    imposed through the per-node arg-max / the uniform table; its 4-wide head_dim and 8-wide hidden rows sit in the first columns of the
    engine's 128 / 256-wide rows (the kernels copy whole rows)."""
    g = load(golden_dir, "g11_update.npz")
    sm, _, _ = build(50, 60, True)
    eng = sm.engine
    V, D = T["V"], T["D"]
    for i in range(int(g["n"])):
        ids, ri, cand = g[f"ids{i}"], g[f"ri{i}"], g[f"cand{i}"]
        best, acc, sampling = int(g[f"best{i}"]), int(g[f"acc{i}"]), bool(g[f"sampling{i}"])
        n, Tn = len(ids), int(ri.max()) + 1
        hid, sp = g[f"hid{i}"][0], g[f"sp{i}"]
        tree_tok = np.zeros(Tn, np.int64)
        tree_tok[ri[ri >= 0]] = cand[ri >= 0]
        path = ri[best, : acc + 1]
        for pass_, (data, o_data) in enumerate(((g[f"data{i}"], g[f"o_data{i}"]), (g[f"data2_{i}"], g[f"o_data2_{i}"]))):
            eng.set_total_token(Tn)
            eng.begin_request(ids.astype(np.int32), 200)
            depth = np.zeros(Tn, np.int32)
            bits = np.zeros(Tn, np.uint64)
            for row in ri:
                for j, node in enumerate(row):
                    if node >= 0:
                        depth[node] = j
                        bits[node] = np.uint64(sum(1 << int(a) for a in row[: j + 1]))
            eng.set_tree(tree_tok.astype(np.int32), depth, bits, ri.astype(np.int32))
            # the cache tensors: fixture [slabs, 1, heads, 32, 4] -> the first rows / columns of [2 * layers, 1, H_kv, max_pos, 128]
            kv = eng.target_kv
            kv.zero_()
            ns = data.shape[0]
            kv[:ns, :, :, :32, :4] = torch.from_numpy(data).to(torch.bfloat16).to(kv.device)
            hn = eng.buffer("hidden_new", (64, D))
            hn[:Tn].zero_()
            hn[:Tn, :8] = torch.from_numpy(hid).to(torch.bfloat16).to(hn.device)
            lb = eng.buffer("logits", (64, V))
            lb[:Tn] = PAD
            am = eng.buffer("am", (64,), torch.int32)
            if not sampling:
                # greedy: node `path[j]` predicts the token of `path[j + 1]`, the last accepted node predicts arg-max(sample_p); every other node
                # predicts a token no child carries — the only path of accept length `acc` is the fixture's
                a_np = np.full(64, V - 1, np.int32)
                for j in range(acc):
                    a_np[path[j]] = tree_tok[path[j + 1]]
                a_np[path[acc]] = int(vo.argmax_first(sp))
                am.copy_(torch.from_numpy(a_np))
                eng.set_sampling(0.0, 0)
            else:
                # sampling: distribution of the last accepted node = sample_p (log-probabilities as logits, T = 1); the uniform table accepts
                # exactly the fixture's path (u = 0 at its rows, 2 elsewhere) and the final multinomial draws with the recorded uniform
                lp = np.full((Tn, V), PAD, np.float32)
                lp[:, : len(sp)] = 0.0  # (uniform rows: every candidate has p > 0, so u = 0 accepts it)
                lp[path[acc], : len(sp)] = np.log(np.maximum(sp, 1e-30))
                lb[:Tn] = torch.from_numpy(lp).to(torch.bfloat16).to(lb.device)
                u = np.full(ri.shape, 2.0, np.float32)
                for lvl in range(1, acc + 1):
                    rows = [r for r in range(ri.shape[0]) if (ri[r, : lvl + 1] == ri[best, : lvl + 1]).all()]
                    u[min(rows), lvl] = 0.0
                eng.set_sampling(1.0, seed=3)
                L.check(lib.vispec_set_uniform_override_host(eng.h, stream(), u.ctypes.data_as(C.c_void_p), ri.shape[0], ri.shape[1], C.c_float(float(g[f"u{i}"]))))
            eng.accept()
            st = eng.state()
            got_best, got_acc = eng.last_accept()
            assert got_acc == acc and (ri[got_best, : acc + 1] == path).all(), f"case {i}: accepted path"
            np.testing.assert_array_equal(eng.tokens(st["n_ctx"]), g[f"o_ids{i}"], err_msg=f"case {i}: input_ids")
            assert st["n_ctx"] == int(g[f"o_cur{i}"][0]) and st["new_token"] + 5 == int(g[f"o_new_token{i}"])  # (the fixture starts at new_token = 5)
            want_kv = torch.from_numpy(o_data).to(torch.bfloat16)
            assert torch.equal(kv[:ns, :, :, :32, :4].cpu(), want_kv), f"case {i}, cache tensor {pass_}: KV rows after the gather-compaction"
            assert not kv[:ns, :, :, :32, 4:].any() and not kv[ns:].any()
            ah = eng.buffer("accept_hidden", (16, D))[: acc + 1, :8].float().cpu().numpy()
            np.testing.assert_array_equal(ah, synth.bf16_grid(g[f"o_hidden{i}"]), err_msg=f"case {i}: hidden rows handed to the draft")
            di = eng.buffer("draft_ids", (16,), torch.int32)[: acc + 1].cpu().numpy()
            np.testing.assert_array_equal(di, g[f"o_draft_ids{i}"][-(acc + 1):], err_msg=f"case {i}: ids handed to the draft")
            if sampling:
                want_tok = int(g[f"o_token{i}"])
                if st["next_token"] != want_tok:  # bf16 log-probabilities move the CDF steps by < 1 %: only a draw on a step may differ
                    cdf = np.cumsum(sp.astype(np.float64))
                    assert np.abs(cdf - float(g[f"u{i}"]) * cdf[-1]).min() < 1e-2 and abs(st["next_token"] - want_tok) == 1
                L.check(lib.vispec_set_uniform_override_host(eng.h, stream(), None, 0, 0, C.c_float(0)))
                eng.set_sampling(0.0, 0)
            else:
                assert st["next_token"] == int(g[f"o_token{i}"]), f"case {i}: next token"


def _tuple_np(r):
    return r[0][0].cpu().numpy(), r[1].numpy(), r[2][0, 0].numpy() > 0, r[3].cpu().numpy()


def test_g4_g14_topk_genrate_fixture_inputs_through_the_draft_kernels(golden_dir):
    """Model.topK_genrate (cnets_ours.py:1043-1238) on the reference's own INPUTS (g4 / g14: hidden states, embeddings, image mask, ids of
    (a) the first call — draft prefill with image-token compression + the first tree — and (b) a decode round on accepted hidden rows) through
    vispec_draft_prefill / vispec_draft_round.

    What can be compared with what: these fixtures are the reference's fp32 CPU run of a RANDOM-weight pair, whose next-token distributions are
    nearly flat — in bf16 (what the reference computes in on a GPU, and the kernels here) the top log-probabilities of a row are EQUAL numbers
    or one ulp apart, so the tree a bf16 implementation grows is decided by ties, not by the fp32 order the fixture recorded: the bf16-emulating
    oracle itself shares only 16 / 12 of the fixture's 30 nodes, and two correct bf16 implementations differ from each other in a few nodes
    (an ulp of accumulation order).  Hence:
      (1) selection-independent floats against the FIXTURE: the last hidden row of the prefill forward and of the catch-up forward — g14's
          forward hook on the reference — within 2^-6 of scale;
      (2) everything that depends on selections against the ORACLE (bf16 mode) on the same fixture inputs, replayed along the device's own
          selections (tests/test_draft_round_gpu.py: every kept (token, score) equals the oracle's value, nothing better than the k-th pick is
          missing, the carried frontier is within tolerance of the best k; last level's hidden rows within 2^-6), and the integer tree tables
          EXACTLY equal to the oracle's build_tree on the device's own candidate lists — greedy and the sampling row order;
    the oracle being held to these very fixtures in fp32 mode by tests/test_oracle_golden.py (test_g4_topk_genrate, test_g14_...)."""
    from helpers import oracle_draft, oracle_target
    from test_draft_round_gpu import replay_levels
    from test_loop_gpu import dev_tree_inputs
    D = T["D"]
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def tree_exact(eng, sampling):
        sc, tk, pa = dev_tree_inputs(eng)
        tok, pos, mask, ret = eng.tree()
        w_tok, w_ret, w_mask, w_pos = vo.build_tree(sc, tk.astype(np.int64), pa.astype(np.int64), tok[0], eng.total_token - 1, eng.top_k, sampling=sampling)
        np.testing.assert_array_equal(tok, w_tok)
        np.testing.assert_array_equal(pos, w_pos)
        np.testing.assert_array_equal(mask, w_mask)
        np.testing.assert_array_equal(ret, w_ret)

    identical = []
    for fname, sampling in (("g4_topk.npz", False), ("g4_topk.npz", True), ("g14_tree_levels.npz", False)):
        g = load(golden_dir, fname)
        sm, _, _ = build(20, 14, False)
        eng, dl = sm.engine, sm.spec_layer
        head = sm.base_model.lm_head
        ot, _ = oracle_target(seed=20, bf16=True)
        od, _ = oracle_draft(num_q=2, seed=14, bf16=True)
        hidden, ids, emb, mask = synth.bf16_grid(g["hidden"]), g["ids"], synth.bf16_grid(g["embeds"]), g["mask"]
        L = len(ids) - 1
        # ---- (a) first call: prompt = ids[:-1], first token = ids[-1]
        eng.set_sampling(1.0 if sampling else 0.0, 0)
        eng.begin_request(ids[:-1].astype(np.int32), 200)
        dl.reset_kv()
        r = dl.topK_genrate(t_(hidden)[None].cuda(), t_(ids)[None].cuda(), head, None, inputs_embeds=t_(emb)[None].to(torch.bfloat16).cuda(),
                            image_mask=t_(mask)[None].cuda())
        od.reset_kv()
        e_shift = np.concatenate([emb[1:], od.ops.rd(od.w["embed_tokens.weight"][[int(ids[-1])]])], 0)  # cnets_ours.py:1081-1082
        out_c, kv, _ = od.forward_prefill(hidden, e_shift, mask.astype(bool))
        dlast = eng.buffer("draft_last", (16, D))[0].float().cpu().numpy()
        _close(dlast, out_c[-1])
        if fname.startswith("g14"):
            _close(dlast, g["a_first_last_row"])  # against the reference itself
        worst = [0.0]
        replay_levels(od, ot.lm_head, out_c[-1:], kv, L, eng, worst)
        tree_exact(eng, sampling)
        od2, _ = oracle_draft(num_q=2, seed=14, bf16=True)
        od2.reset_kv()
        w = od2.topK_genrate(hidden, ids, ot.lm_head, inputs_embeds=emb, image_mask=mask, sampling=sampling)
        identical.append(np.array_equal(_tuple_np(r)[0], w[0]))
        # ---- (b) decode round: the reference passes accept_hidden_state_new = h2 [a + 1, D] and the ids grown by a + 1 tokens; the library
        # stages the same through its accept step — impose that accept on a chain tree whose nodes carry h2 as their hidden states
        h2, ids2 = synth.bf16_grid(g["h2"]), g["ids2"]
        a1 = h2.shape[0]
        new = ids2[len(ids) - 1:]  # root (= the first token) + the a accepted draft tokens + the next token: a + 2 ids
        assert len(new) == a1 + 1 and new[0] == ids[-1]
        eng.set_total_token(a1)
        chain = np.arange(a1, dtype=np.int32)
        eng.set_tree(new[:a1].astype(np.int32), chain, np.array([(1 << (j + 1)) - 1 for j in range(a1)], np.uint64), chain[None])
        hn = eng.buffer("hidden_new", (64, D))
        hn[:a1] = t_(h2).to(torch.bfloat16).to(hn.device)
        am = eng.buffer("am", (64,), torch.int32)
        am[:a1] = t_(new[1:].astype(np.int32)).to(am.device)
        eng.set_sampling(0.0, 0)  # (the imposed accept is the greedy one, whatever the tree ordering under test)
        eng.accept()
        assert eng.last_accept() == (0, a1 - 1)
        eng.set_sampling(1.0 if sampling else 0.0, 0)
        eng.set_total_token(30)
        r2 = dl.topK_genrate(t_(h2)[None].cuda(), t_(ids2)[None].cuda(), head, None)
        out2, kv2 = od.forward_decode(h2, new[1:].astype(np.int64), kv)
        dlast2 = eng.buffer("draft_last", (16, D))[0].float().cpu().numpy()
        _close(dlast2, out2[-1])
        if fname.startswith("g14"):
            _close(dlast2, g["b_first_last_row"])  # against the reference itself
        replay_levels(od, ot.lm_head, out2[-1:], kv2, len(ids2) - 1, eng, worst)
        tree_exact(eng, sampling)
        w2 = od2.topK_genrate(h2, ids2, ot.lm_head, sampling=sampling)
        identical.append(np.array_equal(_tuple_np(r2)[0], w2[0]))
    print(f"g4 / g14 inputs through the draft kernels: {sum(identical)} of {len(identical)} trees token-identical to the bf16 oracle's own run "
          f"(the others differ at ties, every selection within tolerance)")
    assert sum(identical) >= len(identical) // 2


def _close(got, want, frac=2.0 ** -6):
    np.testing.assert_allclose(got, want, rtol=0, atol=frac * float(np.abs(want).max()))


def test_g13_lm_head_logsoftmax_topk_and_input_fusion_at_the_real_dims(lib, golden_dir):
    """The reference's torch ops at the REAL LLaVA-7B dims (D = 4096, V = 32064; weights re-derived from the fixture's seeds):
    LM head -> log-softmax -> top-k (cnets_ours.py:1109-1123) through vispec_gemm_skinny + vispec_logsoftmax_topk, and the draft's input fusion
    fc(cat(emb, img_fc(cat(h, g)))) (cnets_ours.py:982-988) through two K = 8192 GEMMs with bias.  The fixture is torch fp32 with fp32 weights; the
    kernels hold bf16 weights and bf16 logits (one ulp at this logit scale = 0.03, the spacing of the top order statistics of 32 064 logits): every
    selected index must be, in the reference's own fp32 numbers, within 2 ulp of the reference's k-th value, every log-probability within 2 ulp,
    and at least half of the rows index-identical (5 of 8 measured)."""
    g = load(golden_dir, "g13_real_dims.npz")
    sm, _, _ = build(50, 60, True)
    eng = sm.engine
    D, V, k = 4096, 32064, 8
    rng = np.random.default_rng(1300)
    W = rng.standard_normal((V, D), dtype=np.float32) * np.float32(0.02)
    h = g["h"]
    M = h.shape[0]
    X, Wp = tb(h), packed(synth.bf16_grid(W))
    Y = torch.zeros(M, V, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny(eng.h, stream(), p(X), D, p(Wp), None, p(Y), V, None, 0, M, V, D, 0))
    idx = torch.zeros(M, k, dtype=torch.int32, device=dev())
    lp = torch.zeros(M, k, dtype=torch.float32, device=dev())
    L.check(lib.vispec_logsoftmax_topk(eng.h, stream(), p(Y), V, M, V, k, p(idx), p(lp)))
    torch.cuda.synchronize()
    got_idx, got_lp = idx.cpu().numpy(), lp.cpu().numpy()
    logits32 = h @ W.T  # the reference's fp32 logits (the fixture stores only their top-k and lse)
    logp32 = logits32 - g["lse"][:, None]
    ulp = 2.0 ** -7 * float(np.abs(logits32).max())  # one bf16 ulp at the logit scale
    exact_rows = 0
    for r in range(M):
        np.testing.assert_allclose(logp32[r][g["top_idx"][r]], g["top_logp"][r], rtol=0, atol=3e-5)  # the recomputation is the fixture's
        if np.array_equal(got_idx[r], g["top_idx"][r]):
            exact_rows += 1
        else:  # a swap / replacement is legitimate only between candidates closer than 2 ulp in the reference's own numbers
            kth = g["top_logp"][r][-1]
            assert (logp32[r][got_idx[r]] >= kth - 2 * ulp).all() and len(set(got_idx[r].tolist())) == k, f"row {r}: {got_idx[r]} vs {g['top_idx'][r]}"
            assert (np.abs(np.sort(logp32[r][got_idx[r]])[::-1] - g["top_logp"][r]) <= 2 * ulp).all()
        np.testing.assert_allclose(got_lp[r], logp32[r][got_idx[r]], rtol=0, atol=2 * ulp)
        assert (np.diff(got_lp[r]) <= 0).all()
    assert exact_rows >= M // 2, f"only {exact_rows} of {M} rows select the reference's indices"
    print(f"g13 through the HIP LM head + top-k: {exact_rows}/{M} rows index-identical to the reference (fp32), all within 2 bf16 ulp")
    del W, Wp
    # ---- input fusion at K = 2 D
    rng2 = np.random.default_rng(1301)
    fc_w = rng2.standard_normal((D, 2 * D), dtype=np.float32) * np.float32(0.02)
    fc_b = rng2.standard_normal(D, dtype=np.float32) * np.float32(0.02)
    ifc_w = rng2.standard_normal((D, 2 * D), dtype=np.float32) * np.float32(0.02)
    ifc_b = rng2.standard_normal(D, dtype=np.float32) * np.float32(0.02)
    emb, hid, gg = g["emb"], g["hid"], g["g"]
    n = emb.shape[0]
    x1 = tb(np.concatenate([hid, np.broadcast_to(gg, hid.shape)], -1))
    y1 = torch.zeros(n, D, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny(eng.h, stream(), p(x1), 2 * D, p(packed(synth.bf16_grid(ifc_w))), p(tb(synth.bf16_grid(ifc_b))), p(y1), D, None, 0, n, D, 2 * D, 0))
    x2 = torch.cat([tb(emb), y1], -1).contiguous()
    y2 = torch.zeros(n, D, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny(eng.h, stream(), p(x2), 2 * D, p(packed(synth.bf16_grid(fc_w))), p(tb(synth.bf16_grid(fc_b))), p(y2), D, None, 0, n, D, 2 * D, 0))
    torch.cuda.synchronize()
    _close(y2.float().cpu().numpy(), g["fused"])
