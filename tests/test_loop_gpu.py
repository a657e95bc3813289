"""GPU parity of the whole draft-and-verify path (through SpecModel -> C-ABI -> HIP) against the numpy oracle in
bf16-emulation mode, on the tiny seeded models of tests/golden (same weights as the reference-captured fixtures).

 * integer logic (tree construction, accept, KV compaction bookkeeping) is compared EXACTLY, by replaying the oracle on
   the very inputs the device kernels saw (read back from the ctx buffers);
 * float stages (draft prefill with image compression, draft round, target verify) within bf16 tolerances stated inline;
 * token streams and per-round accept lengths of the structured (successor) pairs: exact vs the oracle AND vs the
   committed golden fixtures captured from the reference itself (g8_loop.npz);
 * the reference's own invariant: speculative output == greedy AR output of the same target (same HIP kernels, T=1).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, oracle_draft, oracle_target, vo  # noqa: E402
from vispec_amd import synth  # noqa: E402
from vispec_amd.engine import DraftConfig, TargetConfig  # noqa: E402
from vispec_amd.model import SpecModel  # noqa: E402

IMG_TOK = T["V"] - 1


def build(seed_t, seed_d, structured, rho=0.25, arch="LlamaForCausalLM", num_q=2, **kw):
    tw = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=seed_t, structured=structured)
    dw = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], num_q=num_q, seed=seed_d, structured=structured,
                                  target_embed=tw["model.embed_tokens.weight"] if structured else None, rho=rho)
    tcfg = TargetConfig(hidden_size=T["D"], num_heads=T["H"], num_kv_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"],
                        num_layers=T["NL"], max_position_embeddings=T["max_pos"], architectures=(arch,), image_token_index=IMG_TOK)
    dcfg = DraftConfig(hidden_size=T["D"], num_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"], max_position_embeddings=T["max_pos"])
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw, num_q=num_q, **kw)
    ot = vo.TargetLlama(vo.TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"]), tw, bf16=True)
    od = vo.DraftModel(vo.DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"], num_q=num_q), dw, bf16=True)
    return sm, ot, od


def dev_tree_inputs(eng):
    k, d = eng.top_k, eng.depth
    n_all = k + d * k * k
    sc = eng.buffer("scores_all", (n_all,), torch.float32).cpu().numpy()
    tk = eng.buffer("tokens_all", (n_all,), torch.int32).cpu().numpy()
    pa = eng.buffer("parents_all", (1 + d * k,), torch.int32).cpu().numpy()
    return sc, tk, pa


def check_tree_exact(eng):
    """device tree == oracle build_tree (cnets_ours.py:1169-1238) on the device's own score/token/parent lists."""
    sc, tk, pa = dev_tree_inputs(eng)
    tok, pos, mask, ret = eng.tree()
    w_tok, w_ret, w_mask, w_pos = vo.build_tree(sc, tk.astype(np.int64), pa.astype(np.int64), tok[0], eng.total_token - 1, eng.top_k)
    np.testing.assert_array_equal(tok, w_tok)
    np.testing.assert_array_equal(pos, w_pos)
    np.testing.assert_array_equal(mask, w_mask)
    np.testing.assert_array_equal(ret, w_ret)
    return tok, pos, mask, ret


@pytest.mark.parametrize("case", ["succ0", "succ1"])
def test_text_loop_matches_oracle_and_reference_golden(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    si = int(case[4:])
    sm, ot, od = build(50 + si, 60 + si, True)
    ids = g[f"{case}_ids"]
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=40, log=True, return_acceptance_len=True)
    out = out[0].cpu().numpy()
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=40, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out, o_out)                      # HIP == oracle(bf16)
    assert (new_token, idx) == (o_new, o_idx) and acc == o_acc
    np.testing.assert_array_equal(out, g[f"{case}_out"])           # == the reference's own fp32 run (committed fixture)
    np.testing.assert_array_equal(acc, g[f"{case}_acc"])
    assert max(acc) == 4 and min(acc) == 0
    # greedy invariance with the same kernels at T=1
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=len(out) - len(ids) - 1)[0].cpu().numpy()
    np.testing.assert_array_equal(ar[: len(out)], out[: len(ar)])


def test_image_loop_matches_oracle_and_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    sm, ot, od = build(70, 71, True, arch="LlavaNextForConditionalGeneration")
    ids, emb, mask = g["img_ids"].copy(), g["img_emb"], g["img_mask"]
    ids_in = ids.copy()
    ids_in[mask] = IMG_TOK
    feats = torch.from_numpy(emb[mask]).to(torch.bfloat16)
    # text rows of the fixture are the target's own embeddings of `ids`; feed ids + image features like the reference harness
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids_in)[None], pixel_values=feats.cuda(), max_new_tokens=30, log=True,
                                               return_acceptance_len=True)
    out = out[0].cpu().numpy()
    L = len(ids)
    np.testing.assert_array_equal(out[L:], g["img_out"][L:])
    np.testing.assert_array_equal(acc, g["img_acc"])
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[L:], o_out[L:])
    assert acc == o_acc
    st = sm.engine.state()
    assert st["draft_len"] == st["n_ctx"] - int(mask.sum()) + (sm.engine.num_q - 1)  # compressed draft KV


@pytest.mark.parametrize("num_q,n_pre,n_img,n_post", [(2, 6, 21, 9), (5, 9, 17, 4), (2, 0, 0, 30), (3, 1, 70, 2)])
def test_draft_prefill_stage(num_q, n_pre, n_img, n_post):
    """Fused vision-adaptor prefill (cnets_ours.py:879-975): last hidden row, compressed KV, g, first tree."""
    sm, ot, od = build(20, 14, False, arch="LlavaNextForConditionalGeneration", num_q=num_q)
    eng = sm.engine
    rng = np.random.default_rng(n_img + num_q)
    L = n_pre + n_img + n_post
    hidden = synth.bf16_grid(rng.standard_normal((L, T["D"]), dtype=np.float32))
    embeds = synth.bf16_grid(rng.standard_normal((L, T["D"]), dtype=np.float32) * 0.05)
    mask = np.zeros(L, bool)
    mask[n_pre : n_pre + n_img] = True
    ids = rng.integers(3, IMG_TOK, size=L + 1)
    eng.begin_request(ids[:L], 64)
    first = torch.tensor([ids[L]], dtype=torch.int32, device="cuda")
    eng.draft_prefill(torch.from_numpy(hidden).to(torch.bfloat16).cuda(), torch.from_numpy(embeds).to(torch.bfloat16).cuda(),
                      mask if n_img else None, first)
    tok, pos, tmask, ret = check_tree_exact(eng)
    od.reset_kv()
    od.topK_genrate(hidden, ids, ot.lm_head, inputs_embeds=embeds, image_mask=mask if n_img else None)
    Lc = od.stable_kv[0].shape[1]
    assert eng.state()["draft_len"] == Lc
    D, H = T["D"], T["H"]
    kv = eng.draft_kv.float().cpu().numpy()  # [2, H, max_pos, hd]
    # K/V of the compressed sequence: 2 ulp at bf16 (GEMM accumulation order) ; g and last hidden likewise
    tol = lambda a, b, s=None: np.testing.assert_allclose(a, b, rtol=2.0 ** -6, atol=2.0 ** -6 * (np.abs(b).max() if s is None else s))
    tol(kv[0][:, :Lc], od.stable_kv[0])
    tol(kv[1][:, :Lc], od.stable_kv[1])
    tol(eng.buffer("draft_g", (1, D)).float().cpu().numpy(), od.last_img_hidden)
    assert tok[0] == ids[L]


@pytest.mark.parametrize("tag", ["two", "three_q3", "image_last", "one_q5"])
def test_multi_image_draft_prefill_against_the_repaired_reference_fixture(golden_dir, tag):
    """The HIP draft prefill on MULTI-IMAGE prompts, compared DIRECTLY with fixture G16 — the reference's own Model.forward with its two
    scatter-matrix indices repaired (reference-intent; the published code crashes on a second image run): compressed
    K/V of every run, the final global feature g and the draft length.  The fixture is at the level of Model.forward (embeddings
    already shifted by one, last row = the draft's embedding of the sampled token); the C-ABI takes the un-shifted rows + that token."""
    g = np.load(os.path.join(golden_dir, "g16_multi_image_repaired.npz"))
    q = int(g[f"{tag}_q"])
    sm, ot, od = build(20, 16, False, arch="LlavaNextForConditionalGeneration", num_q=q)
    eng = sm.engine
    hidden, shifted, mask = g[f"{tag}_hidden"], g[f"{tag}_embeds"], g[f"{tag}_mask"]
    L = hidden.shape[0]
    unshifted = np.concatenate([np.zeros((1, T["D"]), np.float32), shifted[:-1]], 0)  # row 0 is never read (cnets_ours.py:1081)
    ids = np.random.default_rng(5).integers(3, IMG_TOK, size=L)
    eng.begin_request(ids, 64)
    first = torch.tensor([int(g[f"{tag}_first_tok"])], dtype=torch.int32, device="cuda")
    eng.draft_prefill(torch.from_numpy(hidden).to(torch.bfloat16).cuda(), torch.from_numpy(unshifted).to(torch.bfloat16).cuda(), mask, first)
    check_tree_exact(eng)
    want_k, want_v = g[f"{tag}_k"], g[f"{tag}_v"]  # [H, L_c, hd]
    Lc = want_k.shape[1]
    assert eng.state()["draft_len"] == Lc
    kv = eng.draft_kv.float().cpu().numpy()
    tol = lambda a, b: np.testing.assert_allclose(a, b, rtol=2.0 ** -6, atol=2.0 ** -6 * np.abs(b).max())
    tol(kv[0][:, :Lc], want_k)
    tol(kv[1][:, :Lc], want_v)
    tol(eng.buffer("draft_g", (1, T["D"])).float().cpu().numpy(), g[f"{tag}_g"])
    dlast = eng.buffer("draft_last", (16, T["D"]))[:1].float().cpu().numpy()
    np.testing.assert_allclose(dlast[0], g[f"{tag}_out_last"], rtol=0, atol=2.0 ** -6 * np.abs(g[f"{tag}_out_last"]).max())


def test_draft_prefill_stage_on_the_prefill_gemm():
    """Stages of >= 64 rows run on the prefill GEMM (csrc/gemm_prefill.h: 128 x 128 tiles, operand rows gathered / concatenated while
    staged, K|V scatter and rotary+append epilogues) instead of 32-row passes of the skinny kernel.  A draft wide enough for the
    one-launch q|k|v form (3D = 4608 rows) with a prompt whose text segments, image run and compressed length all exceed 64 rows,
    against the oracle: compressed K/V (rotary at the ORIGINAL positions), g, last hidden row, first tree."""
    D, H, I, V, NL, P = 1536, 12, 1024, 1008, 1, 512
    tw = synth.make_target_weights(D, H, I, V, NL, seed=120)
    dw = synth.make_draft_weights(D, H, I, V, num_q=3, seed=121)
    tcfg = TargetConfig(hidden_size=D, num_heads=H, num_kv_heads=H, intermediate_size=I, vocab_size=V, num_layers=NL,
                        max_position_embeddings=P, architectures=("LlavaNextForConditionalGeneration",), image_token_index=IMG_TOK)
    dcfg = DraftConfig(hidden_size=D, num_heads=H, intermediate_size=I, vocab_size=V, max_position_embeddings=P)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw, num_q=3)
    ot = vo.TargetLlama(vo.TargetConfig(D, H, H, I, V, NL, P), tw, bf16=True)
    od = vo.DraftModel(vo.DraftConfig(D, H, I, V, P, num_q=3), dw, bf16=True)
    eng = sm.engine
    assert eng.lib.vispec_qkv_rope_fused(3 * D)
    rng = np.random.default_rng(122)
    n_pre, n_img, n_post = 70, 150, 81
    L = n_pre + n_img + n_post
    hidden = synth.bf16_grid(rng.standard_normal((L, D), dtype=np.float32))
    embeds = synth.bf16_grid(rng.standard_normal((L, D), dtype=np.float32) * 0.05)
    mask = np.zeros(L, bool)
    mask[n_pre: n_pre + n_img] = True
    ids = rng.integers(3, IMG_TOK, size=L + 1)
    eng.begin_request(ids[:L], 64)
    first = torch.tensor([ids[L]], dtype=torch.int32, device="cuda")
    eng.draft_prefill(torch.from_numpy(hidden).to(torch.bfloat16).cuda(), torch.from_numpy(embeds).to(torch.bfloat16).cuda(), mask, first)
    tok, pos, tmask, ret = check_tree_exact(eng)
    od.reset_kv()
    od.topK_genrate(hidden, ids, ot.lm_head, inputs_embeds=embeds, image_mask=mask)
    Lc = od.stable_kv[0].shape[1]
    assert Lc == L - n_img + 2 and eng.state()["draft_len"] == Lc
    kv = eng.draft_kv.float().cpu().numpy()
    tol = lambda a, b: np.testing.assert_allclose(a, b, rtol=2.0 ** -6, atol=2.0 ** -6 * np.abs(b).max())
    tol(kv[0][:, :Lc], od.stable_kv[0])
    tol(kv[1][:, :Lc], od.stable_kv[1])
    tol(eng.buffer("draft_g", (1, D)).float().cpu().numpy(), od.last_img_hidden)
    assert tok[0] == ids[L]


def test_round_stages_against_oracle():
    """One verify + accept + draft round on the random (unstructured) pair, stage by stage."""
    sm, ot, od = build(21, 13, False)
    eng = sm.engine
    rng = np.random.default_rng(77)
    ids = rng.integers(3, T["V"], size=23)
    # prefill through the product path
    out = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=0, log=True, return_acceptance_len=True)
    # replay round 1 in the oracle from the device's own tree
    pkv, pkv_data, cur = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    lg, hid = ot.forward(pkv, input_ids=ids)
    first = int(np.argmax(lg[-1]))
    toks = out[0][0].cpu().numpy()
    assert toks[len(ids)] == first
    # a fresh request so that the device state is exactly "after prefill"
    sm2, _, _ = build(21, 13, False)
    e2 = sm2.engine
    emb = torch.nn.functional.embedding(torch.from_numpy(ids).cuda(), e2.tw.embed)
    logits, hidden = sm2.base_model.prefill(emb)
    f = sm2._first_token(logits)
    e2.begin_request(ids, 64)
    demb = torch.nn.functional.embedding(torch.cat([torch.from_numpy(ids).cuda(), f.long()])[:-1], e2.dw.t["embed"]).contiguous()
    e2.draft_prefill(hidden, demb, None, f)
    tok, pos, tmask, ret = check_tree_exact(e2)
    # --- target verify forward: logits of all T nodes vs oracle with the same tree
    e2.target_forward()
    V, D = T["V"], T["D"]
    got_logits = e2.buffer("logits", (64, V))[: len(tok)].float().cpu().numpy()
    got_hidden = e2.buffer("hidden_new", (64, D))[: len(tok)].float().cpu().numpy()
    ot.tree_mask = tmask
    want_logits, want_hidden = ot.forward(pkv, input_ids=tok, position_ids=pos + len(ids))
    # 2 target layers of bf16 with different accumulation orders: a few ulp at the activations' scale
    np.testing.assert_allclose(got_hidden, want_hidden, rtol=0, atol=2.0 ** -6 * np.abs(want_hidden).max())
    np.testing.assert_allclose(got_logits, want_logits, rtol=0, atol=2.0 ** -6 * np.abs(want_logits).max())
    # BASELINE.json asks for "logits within 1e-3".  The logits are bf16 tensors in the reference too (one ulp = 3.9e-3 of the value),
    # so 1e-3 can only hold on average: measured 1.4e-3 of the logit scale in the mean, 8e-3 (two ulps) at worst, a quarter of the
    # entries bit-equal.  Asserted with a 2x margin; the token decisions built on them are compared exactly below.
    rel = np.abs(got_logits - want_logits) / np.abs(want_logits).max()
    assert rel.mean() <= 3e-3 and rel.max() <= 2.0 ** -6, (rel.mean(), rel.max())
    am = e2.buffer("am", (64,), torch.int32)[: len(tok)].cpu().numpy()
    np.testing.assert_array_equal(am, np.argmax(got_logits, axis=1))  # argmax kernel == numpy on the SAME logits
    # --- accept: oracle evaluate_posterior on the device's logits gathered like utils.py:411
    cand = np.concatenate([tok, [-1]])[ret]
    best, a, _ = vo.evaluate_posterior_greedy(got_logits[ret], cand)
    e2.accept()
    st = e2.state()
    assert (st["accept_len"], st["n_ctx"], st["new_token"]) == (a, len(ids) + a + 1, a + 1)
    assert st["next_token"] == int(np.argmax(got_logits[ret[best, a]]))
    sel = e2.buffer("sel", (16,), torch.int32).cpu().numpy()
    np.testing.assert_array_equal(sel[: a + 1], ret[best, : a + 1])
    # --- KV compaction (utils.py:529-538): rows n+sel[j] -> n+j, bit-exact copy of what the verify forward wrote
    n = len(ids)
    kvd = e2.target_kv.float().cpu().numpy()
    for j in range(a + 1):
        np.testing.assert_allclose(kvd[:, 0, :, n + j], pkv_data[0][:, 0, :, n + ret[best, j]], rtol=0,
                                   atol=2.0 ** -6 * np.abs(pkv_data[0]).max())
    # --- next draft round: tree logic exact on the device's own candidate lists
    e2.draft_round()
    check_tree_exact(e2)
    assert e2.state()["draft_len"] == len(ids) + a + 1


def test_forced_accept_is_bench_only_and_consistent():
    sm, _, _ = build(50, 60, True)
    rng = np.random.default_rng(5)
    ids = rng.integers(3, T["V"], size=12)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=20, log=True, return_acceptance_len=True,
                                               forced_accept=lambda r: r % 3)
    assert all(a <= (r % 3) for r, a in enumerate(acc))
    assert out.shape[1] == len(ids) + sum(a + 1 for a in acc)


def test_multi_image_prompt_matches_oracle():
    """4 image runs (BASELINE config 3's shape of prompt).  The reference crashes on this (index bug in the trans_mat it builds
    for a scatter nobody reads, cnets_ours.py:938-941); oracle and HIP implement the intent — parity unpinned
    upstream, pinned between oracle and HIP."""
    sm, ot, od = build(70, 71, True, arch="LlavaNextForConditionalGeneration")
    rng = np.random.default_rng(123)
    D = T["D"]
    segs = [(3, 17), (5, 40), (1, 9), (7, 33)]
    ids, mask = [], []
    for n_txt, n_img in segs:
        ids += rng.integers(3, IMG_TOK, size=n_txt).tolist() + [IMG_TOK] * n_img
        mask += [False] * n_txt + [True] * n_img
    ids += rng.integers(3, IMG_TOK, size=6).tolist()
    mask += [False] * 6
    ids, mask = np.array(ids), np.array(mask)
    feats = synth.bf16_grid(rng.standard_normal((int(mask.sum()), D), dtype=np.float32) * 0.05)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(),
                                               max_new_tokens=24, log=True, return_acceptance_len=True)
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    emb[mask] = feats
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=24, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and max(acc) >= 3
    st = sm.engine.state()
    assert st["draft_len"] == st["n_ctx"] - int(mask.sum()) + len(segs) * (sm.engine.num_q - 1)


def test_multi_image_loop_matches_the_repaired_reference_fixture(golden_dir):
    """Fixture g17: the reference's loop body on a prompt with THREE image runs, its draft forward repaired (reference-intent, see
    tests/golden/gen_golden.py): the HIP loop's token stream and accept lengths are the reference's, bit for bit."""
    g = np.load(os.path.join(golden_dir, "g17_multi_image_loop.npz"))
    sm, ot, od = build(70, 71, True, arch="LlavaNextForConditionalGeneration")
    ids, emb, mask = g["ids"].copy(), g["emb"], g["mask"]
    feats = torch.from_numpy(emb[mask]).to(torch.bfloat16)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], pixel_values=feats.cuda(), max_new_tokens=30, log=True,
                                               return_acceptance_len=True)
    out = out[0].cpu().numpy()
    L = len(ids)
    np.testing.assert_array_equal(out[L:], g["out"][L:])
    np.testing.assert_array_equal(acc, g["acc"])
    st = sm.engine.state()
    assert st["draft_len"] == st["n_ctx"] - int(mask.sum()) + 3 * (sm.engine.num_q - 1)


def test_llava15_semantics_no_compression(golden_dir):
    """BASELINE config 0 semantics (LLaVA-1.5): image features reach the target, the draft never compresses."""
    sm, ot, od = build(70, 71, True, arch="LlavaForConditionalGeneration")
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids, emb, mask = g["img_ids"].copy(), g["img_emb"], g["img_mask"]
    ids_in = ids.copy()
    ids_in[mask] = IMG_TOK
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids_in)[None], pixel_values=torch.from_numpy(emb[mask]).to(torch.bfloat16).cuda(),
                                               max_new_tokens=20, log=True, return_acceptance_len=True)
    st = sm.engine.state()
    assert st["draft_len"] == st["n_ctx"]  # no image-token compression in the draft's KV
    # target saw the image features: same greedy tokens as the LLaVA-NeXT run of the same request
    L = len(ids)
    np.testing.assert_array_equal(out[0].cpu().numpy()[L:L + 16], g["img_out"][L:L + 16])


def test_hipgraph_replay_equals_direct_launches():
    """On a side stream the round functions are captured once and replayed (hipGraph); on the null stream they launch directly.
    Same tokens, same accept lengths; a second request (different prompt length -> new capture key) still matches the oracle."""
    sm, ot, od = build(50, 60, True)
    rng = np.random.default_rng(11)
    side = torch.cuda.Stream()
    for n in (14, 23):
        ids = rng.integers(3, T["V"], size=n)
        direct = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=30, log=True, return_acceptance_len=True)
        with torch.cuda.stream(side):
            graphed = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=30, log=True, return_acceptance_len=True)
            ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=20)
            side.synchronize()
        np.testing.assert_array_equal(direct[0].cpu().numpy(), graphed[0].cpu().numpy())
        assert direct[1:] == graphed[1:]
        o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=30, max_pos=T["max_pos"])
        np.testing.assert_array_equal(graphed[0][0].cpu().numpy(), o_out)
        k = min(ar.shape[1], len(o_out))
        np.testing.assert_array_equal(ar[0, :k].cpu().numpy(), o_out[:k])
    gs = sm.engine.graph_stats()
    assert gs["captures"] >= 3 and gs["replays"] > gs["captures"] and gs["direct"] > 0, gs


def build_qwen(seed_t=91, seed_d=92, structured=True, rho=0.25):
    Q = synth.QWEN_TINY
    IMG = Q["V"] - 1
    tw = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=seed_t, structured=structured, qkv_bias=True, H_kv=Q["Hkv"])
    dw = synth.make_draft_weights(Q["D"], Q["H"], Q["I"], Q["V"], seed=seed_d, structured=structured, qkv_bias=True,
                                  target_embed=tw["model.embed_tokens.weight"] if structured else None, rho=rho)
    tcfg = TargetConfig(hidden_size=Q["D"], num_heads=Q["H"], num_kv_heads=Q["Hkv"], intermediate_size=Q["I"], vocab_size=Q["V"], num_layers=Q["NL"],
                        max_position_embeddings=Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True,
                        architectures=("Qwen2_5_VLForConditionalGeneration",), image_token_index=IMG, attn_impl="sdpa",
                        mrope_section=Q["mrope_section"])
    dcfg = DraftConfig(hidden_size=Q["D"], num_heads=Q["H"], intermediate_size=Q["I"], vocab_size=Q["V"], max_position_embeddings=Q["max_pos"],
                       rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw)
    ot = vo.TargetLlama(vo.TargetConfig(Q["D"], Q["H"], Q["Hkv"], Q["I"], Q["V"], Q["NL"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"],
                                        attn_impl="sdpa", mrope_section=Q["mrope_section"]), tw, bf16=True)
    od = vo.DraftModel(vo.DraftConfig(Q["D"], Q["H"], Q["I"], Q["V"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"]), dw, bf16=True)
    return sm, ot, od, IMG


def test_qwen25vl_target_loop_matches_oracle():
    """Qwen2.5-VL-shaped target (GQA 4/2, q/k/v bias, theta 1e6, eps 1e-6, SDPA scores, multimodal rotary prefill with two image
    runs, rope_delta-shifted decode positions) + its ViSpec draft (MHA with q/k/v bias): HIP loop == oracle loop, == greedy AR."""
    sm, ot, od, IMG = build_qwen()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(17)
    grids = [(1, 6, 8), (1, 4, 4)]  # -> 12 and 4 merged image tokens
    ids = np.concatenate([rng.integers(3, IMG, 4), np.full(12, IMG), rng.integers(3, IMG, 3), np.full(4, IMG), rng.integers(3, IMG, 6)])
    mask = ids == IMG
    feats = synth.bf16_grid(rng.standard_normal((int(mask.sum()), Q["D"]), dtype=np.float32) * 0.05)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(),
                                               image_grid_thw=torch.tensor(grids), max_new_tokens=24, log=True, return_acceptance_len=True)
    pos3, delta = synth.qwen_rope_index(ids, IMG, grids)
    assert delta < 0
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    emb[mask] = feats
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=24, max_pos=Q["max_pos"],
                                                 position_ids=pos3, rope_delta=delta)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and max(acc) >= 3
    st = sm.engine.state()
    assert st["draft_len"] == st["n_ctx"] - int(mask.sum()) + 2 * (sm.engine.num_q - 1)
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(),
                              image_grid_thw=torch.tensor(grids), max_new_tokens=20)
    n = min(ar.shape[1], len(o_out))
    np.testing.assert_array_equal(ar[0, :n].cpu().numpy(), o_out[:n])


def fp8_codes_of(sm, D, H, Hkv, I, NL):
    """The product's own e4m3 codes and per-output-channel scales, split back per projection: name -> (values fp32, scales) for the oracle
    (handing it the very codes the kernels stream means exact .5 ties of w / scale cannot differ between the two quantisers)."""
    etw = sm.engine.tw
    f8 = lambda t: t.view(torch.float8_e4m3fn).float().cpu().numpy()
    hd, codes = D // H, {}
    for i in range(NL):
        p_ = f"model.layers.{i}."
        c8, s8 = etw.codes8[i], etw.scales8[i]
        nq, nk = H * hd, Hkv * hd
        for nm, key, lo, hi in (("self_attn.q_proj", "wqkv", 0, nq), ("self_attn.k_proj", "wqkv", nq, nq + nk), ("self_attn.v_proj", "wqkv", nq + nk, nq + 2 * nk),
                                ("self_attn.o_proj", "wo", 0, D), ("mlp.gate_proj", "wgu", 0, I), ("mlp.up_proj", "wgu", I, 2 * I),
                                ("mlp.down_proj", "wdown", 0, D)):
            codes[p_ + nm + ".weight"] = (f8(c8[key][lo:hi]), s8[key][lo:hi].cpu().numpy())
    codes["lm_head.weight"] = (f8(etw.c_lm_head8), etw.s_lm_head8.cpu().numpy())
    return codes


def build_qwen_fp8():
    """Qwen2.5-VL-tiny with fp8 (e4m3, per-output-channel) target weights + the oracle built on the product's own codes and scales."""
    Q = synth.QWEN_TINY
    IMG = Q["V"] - 1
    tw = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=91, structured=True, qkv_bias=True, H_kv=Q["Hkv"])
    dw = synth.make_draft_weights(Q["D"], Q["H"], Q["I"], Q["V"], seed=92, structured=True, qkv_bias=True,
                                  target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    tcfg = TargetConfig(hidden_size=Q["D"], num_heads=Q["H"], num_kv_heads=Q["Hkv"], intermediate_size=Q["I"], vocab_size=Q["V"], num_layers=Q["NL"],
                        max_position_embeddings=Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True,
                        architectures=("Qwen2_5_VLForConditionalGeneration",), image_token_index=IMG, attn_impl="sdpa",
                        mrope_section=Q["mrope_section"])
    dcfg = DraftConfig(hidden_size=Q["D"], num_heads=Q["H"], intermediate_size=Q["I"], vocab_size=Q["V"], max_position_embeddings=Q["max_pos"],
                       rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw, target_weight_dtype="fp8")
    codes = fp8_codes_of(sm, Q["D"], Q["H"], Q["Hkv"], Q["I"], Q["NL"])
    ot = vo.TargetLlama(vo.TargetConfig(Q["D"], Q["H"], Q["Hkv"], Q["I"], Q["V"], Q["NL"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"],
                                        attn_impl="sdpa", mrope_section=Q["mrope_section"]), tw, bf16=True, fp8=codes)
    od = vo.DraftModel(vo.DraftConfig(Q["D"], Q["H"], Q["I"], Q["V"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"]), dw, bf16=True)
    return sm, ot, od, IMG


def test_fp8_target_weights_loop_matches_oracle():
    """BASELINE config 5 shape of model: Qwen2.5-VL-like target with fp8 (e4m3, per-output-channel) weights in every streamed
    GEMM incl. lm_head; bf16 draft.  Oracle = same quantised model (numpy e4m3); the product prefill (torch GEMMs) multiplies by the same
    codes and scales on an fp32 accumulator (test_fp8_prefill_computes_with_codes_and_scales_like_the_decode_gemms)."""
    sm, ot, od, IMG = build_qwen_fp8()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(19)
    grids = [(1, 6, 8)]
    ids = np.concatenate([rng.integers(3, IMG, 6), np.full(12, IMG), rng.integers(3, IMG, 8)])
    mask = ids == IMG
    feats = synth.bf16_grid(rng.standard_normal((12, Q["D"]), dtype=np.float32) * 0.05)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(),
                                               image_grid_thw=torch.tensor(grids), max_new_tokens=24, log=True, return_acceptance_len=True)
    pos3, delta = synth.qwen_rope_index(ids, IMG, grids)
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    emb[mask] = feats
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=24, max_pos=Q["max_pos"],
                                         position_ids=pos3, rope_delta=delta)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and max(acc) >= 3
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(),
                              image_grid_thw=torch.tensor(grids), max_new_tokens=20)
    n = min(ar.shape[1], len(o_out))
    np.testing.assert_array_equal(ar[0, :n].cpu().numpy(), o_out[:n])


def test_fp8_prefill_computes_with_codes_and_scales_like_the_decode_gemms():
    """fp8 target weights: the PyTorch prefill multiplies by the e4m3 CODES (exact in bf16) on an fp32 accumulator and applies the
    per-output-channel scale before the one bf16 rounding — the W8A16 arithmetic of the decode kernels and of the oracle's quantised
    model — not by weights rounded a second time when de-quantised.  Prefill hidden rows, K/V rows and last-row logits vs the oracle."""
    sm, ot, od, IMG = build_qwen_fp8()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(29)
    grids = [(1, 6, 8)]
    ids = np.concatenate([rng.integers(3, IMG, 7), np.full(12, IMG), rng.integers(3, IMG, 9)])
    mask = ids == IMG
    feats = synth.bf16_grid(rng.standard_normal((12, Q["D"]), dtype=np.float32) * 0.05)
    hidden, demb, mask_np, first = sm._start_request(torch.from_numpy(ids)[None], None,
                                                     dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(), image_grid_thw=torch.tensor(grids)),
                                                     max_new_tokens=16)
    pos3, delta = synth.qwen_rope_index(ids, IMG, grids)
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    emb[mask] = feats
    pkv, pkv_data, cur = vo.initialize_past_key_values(Q["NL"], Q["Hkv"], Q["max_pos"], Q["D"] // Q["H"])
    lg, hid = ot.forward(pkv, inputs_embeds=emb, position_ids=pos3)
    L_ = len(ids)
    got_h = hidden.float().cpu().numpy()
    np.testing.assert_allclose(got_h, hid, rtol=0, atol=2.0 ** -6 * np.abs(hid).max())
    kvd = sm.engine.target_kv.float().cpu().numpy()
    want_kv = pkv_data[0][:, 0, :, :L_]
    np.testing.assert_allclose(kvd[:, 0, :, :L_], want_kv, rtol=0, atol=2.0 ** -6 * np.abs(want_kv).max())
    # most entries are bit-equal (same products, same single rounding; only the accumulation order differs)
    assert np.mean(kvd[:, 0, :, :L_] == want_kv) > 0.8
    assert int(first.cpu()[0]) == int(np.argmax(lg[-1]))
    # the API-parity handle base_model.lm_head applies the scales too
    h_last = hidden[-1:].contiguous()
    np.testing.assert_allclose(sm.base_model.lm_head(h_last).float().cpu().numpy()[0], lg[-1], rtol=0, atol=2.0 ** -6 * np.abs(lg[-1]).max())


@pytest.mark.parametrize("temperature,seed", [(1.0, 0), (0.7, 3), (6.0, 11)])
def test_sampling_path_matches_oracle_with_shared_randomness(temperature, seed):
    """temperature > 0: sequential-rejection accept (utils.py:453-493) + multinomial next token on the device.  Oracle and device
    draw the same counter-based uniforms, so accept lengths and tokens are compared exactly (parity with the reference's torch RNG
    is distributional; the oracle's branch itself is pinned against the reference in tests/test_oracle_golden.py::g7)."""
    sm, ot, od = build(50, 60, True)
    rng = np.random.default_rng(100 + seed)
    ids = rng.integers(3, T["V"], size=18)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], temperature=temperature, seed=seed, max_new_tokens=32, log=True,
                                               return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=32, max_pos=T["max_pos"], temperature=temperature, seed=seed)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and new_token == o_new
    # sampling really happened (the greedy stream differs) and still accepts several tokens per round on the structured pair
    greedy = sm.specgenerate(torch.from_numpy(ids)[None], temperature=0.0, max_new_tokens=32)
    if temperature <= 1.0:
        assert max(acc) >= 2
    else:  # a hot distribution over a confident (logit gap ~20) synthetic model: the stream leaves the greedy one, rejections happen
        assert not np.array_equal(greedy[0].cpu().numpy()[: len(o_out)], o_out[: greedy.shape[1]])
        assert min(acc) == 0


def test_sampling_acceptance_is_distributionally_sane():
    """Over many seeds the first sampled token follows softmax(logits / T) of the prefill (chi-square-free sanity: the mode is the
    greedy token and its frequency is close to its probability)."""
    sm, ot, od = build(50, 60, True)
    ids = np.random.default_rng(5).integers(3, T["V"], size=12)
    pkv, _, _ = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    logits, _ = ot.forward(pkv, input_ids=ids)
    Tm = 4.0
    p = vo.softmax_T(logits[-1], Tm)
    firsts = []
    for seed in range(60):
        out = sm.specgenerate(torch.from_numpy(ids)[None], temperature=Tm, seed=seed, max_new_tokens=1)
        firsts.append(int(out[0, len(ids)]))
    top = int(np.argmax(p))
    freq = np.mean(np.array(firsts) == top)
    assert abs(freq - p[top]) < 4 * np.sqrt(p[top] * (1 - p[top]) / 60) + 0.02


def test_api_errors_stop_rules_and_return_tuples(golden_dir):
    """The reference's specgenerate contract (spec_model_ours.py:247-266, 363-370, 484-582): argument errors, the three stop
    rules (eos in the generated ids, new_token > max_new_tokens, the max_length round cap) and the flag-dependent return
    tuple — each against the oracle's restatement of the same rule."""
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids = g["succ0_ids"]
    t_ids = torch.from_numpy(ids)[None]
    sm, ot, od = build(50, 60, True)
    with pytest.raises(ValueError):
        sm.specgenerate(None)  # neither ids nor embeds
    with pytest.raises(ValueError):
        sm.specgenerate(t_ids, inputs_embeds=torch.zeros(1, len(ids), T["D"]))
    # return tuples by flags (:555-582)
    r = sm.specgenerate(t_ids, max_new_tokens=12)
    assert torch.is_tensor(r) and r.shape[0] == 1
    r = sm.specgenerate(t_ids, max_new_tokens=12, log=True)
    assert len(r) == 3 and isinstance(r[1], int) and isinstance(r[2], int)
    r = sm.specgenerate(t_ids, max_new_tokens=12, return_acceptance_len=True, return_decode_time=True)
    assert len(r) == 3 and isinstance(r[1], list) and isinstance(r[2], float)
    # new_token > max_new_tokens (:546) for several budgets, incl. 0 (one round always runs)
    for mnt in (0, 1, 7, 23):
        out, new_token, idx, acc = sm.specgenerate(t_ids, max_new_tokens=mnt, log=True, return_acceptance_len=True)
        o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=mnt, max_pos=T["max_pos"])
        np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
        assert (new_token, idx, acc) == (o_new, o_idx, o_acc) and new_token > mnt
    # max_length round cap (:270,:484): max_length - spec_layer.total_tokens (= total_token - 1) - 10 = 4 rounds
    out, new_token, idx, acc = sm.specgenerate(t_ids, max_new_tokens=400, max_length=30 + 10 + 3, log=True, return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=400, max_length=30 + 10 + 3, max_pos=T["max_pos"])
    assert idx == o_idx == 3 and len(acc) == 4 and acc == o_acc
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    # eos among the generated ids (:544): pick a token the stream is known to produce and make it the eos
    full = vo.specgenerate(ot, od, ids, max_new_tokens=40, max_pos=T["max_pos"])[0]
    eos = int(full[len(ids) + 9])
    sm2, ot2, od2 = build(50, 60, True)
    sm2.base_model.cfg.eos_token_id = eos
    sm2 = SpecModel(sm2.base_model, sm2.spec_layer, total_token=30, depth=3, top_k=8, num_q=2)  # engine config carries the eos id
    out, new_token, idx, acc = sm2.specgenerate(t_ids, max_new_tokens=40, log=True, return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot2, od2, ids, max_new_tokens=40, eos_token_id=eos, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert (new_token, idx, acc) == (o_new, o_idx, o_acc) and len(o_out) < len(full) and eos in o_out[len(ids):]


def test_image_feature_count_mismatch_raises():
    """spec_model_ours.py:363-370: number of image placeholder tokens != number of image features -> ValueError."""
    sm, _, _ = build(70, 71, True, arch="LlavaNextForConditionalGeneration")
    ids = np.full(20, 5, np.int64)
    ids[4:12] = IMG_TOK
    feats = torch.zeros(7, T["D"], dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError, match="do not match"):
        sm.specgenerate(torch.from_numpy(ids)[None], pixel_values=feats, max_new_tokens=4)


@pytest.mark.parametrize("total_token", [40, 60, 33])
def test_trees_larger_than_32_nodes(golden_dir, total_token):
    """total_token in (32, 64] (spec_model_ours.py:179-201 autotunes up to 60): the verify pass runs its GEMMs in two 32-row
    passes; token stream and accept lengths equal the oracle's with the same tree size, and greedy AR."""
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids = g["succ1_ids"]
    sm, ot, od = build(51, 61, True, total_token=total_token)
    od.cfg.total_token = total_token
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=36, log=True, return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=36, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert (new_token, idx, acc) == (o_new, o_idx, o_acc) and max(acc) >= 3
    tok, pos, mask, ret = sm.engine.tree()
    assert tok.shape[0] == total_token
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=out.shape[1] - len(ids) - 1)[0].cpu().numpy()
    n = min(len(ar), out.shape[1])
    np.testing.assert_array_equal(ar[:n], out[0].cpu().numpy()[:n])


def test_total_tokens_setter_and_autotune(golden_dir):
    """`model.spec_layer.total_tokens = T - 1` after construction (spec_model_ours.py:201) resizes the tree of later rounds; the
    autotune picks one of the reference's candidates and leaves the model consistent with the oracle at that size."""
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids = g["succ0_ids"]
    sm, ot, od = build(50, 60, True)
    sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=8)  # graphs captured at 30 nodes
    sm.spec_layer.total_tokens = 47
    assert sm.engine.total_token == 48
    od.cfg.total_token = 48
    out, _, _, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=30, log=True, return_acceptance_len=True)
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc
    best = sm.autotune_total_token(iters=3)
    assert best in (40, 48, 50, 56, 60) and sm.spec_layer.total_tokens == best - 1 and sm.engine.total_token == best
    od.cfg.total_token = best
    out, _, _, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=20, log=True, return_acceptance_len=True)
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=20, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc


def test_from_pretrained_real_checkpoint_dirs_with_hf_vision_tower(tmp_path):
    """The whole drop-in surface on checkpoint DIRECTORIES: a (tiny, random) HF LLaVA-NeXT checkpoint written by transformers
    itself + a ViSpec draft dir -> SpecModel.from_pretrained -> specgenerate(input_ids, pixel_values, image_sizes).  The vision
    tower / projector / anyres packing run as HF modules on PyTorch-ROCm; the result equals the same model fed the
    precomputed features, the oracle on the merged embeddings, and greedy AR."""
    import json
    transformers = pytest.importorskip("transformers")
    from safetensors.torch import save_file
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaNextConfig, LlavaNextForConditionalGeneration
    vc = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14)
    tc = LlamaConfig(vocab_size=T["V"], hidden_size=T["D"], intermediate_size=T["I"], num_hidden_layers=T["NL"], num_attention_heads=T["H"],
                     num_key_value_heads=T["H"], rms_norm_eps=1e-5, max_position_embeddings=T["max_pos"])
    cfg = LlavaNextConfig(vision_config=vc, text_config=tc, image_grid_pinpoints=[[28, 56], [56, 28], [56, 56]], image_token_index=IMG_TOK)
    torch.manual_seed(0)
    hf = LlavaNextForConditionalGeneration(cfg).eval()
    # give the language model the structured (successor) weights so that drafts get accepted
    tw = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=70, structured=True)
    dw = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], seed=71, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    lm_sd = {k[len("model."):]: torch.from_numpy(v) for k, v in tw.items() if k.startswith("model.")}
    hf.model.language_model.load_state_dict(lm_sd, strict=False)
    hf.lm_head.weight.data.copy_(torch.from_numpy(tw["lm_head.weight"]))
    tdir, ddir = tmp_path / "target", tmp_path / "draft"
    hf.to(torch.bfloat16).save_pretrained(tdir)
    ddir.mkdir()
    save_file({k: torch.from_numpy(v).to(torch.bfloat16).contiguous() for k, v in dw.items()}, str(ddir / "model.safetensors"))
    json.dump({"hidden_size": T["D"], "num_attention_heads": T["H"], "intermediate_size": T["I"], "vocab_size": T["V"],
               "max_position_embeddings": T["max_pos"]}, open(ddir / "config.json", "w"))
    sm = SpecModel.from_pretrained(base_model_path=str(tdir), spec_model_path=str(ddir), total_token=30, depth=3, top_k=8, num_q=2)
    assert hasattr(sm.base_model.vision, "tower")
    rng = np.random.default_rng(5)
    pv = torch.randn(1, 5, 3, 28, 28, generator=torch.Generator().manual_seed(3))
    sizes = torch.tensor([[50, 30]])
    feats = sm.base_model.get_image_features(pv, sizes)  # HF modules, bf16, on the GPU
    n_img = feats.shape[0]
    ids = np.concatenate([rng.integers(3, 900, 6), np.full(n_img, IMG_TOK), rng.integers(3, 900, 9)])
    t_ids = torch.from_numpy(ids)[None]
    out, new_token, idx, acc = sm.specgenerate(t_ids, pixel_values=pv, image_sizes=sizes, max_new_tokens=24, log=True, return_acceptance_len=True)
    out = out[0].cpu().numpy()
    # (1) == the same model built from in-memory weights and fed the features directly
    sm2, ot, od = build(70, 71, True, arch="LlavaNextForConditionalGeneration")
    out2, _, _, acc2 = sm2.specgenerate(t_ids, pixel_values=feats, max_new_tokens=24, log=True, return_acceptance_len=True)
    np.testing.assert_array_equal(out, out2[0].cpu().numpy())
    assert acc == acc2 and max(acc) >= 2
    # (2) == the oracle on the merged embeddings
    emb = synth.bf16_grid(tw["model.embed_tokens.weight"])[np.where(ids == IMG_TOK, 0, ids)]
    mask = ids == IMG_TOK
    emb[mask] = feats.float().cpu().numpy()
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=24, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[len(ids):], o_out[len(ids):])
    assert acc == o_acc
    # (3) token-count mismatch is reported like the reference (:363-370)
    with pytest.raises(ValueError, match="do not match"):
        sm.specgenerate(torch.from_numpy(ids[:-1 - 9])[None], pixel_values=pv, image_sizes=sizes, max_new_tokens=4)


def test_evaluation_harness_jsonl_and_speed(tmp_path, golden_dir):
    """evaluation harness (gen_spec_answer_coco_caption.py:160-285, speed.py:56-97): JSONL record fields, one line per request,
    speed-up = mean tokens/s (spec) / mean tokens/s (AR), tau = mean accept length; spec and AR answers carry the same tokens."""
    import json
    from vispec_amd.evaluation.harness import get_model_answers, speed
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    sm, _, _ = build(50, 60, True)
    reqs = [(f"q{i}", torch.from_numpy(g[f"succ{i % 2}_ids"])[None].cuda(), {}) for i in range(3)]
    fs, fb = str(tmp_path / "spec.jsonl"), str(tmp_path / "ar.jsonl")
    get_model_answers(sm, reqs, fs, max_new_tokens=24, warmup=1)
    get_model_answers(sm, reqs, fb, max_new_tokens=24, warmup=1, baseline=True)
    recs = [json.loads(l) for l in open(fs)]
    base = [json.loads(l) for l in open(fb)]
    assert [r["question_id"] for r in recs] == ["q0", "q1", "q2"] and len(base) == 3
    for r, b in zip(recs, base):
        c, cb = r["choices"][0], b["choices"][0]
        assert set(c) >= {"index", "turns", "idxs", "new_tokens", "wall_time", "acceptance_length"}
        assert c["new_tokens"][0] > 24 and len(c["acceptance_length"]) == c["idxs"][0] + 1 and c["wall_time"][0] > 0
        n = min(len(c["turns"][0]), len(cb["turns"][0]))
        assert c["turns"][0][:n] == cb["turns"][0][:n]  # greedy invariance through the harness
    s = speed(fs, fb)
    assert s["speedup"] > 0 and 0 < s["tau"] <= 4 and s["spec_tokens_per_s"] > 0


@pytest.mark.parametrize("temperature,top_k,seed", [(6.0, 4, 1), (3.0, 1, 2), (8.0, 20, 5), (1.0, 50, 7)])
def test_sampling_with_top_k_warper(temperature, top_k, seed):
    """prepare_logits_processor(temperature, top_k) (utils.py:39-55): TopKLogitsWarper after the temperature on the first token,
    on every verified distribution and on the bonus token — device == oracle with shared uniforms (the oracle's warped branch is
    pinned against the reference in g7).  top_k = 1 degenerates to greedy; 0 < top_p < 1 is refused (the reference raises too)."""
    sm, ot, od = build(50, 60, True)
    ids = np.random.default_rng(300 + seed).integers(3, T["V"], size=16)
    t_ids = torch.from_numpy(ids)[None]
    out, new_token, idx, acc = sm.specgenerate(t_ids, temperature=temperature, top_k=top_k, seed=seed, max_new_tokens=28, log=True,
                                               return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=28, max_pos=T["max_pos"], temperature=temperature, seed=seed,
                                                 top_k=top_k)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and new_token == o_new
    greedy = sm.specgenerate(t_ids, temperature=0.0, max_new_tokens=28)[0].cpu().numpy()
    n = min(len(greedy), len(o_out))
    if top_k == 1:
        np.testing.assert_array_equal(o_out[:n], greedy[:n])  # only the arg-max survives the warper
    elif temperature >= 6.0:
        unwarped = vo.specgenerate(ot, od, ids, max_new_tokens=28, max_pos=T["max_pos"], temperature=temperature, seed=seed)[0]
        assert not np.array_equal(unwarped[: len(o_out)], o_out[: len(unwarped)])  # the warper changes the hot-temperature stream
    with pytest.raises(NotImplementedError):
        sm.specgenerate(t_ids, temperature=1.0, top_p=0.9, max_new_tokens=4)


def test_kv_capacity_stop_instead_of_overflow(golden_dir):
    """Maximum sizes: a request whose generation would run past the KV cache stops (done bit 2) with everything generated so far
    intact — the reference's KVCache.cat raises at this point (kv_cache.py:40-58).  The tokens are the oracle's / greedy AR's prefix,
    and a prompt that cannot fit at all is refused up front."""
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    base = g["succ0_ids"]
    ids = np.concatenate([base, np.random.default_rng(9).integers(3, 900, 330 - len(base))])  # max_pos = 512 -> guard at 512 - 128
    sm, ot, od = build(50, 60, True)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=400, log=True, return_acceptance_len=True)
    st = sm.engine.state()
    assert st["done"] & 4 and not st["done"] & 3 and new_token < 400
    # the target guard (one maximal tree + 64 rows) comes first; the stop comes within one round of it
    assert T["max_pos"] - 128 < st["n_ctx"] <= T["max_pos"] - 128 + 24
    out = out[0].cpu().numpy()
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=len(out) - len(ids) - 1, max_pos=T["max_pos"])
    n = min(len(out), len(o_out))
    np.testing.assert_array_equal(out[:n], o_out[:n])
    assert acc[: len(o_acc)] == o_acc[: len(acc)]
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=400)[0].cpu().numpy()
    assert sm.engine.state()["done"] & 4 and len(ar) + 1 < T["max_pos"]
    n = min(len(ar), len(out))
    np.testing.assert_array_equal(ar[:n], out[:n])
    with pytest.raises(RuntimeError, match="does not fit"):
        sm.specgenerate(torch.from_numpy(np.full(T["max_pos"] - 20, 5))[None], max_new_tokens=4)


def test_is_llama3_second_stop_token(golden_dir):
    """is_llama3=True (spec_model_ours.py:268-269, 540-542): "<|eot_id|>" among the generated ids ends the request like eos."""
    from types import SimpleNamespace
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids = g["succ1_ids"]
    sm, ot, od = build(51, 61, True)
    full = vo.specgenerate(ot, od, ids, max_new_tokens=40, max_pos=T["max_pos"])[0]
    eot = int(full[len(ids) + 13])
    sm.tokenizer = SimpleNamespace(eos_token_id=2, convert_tokens_to_ids=lambda t: eot if t == "<|eot_id|>" else -1)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=40, is_llama3=True, log=True, return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=40, max_pos=T["max_pos"], stop_token_id=eot)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert (new_token, idx, acc) == (o_new, o_idx, o_acc) and len(o_out) < len(full) and eot in o_out[len(ids):]
    # without the flag the same token does not stop the request
    out2 = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=40)
    np.testing.assert_array_equal(out2[0].cpu().numpy(), full)


@pytest.mark.parametrize("temperature,top_k", [(0.0, 0), (1.0, 0), (5.0, 6)])
def test_stagewise_helpers_reproduce_the_reference_loop_body(golden_dir, temperature, top_k):
    """The reference's loop written against vispec.model.utils (initialize_tree -> [tree_decoding -> evaluate_posterior ->
    update_inference_inputs]*, spec_model_ours.py:455-547) runs unchanged on vispec_amd.model.utils and yields the tokens of
    the fused specgenerate (and of the oracle), greedy and sampling."""
    from vispec_amd.model import utils as U
    from vispec_amd.model.kv_cache import initialize_past_key_values
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids = g["succ0_ids"]
    sm, ot, od = build(50, 60, True)
    want, _, _, want_acc = sm.specgenerate(torch.from_numpy(ids)[None], temperature=temperature, top_k=top_k, seed=3, max_new_tokens=24, log=True,
                                           return_acceptance_len=True)
    o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=24, max_pos=T["max_pos"], temperature=temperature, seed=3, top_k=top_k)
    np.testing.assert_array_equal(want[0].cpu().numpy(), o_out)
    # ---- the reference's loop body, verbatim in structure
    lp = U.prepare_logits_processor(temperature=temperature, top_k=top_k, seed=3) if temperature > 1e-5 else None
    input_ids = torch.from_numpy(ids)[None].cuda()
    sm.spec_layer.reset_kv()
    pkv, pkv_data, cur = initialize_past_key_values(sm.base_model)
    input_len = input_ids.shape[1]
    U.reset_tree_mode(sm)
    draft_tokens, retrieve_indices, tree_mask, tree_position_ids, logits, hidden_state, sample_token = U.initialize_tree(input_ids, sm, pkv, lp)
    new_token, acc = 0, []
    for idx in range(40):
        sm.base_model.model.tree_mask = tree_mask
        draft_tokens = draft_tokens.to(input_ids.device)
        logits, hidden_state_new, outputs = U.tree_decoding(sm, draft_tokens, pkv, tree_position_ids, input_ids, retrieve_indices)
        padding = torch.zeros(1, 1, dtype=torch.long, device=input_ids.device) - 1
        dt = torch.cat((draft_tokens, padding), dim=1)
        candidates = dt[0, retrieve_indices]
        best_candidate, accept_length, sample_p = U.evaluate_posterior(logits, candidates, lp, model=sm)
        acc.append(int(accept_length))
        input_ids, draft_tokens, retrieve_indices, tree_mask, tree_position_ids, new_token, hidden_state, sample_token = U.update_inference_inputs(
            input_ids, candidates, best_candidate, accept_length, retrieve_indices, lp, new_token, pkv_data, cur, sm, hidden_state_new, sample_p)
        if sm.tokenizer.eos_token_id in input_ids[0, input_len:].tolist() or new_token > 24:
            break
    np.testing.assert_array_equal(input_ids[0].cpu().numpy(), want[0].cpu().numpy())
    assert acc == want_acc == o_acc and int(cur[0]) == input_ids.shape[1]


def test_concurrent_lanes_give_the_sequential_results(golden_dir):
    """Two lanes (own ctx, KV cache, host thread and HIP stream; shared packed weights) running different requests at the same time
    return exactly what each request returns alone — the library keeps no cross-request state outside its ctx."""
    import threading
    from vispec_amd.model.cnets_ours import Model
    from vispec_amd.model.target import TargetLM
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    sm0, _, _ = build(50, 60, True)
    sm1 = SpecModel(TargetLM(sm0.base_model.cfg, sm0.base_model.w), Model(sm0.spec_layer.config, sm0.spec_layer.w, total_tokens=30, depth=3, top_k=8, num_q=2),
                    total_token=30, depth=3, top_k=8, num_q=2)  # second lane on the same weights
    reqs = [torch.from_numpy(g[f"succ{i % 2}_ids"][: 30 - 3 * i])[None].cuda() for i in range(6)]
    alone = [sm0.specgenerate(r, max_new_tokens=30, log=True, return_acceptance_len=True) for r in reqs]
    out = [None] * len(reqs)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def lane(sm, k):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[k]):
            for i in range(k, len(reqs), 2):
                out[i] = sm.specgenerate(reqs[i], max_new_tokens=30, log=True, return_acceptance_len=True)
            streams[k].synchronize()

    th = [threading.Thread(target=lane, args=(sm, k)) for k, sm in enumerate((sm0, sm1))]
    [t.start() for t in th]
    [t.join() for t in th]
    for a, b in zip(alone, out):
        np.testing.assert_array_equal(a[0].cpu().numpy(), b[0].cpu().numpy())
        assert a[1:] == b[1:]
    assert sm1.engine.graph_stats()["replays"] > 0  # the lanes ran on capturable streams: rounds were replayed as hipGraphs


def test_qwen2_text_target_with_qkv_bias_and_gqa():
    """Qwen2ForCausalLM targets (spec_model_ours.py:124-125, modeling_qwen2_kv.py): the Llama decoder with q/k/v bias, GQA and eager
    scores — text-only loop == oracle == greedy AR."""
    Q = synth.QWEN_TINY
    tw = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=93, structured=True, qkv_bias=True, H_kv=Q["Hkv"])
    dw = synth.make_draft_weights(Q["D"], Q["H"], Q["I"], Q["V"], seed=94, structured=True, qkv_bias=True,
                                  target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    tcfg = TargetConfig(hidden_size=Q["D"], num_heads=Q["H"], num_kv_heads=Q["Hkv"], intermediate_size=Q["I"], vocab_size=Q["V"], num_layers=Q["NL"],
                        max_position_embeddings=Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True,
                        architectures=("Qwen2ForCausalLM",))
    dcfg = DraftConfig(hidden_size=Q["D"], num_heads=Q["H"], intermediate_size=Q["I"], vocab_size=Q["V"], max_position_embeddings=Q["max_pos"],
                       rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw)
    ot = vo.TargetLlama(vo.TargetConfig(Q["D"], Q["H"], Q["Hkv"], Q["I"], Q["V"], Q["NL"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"]),
                        tw, bf16=True)
    od = vo.DraftModel(vo.DraftConfig(Q["D"], Q["H"], Q["I"], Q["V"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"]), dw, bf16=True)
    ids = np.random.default_rng(23).integers(3, Q["V"] - 2, size=21)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=28, log=True, return_acceptance_len=True)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=28, max_pos=Q["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert (new_token, idx, acc) == (o_new, o_idx, o_acc) and max(acc) >= 2
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=20)[0].cpu().numpy()
    n = min(len(ar), len(o_out))
    np.testing.assert_array_equal(ar[:n], o_out[:n])


def test_qwen25vl_video_prompt_matches_oracle():
    """Qwen2.5-VL VIDEO input (spec_model_ours.py:421-453 + get_rope_index, modeling_qwen2_5_vl_kv.py:1789-1975): a prompt with one image
    run and one 3-frame video run.  The reference keeps the LAST mask it built — the video's — as the draft's image mask (the image
    run is then an ordinary row block for the draft), and the prefill positions take the video's temporal index
    floor(frame * second_per_grid_ts * tokens_per_second); rope_delta shifts every decode position.  HIP loop == oracle loop == AR."""
    Q = synth.QWEN_TINY
    IMG, VID = Q["V"] - 1, Q["V"] - 2
    tw = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=90, structured=True, qkv_bias=True, H_kv=Q["Hkv"])
    dw = synth.make_draft_weights(Q["D"], Q["H"], Q["I"], Q["V"], seed=93, structured=True, qkv_bias=True,
                                  target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    tcfg = TargetConfig(hidden_size=Q["D"], num_heads=Q["H"], num_kv_heads=Q["Hkv"], intermediate_size=Q["I"], vocab_size=Q["V"], num_layers=Q["NL"],
                        max_position_embeddings=Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True,
                        architectures=("Qwen2_5_VLForConditionalGeneration",), image_token_index=IMG, video_token_id=VID, attn_impl="sdpa",
                        mrope_section=Q["mrope_section"], tokens_per_second=2.0)
    dcfg = DraftConfig(hidden_size=Q["D"], num_heads=Q["H"], intermediate_size=Q["I"], vocab_size=Q["V"], max_position_embeddings=Q["max_pos"],
                       rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw)
    ot = vo.TargetLlama(vo.TargetConfig(Q["D"], Q["H"], Q["Hkv"], Q["I"], Q["V"], Q["NL"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"],
                                        attn_impl="sdpa", mrope_section=Q["mrope_section"]), tw, bf16=True)
    od = vo.DraftModel(vo.DraftConfig(Q["D"], Q["H"], Q["I"], Q["V"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"]), dw, bf16=True)
    rng = np.random.default_rng(23)
    igrid, vgrid, sec = [(1, 4, 4)], [(3, 4, 4)], [1.5]  # 4 image tokens; 3 frames x 4 tokens at temporal index 0, 3, 6
    ids = np.concatenate([rng.integers(3, VID, 4), np.full(4, IMG), rng.integers(3, VID, 3), np.full(12, VID), rng.integers(3, VID, 6)])
    ifeat = synth.bf16_grid(rng.standard_normal((4, Q["D"]), dtype=np.float32) * 0.05)
    vfeat = synth.bf16_grid(rng.standard_normal((12, Q["D"]), dtype=np.float32) * 0.05)
    tb = lambda a: torch.from_numpy(a).to(torch.bfloat16).cuda()
    kw = dict(pixel_values=tb(ifeat), image_grid_thw=torch.tensor(igrid), pixel_values_videos=tb(vfeat), video_grid_thw=torch.tensor(vgrid),
              second_per_grid_ts=torch.tensor(sec))
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=24, log=True, return_acceptance_len=True, **kw)
    pos3, delta = synth.qwen_rope_index(ids, IMG, igrid, video_token_id=VID, video_grids=vgrid, second_per_grid_ts=sec, tokens_per_second=2.0)
    assert pos3[0, 11:23].tolist() == [p + int(pos3[0, 11]) for p in (0, 0, 0, 0, 3, 3, 3, 3, 6, 6, 6, 6)] and delta < 0
    assert sm._rope_delta == delta
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    emb[ids == IMG] = ifeat
    emb[ids == VID] = vfeat
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=(ids == VID), max_new_tokens=24, max_pos=Q["max_pos"],
                                                 position_ids=pos3, rope_delta=delta)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and max(acc) >= 3
    st = sm.engine.state()
    assert st["draft_len"] == st["n_ctx"] - 12 + (sm.engine.num_q - 1)  # only the video run is compressed
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=20, **kw)
    n = min(ar.shape[1], len(o_out))
    np.testing.assert_array_equal(ar[0, :n].cpu().numpy(), o_out[:n])
    with pytest.raises(ValueError, match="Video features and video tokens do not match"):
        sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=4, pixel_values_videos=tb(vfeat[:8]), video_grid_thw=torch.tensor(vgrid))
