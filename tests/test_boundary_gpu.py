"""GPU tests of the drop-in boundary's corners: SpecModel.forward serving BOTH of the reference's callers,
the product KVCache on the engine's own buffer, capacity errors raised before anything is written, prompts longer than the
draft's cache / rotary table, zero-round requests, and the hipGraph cache key."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import synth  # noqa: E402
from vispec_amd.model.kv_cache import initialize_past_key_values  # noqa: E402

from test_loop_gpu import IMG_TOK, build  # noqa: E402


def test_forward_serves_prefill_and_the_references_tree_decoding_body():
    """SpecModel.forward (spec_model_ours.py:205-245) is ONE function in the reference: utils.initialize_tree calls it on an empty
    cache (prefill), utils.tree_decoding on a non-empty one with the tree candidates, `position_ids = tree_position_ids + n` and the
    tree mask installed on the target.  The body of the reference's tree_decoding (utils.py:389-412) is restated below and driven
    through `model(...)`; logits / hidden are compared with the oracle's verify forward, the return shapes with the reference's."""
    sm, ot, od = build(21, 13, False)
    eng = sm.engine
    ids = np.random.default_rng(5).integers(3, T["V"], size=19)
    input_ids = torch.from_numpy(ids)[None].cuda()
    pkv, pkv_data, cur = initialize_past_key_values(sm.base_model)
    # -- prefill form (utils.py:280-283)
    outputs, orig, hidden_states = sm(input_ids, past_key_values=pkv, output_orig=True)
    assert outputs is None and orig.dtype == torch.float32 and hidden_states.shape == (1, len(ids), T["D"])
    assert int(cur[0]) == len(ids) == int(pkv[1][1].shape[2])
    o_pkv, o_data, o_cur = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    o_logits, o_hidden = ot.forward(o_pkv, input_ids=ids)
    np.testing.assert_allclose(orig[0, -1].cpu().numpy(), o_logits[-1], rtol=0, atol=2.0 ** -6 * np.abs(o_logits).max())
    token = int(torch.argmax(orig[0, -1]))
    assert token == int(np.argmax(o_logits[-1]))
    # -- a tree over T nodes built by the oracle's draft from the oracle's hidden states (any valid tree will do)
    dt, ri, tm, tp = od.topK_genrate(o_hidden, np.concatenate([ids, [token]]), ot.lm_head)
    # -- the reference's tree_decoding body, verbatim in structure (utils.py:397-411)
    sm.base_model.model.tree_mask = torch.from_numpy(tm.astype(np.float32))[None, None]  # spec_model_ours.py:486-489
    tree_candidates = torch.from_numpy(dt)[None].cuda()
    tree_position_ids = torch.from_numpy(tp).cuda()
    retrieve_indices = torch.from_numpy(ri)
    position_ids = tree_position_ids + input_ids.shape[1]
    outputs, tree_logits, hidden_state = sm(tree_candidates, output_orig=True, past_key_values=pkv, position_ids=position_ids)
    logits = tree_logits[0, retrieve_indices]
    Tn = eng.total_token
    assert outputs is None and tree_logits.shape == (1, Tn, T["V"]) and tree_logits.dtype == torch.float32
    assert hidden_state.shape == (1, Tn, T["D"]) and logits.shape == (ri.shape[0], ri.shape[1], T["V"])
    assert int(cur[0]) == len(ids) + Tn  # KVCache.cat appended the T rows (kv_cache.py:40-58)
    ot.tree_mask = tm
    w_logits, w_hidden = ot.forward(o_pkv, input_ids=dt, position_ids=tp + len(ids))
    np.testing.assert_allclose(tree_logits[0].cpu().numpy(), w_logits, rtol=0, atol=2.0 ** -6 * np.abs(w_logits).max())
    np.testing.assert_allclose(hidden_state[0].float().cpu().numpy(), w_hidden, rtol=0, atol=2.0 ** -6 * np.abs(w_hidden).max())
    # a second call with the lengths rolled back (what update_inference_inputs does) reproduces the same numbers: the device context
    # length, not the host mirror, is authoritative
    cur.fill_(len(ids))
    _, again, _ = sm(tree_candidates, output_orig=True, past_key_values=pkv, position_ids=position_ids)
    assert torch.equal(again, tree_logits)
    # -- without a tree mask / position_ids a non-empty cache means a plain causal continuation of S tokens
    cur.fill_(len(ids))
    sm.base_model.model.tree_mask = None
    cont = np.random.default_rng(6).integers(3, T["V"], size=5)
    _, c_logits, c_hidden = sm(torch.from_numpy(cont)[None].cuda(), output_orig=True, past_key_values=pkv)
    assert c_logits.shape == (1, 5, T["V"]) and eng.total_token == Tn  # the tree size is restored
    o_pkv2, _, _ = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    ot.tree_mask = None
    ot.forward(o_pkv2, input_ids=ids)
    w2, _ = ot.forward(o_pkv2, input_ids=cont)
    np.testing.assert_allclose(c_logits[0].cpu().numpy(), w2, rtol=0, atol=2.0 ** -6 * np.abs(w2).max())
    with pytest.raises(ValueError):  # positions that are not tree depths over the context
        sm(tree_candidates, output_orig=True, past_key_values=pkv, position_ids=tree_position_ids + 3)


def test_product_kvcache_cat_and_copy_on_the_engine_buffer(golden_dir):
    """vispec_amd.model.kv_cache.KVCache.cat / .copy (kv_cache.py:38-62) on the tensor the kernels use: same moves as the reference
    fixture G12 (tests/test_oracle_golden.py checks the class on a CPU slab), and the rows land where vispec_tree_attention reads."""
    sm, _, _ = build(21, 13, False)
    pkv, pkv_data, cur = initialize_past_key_values(sm.base_model)
    assert pkv_data[0].data_ptr() == sm.engine.target_kv.data_ptr() and len(pkv) == T["NL"] and cur.device.type == "cpu"
    kv = pkv[1][0]  # K of layer 1
    H, hd = T["H"], T["D"] // T["H"]
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1, H, 5, hd, generator=g).to(torch.bfloat16).cuda()
    b = torch.randn(1, H, 6, hd, generator=g).to(torch.bfloat16).cuda()
    v1 = kv.cat(a)
    assert tuple(kv.shape) == (1, H, 5, hd) and torch.equal(v1, a)
    v2 = kv.cat(b)
    assert tuple(kv.shape) == (1, H, 11, hd) and torch.equal(v2, torch.cat([a, b], 2))
    assert torch.equal(sm.engine.target_kv[2, :, :, :11], v2) and int(cur[2]) == 11 and int(cur[3]) == 0
    idx = torch.tensor([7, 9, 10]).cuda()
    want = v2.index_select(2, idx)
    kv.copy(idx, 5)
    assert int(cur[2]) == 8 and torch.equal(sm.engine.target_kv[2, :, :, 5:8], want)
    assert not sm.engine.target_kv[3].any() and not sm.engine.target_kv[0].any()  # the neighbouring slabs are untouched


def test_capacity_errors_are_raised_before_anything_is_written():
    """A prompt that cannot fit the target cache is refused before the prefill writes a row (the reference fails cleanly in
    KVCache.cat): every slab keeps its sentinel, the neighbouring draft cache too."""
    sm, _, _ = build(21, 13, False)
    eng = sm.engine
    eng.target_kv.fill_(1.5)
    eng.draft_kv.fill_(2.5)
    ids = torch.from_numpy(np.full(T["max_pos"] - 20, 5))[None]
    with pytest.raises(RuntimeError, match="does not fit"):
        sm.specgenerate(ids, max_new_tokens=4)
    with pytest.raises(RuntimeError, match="does not fit"):
        sm.baseline_generate(ids, max_new_tokens=4)
    torch.cuda.synchronize()
    assert bool((eng.target_kv == 1.5).all()) and bool((eng.draft_kv == 2.5).all())
    # the C entry point refuses rows beyond the cache when the offset is host-known
    import ctypes as C
    from vispec_amd import lib as L
    qkv = torch.zeros(8, 3 * T["D"], dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(T["H"], 4, 128, dtype=torch.bfloat16, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = eng.lib.vispec_rope_append(eng.h, eng._stream(), p(qkv), 8, T["H"], T["H"], 128, p(eng.t_cos), p(eng.t_sin), None, None, p(kc), p(kc), 4, None)
    assert rc != 0 and b"do not fit" in eng.lib.vispec_last_error()


def test_prompt_longer_than_the_draft_cache_and_rotary_positions_past_it():
    """The draft's KV holds the COMPRESSED prompt, its rotary positions are the uncompressed ones (cnets_ours.py:845-868): a prompt
    longer than draft_max_pos is fine when its compressed form fits, and generation continues past position draft_max_pos (the
    rotary tables cover the target cache).  Tokens == oracle.  A prompt whose compressed form does not fit is refused."""
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration", kv_max_pos=512, draft_max_pos=192)
    assert sm.engine.d_cos.shape[0] == 512 and sm.engine.draft_kv.shape[2] == 192
    rng = np.random.default_rng(44)
    n_img = 215
    ids = np.concatenate([rng.integers(3, IMG_TOK, 9), np.full(n_img, IMG_TOK), rng.integers(3, IMG_TOK, 16)])  # L = 240 > 192
    feats = synth.bf16_grid(rng.standard_normal((n_img, T["D"]), dtype=np.float32) * 0.05)
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(),
                                               max_new_tokens=40, log=True, return_acceptance_len=True)
    st = sm.engine.state()
    assert st["n_ctx"] > 240 + 30 and st["draft_len"] == st["n_ctx"] - n_img + 1 and not st["done"] & 4
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    mask = ids == IMG_TOK
    emb[mask] = feats
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=40, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc
    txt = torch.from_numpy(rng.integers(3, IMG_TOK, 180))[None]  # text-only: nothing to compress, 180 + first tree > 192 rows
    with pytest.raises(RuntimeError, match="does not fit the draft KV"):
        sm.specgenerate(txt, max_new_tokens=4)


def test_zero_round_request_returns_the_prompt():
    """max_length <= total_tokens + 10 runs no round (spec_model_ours.py:270,484): the reference returns input_ids."""
    sm, _, _ = build(50, 60, True)
    ids = np.random.default_rng(8).integers(3, T["V"], size=14)
    out, new_token, idx = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=8, max_length=sm.spec_layer.total_tokens + 10, log=True)
    np.testing.assert_array_equal(out[0].cpu().numpy(), ids)
    assert (new_token, idx) == (0, 0)


def test_kv_full_is_reported_not_silent(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    base = g["succ0_ids"]
    ids = np.concatenate([base, np.random.default_rng(9).integers(3, 900, 330 - len(base))])
    sm, _, _ = build(50, 60, True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=400)
    assert any("would not fit a KV cache" in str(x.message) for x in w)


def test_graph_cache_key_separates_sampling_configurations():
    """A captured round bakes temperature / seed / top_k / forced_accept in as kernel arguments: the cache key compares the whole
    tuple, so switching any of them on one ctx re-captures instead of replaying the wrong graph (results == oracle each time)."""
    sm, ot, od = build(50, 60, True)
    ids = np.random.default_rng(12).integers(3, T["V"], size=15)
    t_ids = torch.from_numpy(ids)[None]
    s = torch.cuda.Stream()
    cfgs = [dict(temperature=6.0, top_k=4, seed=1), dict(temperature=6.0, top_k=20, seed=1), dict(temperature=6.0, top_k=20, seed=2),
            dict(temperature=3.0, top_k=20, seed=2), dict(temperature=0.0, top_k=0, seed=0), dict(temperature=6.0, top_k=4, seed=1)]
    with torch.cuda.stream(s):
        for c in cfgs:
            out, _, _, acc = sm.specgenerate(t_ids, max_new_tokens=20, log=True, return_acceptance_len=True, **c)
            s.synchronize()
            o_out, _, _, o_acc = vo.specgenerate(ot, od, ids, max_new_tokens=20, max_pos=T["max_pos"], temperature=c["temperature"],
                                                 seed=c["seed"], top_k=c["top_k"])
            np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
            assert acc == o_acc
    gs = sm.engine.graph_stats()
    assert gs["replays"] > 0 and gs["captures"] >= len(cfgs)
