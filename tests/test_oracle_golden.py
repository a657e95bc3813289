"""CPU: the numpy oracle (oracle/vispec_oracle.py) against golden vectors captured from the reference itself
(tests/golden/gen_golden.py).  fp32 floats: 2e-4 abs on O(1) activations / logits; integer outputs exact."""
import os

import numpy as np
import pytest

from helpers import T, oracle_draft, oracle_target, vo

ATOL = 2e-4


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def close(a, b, atol=ATOL, rtol=1e-4):
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_bf16_round_is_rne():
    x = np.array([1.0, 1.00390625, 1.0 + 2 ** -8, 1.0 + 3 * 2 ** -8, -3.3895314e38, 0.1], np.float32)
    import torch
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(vo.bf16_round(x), want)


def test_g1_imgadaptor(golden_dir):
    g = load(golden_dir, "g1_imgadaptor.npz")
    for q in (2, 5):
        d, _ = oracle_draft(num_q=q, seed=11)
        close(d.imgadaptor(g[f"x_q{q}"][0]), g[f"y_q{q}"])


@pytest.mark.parametrize("tag,q", [("img_q2", 2), ("img_q5", 5), ("txt", 2)])
def test_g2_prefill(golden_dir, tag, q):
    g = load(golden_dir, "g2_prefill.npz")
    d, _ = oracle_draft(num_q=q, seed=12)
    mask = g[f"{tag}_mask"] if tag != "txt" else None
    out, kv, pos = d.forward_prefill(g[f"{tag}_hidden"], g[f"{tag}_embeds"], mask)
    close(kv[0], g[f"{tag}_k"])
    close(kv[1], g[f"{tag}_v"])
    assert kv[2] == int(g[f"{tag}_real_len"])
    close(d.last_img_hidden, g[f"{tag}_g"])
    ref = g[f"{tag}_out"]  # [L, D] scattered back through trans_mat: compressed row i lives at original row pos[i]...
    close(out[-1], ref[-1])  # the only row topK_genrate consumes (cnets_ours.py:1109)
    if tag == "txt":
        close(out, ref)
    else:
        # text rows keep their index; the (q-1) compressed tokens sit on the last (q-1) image positions
        close(out, ref[pos])
        others = np.setdiff1d(np.arange(ref.shape[0]), pos)
        assert np.all(ref[others] == 0)


def test_g3_decode(golden_dir):
    g = load(golden_dir, "g3_decode.npz")
    d, _ = oracle_draft(num_q=2, seed=13)
    _, kv, _ = d.forward_prefill(g["hidden"], g["embeds"], g["mask"])
    o2, kv2 = d.forward_decode(g["h2"], g["ids2"], kv)
    close(o2, g["o2"])
    L = g["hidden"].shape[0]
    o3, kv3 = d.forward_decode(g["h3"], g["ids3"], kv2, pos=np.full(8, L + 3), tree_mask=np.eye(8, dtype=bool))
    close(o3, g["o3"])
    o4, kv4 = d.forward_decode(g["h4"], g["ids4"], kv3, pos=np.full(8, L + 4), tree_mask=g["tm4"] > 0)
    close(o4, g["o4"])
    close(kv4[0], g["k4"])
    close(kv4[1], g["v4"])
    assert kv4[2] == int(g["real_len4"])


@pytest.mark.parametrize("tag", ["greedy", "sampling"])
def test_g4_topk_genrate(golden_dir, tag):
    g = load(golden_dir, "g4_topk.npz")
    t, _ = oracle_target(seed=20)
    d, _ = oracle_draft(num_q=2, seed=14)
    d.reset_kv()
    r = d.topK_genrate(g["hidden"], g["ids"], t.lm_head, inputs_embeds=g["embeds"], image_mask=g["mask"], sampling=tag == "sampling")
    r2 = d.topK_genrate(g["h2"], g["ids2"], t.lm_head, sampling=tag == "sampling")
    for nm, rr in (("a", r), ("b", r2)):
        np.testing.assert_array_equal(rr[0], g[f"{tag}_{nm}_tokens"])
        np.testing.assert_array_equal(rr[1], g[f"{tag}_{nm}_retrieve"])
        np.testing.assert_array_equal(rr[2], g[f"{tag}_{nm}_mask"] > 0)
        np.testing.assert_array_equal(rr[3], g[f"{tag}_{nm}_pos"])


def test_g14_tree_level_hidden_rows(golden_dir):
    """Every draft forward inside topK_genrate against the reference's own (forward hook): the last row of the prefill / catch-up
    forward and the k hidden rows of each tree level.  Levels >= 1 depend on the evolution of the level mask — a node inherits its
    PARENT'S row (cnets_ours.py:1163-1165) — which the integer outputs pinned by g4 do not feel."""
    g = load(golden_dir, "g14_tree_levels.npz")
    t, _ = oracle_target(seed=20)
    d, _ = oracle_draft(num_q=2, seed=14)
    d.reset_kv()
    for nm, args, kw in (("a", (g["hidden"], g["ids"]), dict(inputs_embeds=g["embeds"], image_mask=g["mask"])), ("b", (g["h2"], g["ids2"]), {})):
        r = d.topK_genrate(args[0], args[1], t.lm_head, **kw)
        np.testing.assert_array_equal(r[0], g[f"{nm}_tokens"])
        np.testing.assert_array_equal(r[2], g[f"{nm}_mask"] > 0)
        assert len(d.level_debug) == 3
        for lvl in range(3):
            close(d.level_debug[lvl]["out"], g[f"{nm}_level{lvl}_out"])


def test_g5_target_verify(golden_dir):
    g = load(golden_dir, "g5_verify.npz")
    t, _ = oracle_target(seed=21)
    pkv, data, cur = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    logits, hidden = t.forward(pkv, input_ids=g["ids"])
    close(logits, g["prefill_logits"])
    close(hidden, g["prefill_hidden"])
    t.tree_mask = g["tree_mask"] > 0
    L = g["ids"].shape[0]
    logits, hidden = t.forward(pkv, input_ids=g["cand"], position_ids=g["tree_pos"] + L)
    close(logits, g["logits"])
    close(hidden, g["hidden"])
    np.testing.assert_array_equal(cur, g["cur"])
    n = L + g["cand"].shape[0]
    close(data[0][0, 0, :, :n], g["k0"])
    close(data[0][3, 0, :, :n], g["v1"])


def test_g6_evaluate_posterior(golden_dir):
    g = load(golden_dir, "g6_posterior.npz")
    seen = set()
    for i in range(int(g["n"])):
        b, a, p = vo.evaluate_posterior_greedy(g[f"logits{i}"], g[f"cand{i}"])
        assert (b, a) == (int(g[f"best{i}"]), int(g[f"acc{i}"]))
        np.testing.assert_array_equal(p, g[f"p{i}"])
        seen.add(a)
    assert 0 in seen and max(seen) >= 2


@pytest.mark.parametrize("case", ["rand0", "rand1", "rand2", "succ0", "succ1"])
def test_g8_whole_loop_text(golden_dir, case):
    g = load(golden_dir, "g8_loop.npz")
    kind, si = case[:4], int(case[4:])
    if kind == "rand":
        t, _ = oracle_target(seed=30 + si)
        d, _ = oracle_draft(seed=40 + si)
        mnt = 24
    else:
        t, tw = oracle_target(seed=50 + si, structured=True)
        d, _ = oracle_draft(seed=60 + si, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
        mnt = 40
    out, new_token, idx, acc = vo.specgenerate(t, d, g[f"{case}_ids"], max_new_tokens=mnt, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out, g[f"{case}_out"])
    assert new_token == int(g[f"{case}_new_token"]) and idx == int(g[f"{case}_idx"])
    np.testing.assert_array_equal(acc, g[f"{case}_acc"])
    # the reference's own invariant: speculative output == greedy AR of the same target
    L = g[f"{case}_ids"].shape[0]
    np.testing.assert_array_equal(out[L:], g[f"{case}_ar"])
    ar = vo.baseline_forward(t, g[f"{case}_ids"], max_steps=len(out) - L, max_pos=T["max_pos"])
    np.testing.assert_array_equal(ar[: len(out)], out)


def test_g8_whole_loop_image(golden_dir):
    g = load(golden_dir, "g8_loop.npz")
    t, tw = oracle_target(seed=70, structured=True)
    d, _ = oracle_draft(seed=71, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    out, new_token, idx, acc = vo.specgenerate(t, d, g["img_ids"], inputs_embeds=g["img_emb"], image_mask=g["img_mask"],
                                               max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out, g["img_out"])
    np.testing.assert_array_equal(acc, g["img_acc"])
    assert max(acc) == 4 and d.stable_kv[0].shape[1] < len(out) - 20  # compressed draft KV is shorter than the context


@pytest.mark.parametrize("case", ["succ0", "succ1", "rand1", "img"])
def test_torch_cpu_backend_reproduces_the_reference_streams(golden_dir, case):
    """oracle/torch_cpu.py (the PyTorch-CPU back end bench.py's cpu_baseline leg times) is pinned like the numpy oracle: the
    reference-captured token streams and accept lengths of g8, through TorchOps, and through timed_request's call sequence."""
    pytest.importorskip("torch")
    from oracle import torch_cpu as tc
    g = load(golden_dir, "g8_loop.npz")
    if case == "img":
        t, tw = oracle_target(seed=70, structured=True)
        d, _ = oracle_draft(seed=71, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
        kw, mnt = dict(inputs_embeds=g["img_emb"], image_mask=g["img_mask"]), 30
    elif case.startswith("rand"):
        t, _ = oracle_target(seed=30 + int(case[4:]))
        d, _ = oracle_draft(seed=40 + int(case[4:]))
        kw, mnt = {}, 24
    else:
        t, tw = oracle_target(seed=50 + int(case[4:]), structured=True)
        d, _ = oracle_draft(seed=60 + int(case[4:]), structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
        kw, mnt = {}, 40
    t.ops, d.ops = tc.TorchOps(), tc.TorchOps()
    out, new_token, idx, acc = vo.specgenerate(t, d, g[f"{case}_ids"], max_new_tokens=mnt, max_pos=T["max_pos"], **kw)
    np.testing.assert_array_equal(out, g[f"{case}_out"])
    np.testing.assert_array_equal(acc, g[f"{case}_acc"])
    # the timed leg's own call sequence: same rounds, same accept lengths, AR continuation == the speculative stream
    rounds = 5
    r = tc.timed_request(t, d, g[f"{case}_ids"], kw.get("inputs_embeds"), kw.get("image_mask"), rounds=rounds, ar_steps=3, max_pos=T["max_pos"])
    assert r["accept_lengths"] == list(g[f"{case}_acc"][:rounds])
    np.testing.assert_array_equal(r["tokens"], g[f"{case}_out"][: len(r["tokens"])])
    assert len(r["verify_s"]) == rounds and len(r["ar_s"]) == 3 and r["context"] == len(r["tokens"])
    # ... and with the target prefill handed over (bench.py passes the GPU's KV rows, hidden states and last logits)
    ids = g[f"{case}_ids"]
    pkv, data, cur = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    t.tree_mask = None
    lg, hid = t.forward(pkv, input_ids=ids) if kw.get("inputs_embeds") is None else t.forward(pkv, inputs_embeds=kw["inputs_embeds"])
    r2 = tc.timed_request(t, d, ids, kw.get("inputs_embeds"), kw.get("image_mask"), rounds=rounds, ar_steps=0, max_pos=T["max_pos"],
                          prefilled=(data[0][:, 0, :, : len(ids)].copy(), hid, lg[-1]))
    assert r2["accept_lengths"] == r["accept_lengths"]
    np.testing.assert_array_equal(r2["tokens"], r["tokens"])


def test_g9_bf16_rounding_points(golden_dir):
    """Oracle in bf16-emulation mode vs the reference run in torch-bf16 on CPU.  Reduction orders differ, so the
    bound is a few bf16 ulps (2^-8 relative), not bitwise."""
    g = load(golden_dir, "g9_bf16.npz")
    d, _ = oracle_draft(num_q=2, seed=15, bf16=True)
    out, kv, _ = d.forward_prefill(g["d_hidden"], g["d_embeds"], g["d_mask"])
    tol = dict(atol=0.02, rtol=0.02)
    np.testing.assert_allclose(kv[0], g["d_k"], **tol)
    np.testing.assert_allclose(out[-1], g["d_out_last"], **tol)
    o2, _ = d.forward_decode(g["d_h2"], g["d_ids2"], kv)
    np.testing.assert_allclose(o2, g["d_o2"], **tol)
    # typical error is far below the bound: most entries agree to 1 ulp
    rel = np.abs(o2 - g["d_o2"]) / (np.abs(g["d_o2"]) + 1e-2)
    assert np.median(rel) < 2 ** -8
    t, _ = oracle_target(seed=22, bf16=True)
    pkv, _, _ = vo.initialize_past_key_values(T["NL"], T["H"], T["max_pos"], T["D"] // T["H"])
    logits, hidden = t.forward(pkv, input_ids=g["t_ids"])
    np.testing.assert_allclose(hidden, g["t_hidden"], **tol)
    np.testing.assert_allclose(logits, g["t_logits"], atol=0.02, rtol=0.03)


def test_g10_qwen_target(golden_dir):
    """Qwen2.5-VL text model of the reference (GQA, q/k/v bias, multimodal rotary with an image block, SDPA, tree verify at
    rope_delta-shifted positions) vs the oracle."""
    from vispec_amd import synth
    Q = synth.QWEN_TINY
    g = load(golden_dir, "g10_qwen.npz")
    w = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=90, qkv_bias=True, H_kv=Q["Hkv"])
    cfg = vo.TargetConfig(Q["D"], Q["H"], Q["Hkv"], Q["I"], Q["V"], Q["NL"], Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"],
                          attn_impl="sdpa", mrope_section=Q["mrope_section"])
    t = vo.TargetLlama(cfg, w)
    pos3, delta = synth.qwen_rope_index(g["ids"], Q["V"] - 1, [(1, 6, 8)])
    np.testing.assert_array_equal(pos3, g["pos3"])
    assert delta == int(g["delta"]) and delta < 0
    pkv, data, cur = vo.initialize_past_key_values(Q["NL"], Q["Hkv"], Q["max_pos"], Q["D"] // Q["H"])
    logits, hidden = t.forward(pkv, inputs_embeds=g["emb"], position_ids=pos3)
    close(hidden, g["prefill_hidden"])
    close(logits, g["prefill_logits"])
    t.tree_mask = g["tree_mask"] > 0
    L = len(g["ids"])
    logits, hidden = t.forward(pkv, input_ids=g["cand"], position_ids=g["tree_pos"] + L + delta)
    close(hidden, g["hidden"])
    close(logits, g["logits"])
    np.testing.assert_array_equal(cur, g["cur"])
    n = L + len(g["cand"])
    close(data[0][0, 0, :, :n], g["k0"])
    close(data[0][3, 0, :, :n], g["v1"])


def test_e4m3_round_matches_torch_float8():
    import torch
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-3, 0.1, 1, 20, 200)] +
                       [np.array([0, 448, -448, 449.9, 463.9, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 0.0009765625, 17.0, 18.0, 19.0], np.float32)])
    want = torch.from_numpy(np.clip(x, -448, 448)).to(torch.float8_e4m3fn).float().numpy()
    got = vo.e4m3_round(x)
    np.testing.assert_array_equal(got, want)
    q, s = vo.quantize_fp8(rng.standard_normal((16, 64)).astype(np.float32))
    assert np.abs(q).max() == 448 and (np.abs(q).max(axis=1) == 448).all() and s.shape == (16,)


def test_g7_evaluate_posterior_sampling(golden_dir):
    """Sampling branch of evaluate_posterior (temperature > 0): same decisions and the same residual distribution as the
    reference when fed the reference's own uniform draws."""
    g = load(golden_dir, "g7_posterior_sampling.npz")
    accs = []
    for i in range(int(g["n"])):
        u = g[f"u{i}"]
        K = int(g[f"K{i}"])
        b, a, p = vo.evaluate_posterior_sampling(g[f"logits{i}"], g[f"cand{i}"], float(g[f"T{i}"]), lambda j, c: u[j, c], top_k=K)
        assert (b, a) == (int(g[f"best{i}"]), int(g[f"acc{i}"])), i
        np.testing.assert_allclose(p, g[f"p{i}"], rtol=2e-5, atol=1e-7)
        if K:  # TopKLogitsWarper: at most K tokens keep probability mass
            assert 0 < (p > 0).sum() <= K
        accs.append((a, K))
    assert 0 in [a for a, _ in accs] and max(a for a, _ in accs) >= 2 and max(a for a, K in accs if K) >= 1 and int(g["n"]) == 36
    # the explicit-uniform multinomial is a proper inverse CDF
    p = np.array([0.1, 0.0, 0.6, 0.3])
    assert [vo.multinomial_inverse_cdf(p, u) for u in (0.0, 0.0999, 0.1, 0.69, 0.7, 0.9999)] == [0, 0, 2, 2, 3, 3]
    assert 0.0 <= vo.uniform_hash(1, 2, 3, 4) < 1.0 and vo.uniform_hash(1, 2, 3, 4) != vo.uniform_hash(1, 2, 3, 5)


def test_g11_update_inference_inputs(golden_dir):
    """utils.update_inference_inputs in isolation: appended ids, KV gather-compaction of every cache tensor, lengths, the hidden
    rows handed to the draft, next token (argmax / multinomial with the recorded uniform), new_token."""
    g = load(golden_dir, "g11_update.npz")
    for i in range(int(g["n"])):
        seen = {}

        class Draft:
            def topK_genrate(self, hidden, ids, head_w, sampling=False):
                seen["hidden"], seen["ids"], seen["sampling"] = hidden, ids, sampling
                return "dt", "ri", "tm", "tp"

        st = vo.LoopState(g[f"ids{i}"].copy(), None, g[f"ri{i}"], None, None, new_token=5)
        data, data2 = g[f"data{i}"].copy(), g[f"data2_{i}"].copy()
        cur = np.zeros(6, np.int64)
        samp = bool(g[f"sampling{i}"])
        tok = vo.update_inference_inputs(st, g[f"cand{i}"], int(g[f"best{i}"]), int(g[f"acc{i}"]), [data, data2], cur, g[f"hid{i}"][0], g[f"sp{i}"],
                                         Draft(), None, sample_u=float(g[f"u{i}"]) if samp else None)
        np.testing.assert_array_equal(st.input_ids, g[f"o_ids{i}"])
        np.testing.assert_array_equal(data, g[f"o_data{i}"])
        np.testing.assert_array_equal(data2, g[f"o_data2_{i}"])
        np.testing.assert_array_equal(cur, g[f"o_cur{i}"])
        assert tok == int(g[f"o_token{i}"]) and st.new_token == int(g[f"o_new_token{i}"])
        np.testing.assert_array_equal(seen["hidden"], g[f"o_hidden{i}"])
        np.testing.assert_array_equal(seen["ids"], g[f"o_draft_ids{i}"])
        assert seen["sampling"] == samp and (st.draft_tokens, st.tree_position_ids) == ("dt", "tp")


def test_g12_kvcache_class_matches_reference(golden_dir):
    """vispec_amd.model.kv_cache.KVCache (the host-side mirror of kv_cache.py:4-66) against the reference's own class on a CPU slab:
    cat returns the [0, len) view and advances the length, copy gathers the accepted rows behind prev_length."""
    torch = pytest.importorskip("torch")
    from vispec_amd.model.kv_cache import KVCache
    g = load(golden_dir, "g12_kvcache.npz")
    data = torch.zeros(1, 2, 16, 4)
    cur = torch.zeros((), dtype=torch.long)
    kv = KVCache(data, cur)
    v1 = kv.cat(torch.from_numpy(g["a"]))
    np.testing.assert_array_equal(v1.numpy(), g["v1"])
    assert tuple(kv.shape) == tuple(g["s1"])
    v2 = kv.cat(torch.from_numpy(g["b"]))
    np.testing.assert_array_equal(v2.numpy(), g["v2"])
    assert tuple(kv.shape) == tuple(g["s2"])
    np.testing.assert_array_equal(data.numpy(), g["d2"])
    kv.copy(torch.from_numpy(g["idx"]), 5)
    np.testing.assert_array_equal(data.numpy(), g["d3"])
    assert int(cur) == int(g["len3"]) and tuple(kv.shape) == tuple(g["s3"])


def test_g13_real_dims_lmhead_topk_and_fusion(golden_dir):
    """The oracle at the REAL LLaVA-7B dims (D=4096, V=32064): LM-head -> log-softmax -> top-k and the draft's input fusion against
    the reference's torch ops; weights re-derived from the seeds the fixture generator used."""
    g = load(golden_dir, "g13_real_dims.npz")
    D, V, k = 4096, 32064, 8
    rng = np.random.default_rng(1300)
    W = rng.standard_normal((V, D), dtype=np.float32) * np.float32(0.02)
    o = vo.Ops(bf16=False)
    logp = o.log_softmax(o.linear(g["h"], W))
    for r in range(logp.shape[0]):
        wv, wi = vo.topk_desc(logp[r], k)
        np.testing.assert_array_equal(wi, g["top_idx"][r])
        np.testing.assert_allclose(wv, g["top_logp"][r], rtol=0, atol=2e-5)
    del W
    rng2 = np.random.default_rng(1301)
    w = {"fc.weight": rng2.standard_normal((D, 2 * D), dtype=np.float32) * np.float32(0.02)}
    w["fc.bias"] = rng2.standard_normal(D, dtype=np.float32) * np.float32(0.02)
    w["img_fc.weight"] = rng2.standard_normal((D, 2 * D), dtype=np.float32) * np.float32(0.02)
    w["img_fc.bias"] = rng2.standard_normal(D, dtype=np.float32) * np.float32(0.02)
    h2 = o.linear(np.concatenate([g["hid"], np.broadcast_to(g["g"], g["hid"].shape)], -1), w["img_fc.weight"], w["img_fc.bias"])
    fused = o.linear(np.concatenate([g["emb"], h2], -1), w["fc.weight"], w["fc.bias"])
    np.testing.assert_allclose(fused, g["fused"], rtol=0, atol=2e-5 * np.abs(g["fused"]).max())


def test_g15_qwen_rope_index_images_and_videos(golden_dir):
    """vispec_amd.synth.qwen_rope_index == the reference's get_rope_index (modeling_qwen2_5_vl_kv.py:1789-1975) on image runs, video
    runs with second_per_grid_ts * tokens_per_second temporal scaling (incl. the default 1.0 and a truncating 0.3 s grid), mixed
    prompts and text only: positions [3, L] and rope_delta, exact."""
    from vispec_amd import synth
    g = np.load(os.path.join(golden_dir, "g15_qwen_rope_index.npz"))
    IMG, VID = int(g["ids_image"]), int(g["ids_video"])
    for ci in range(int(g["n_cases"])):
        ids = g[f"c{ci}_ids"]
        ig = [tuple(int(v) for v in r) for r in g[f"c{ci}_image_grids"]]
        vg = [tuple(int(v) for v in r) for r in g[f"c{ci}_video_grids"]]
        sec = g[f"c{ci}_sec"] if int(g[f"c{ci}_has_sec"]) else None
        pos, delta = synth.qwen_rope_index(ids, IMG, ig, video_token_id=VID, video_grids=vg, second_per_grid_ts=sec,
                                           tokens_per_second=float(g[f"c{ci}_tps"]))
        np.testing.assert_array_equal(pos, g[f"c{ci}_pos"], err_msg=f"case {ci}")
        assert delta == int(g[f"c{ci}_delta"]), ci


@pytest.mark.parametrize("tag", ["two", "three_q3", "image_last", "one_q5"])
def test_g16_multi_image_prefill_against_the_repaired_reference(golden_dir, tag):
    """MULTI-IMAGE draft prefill (BASELINE config 2: Qwen2.5-VL multi-turn with several images).  The published reference crashes on the
    second image run; the fixture comes from the reference's own Model.forward with the two scatter-matrix index
    expressions repaired (`h_s[0]`/`h_s[1]` -> `h_s[-2]`/`h_s[-1]`, tests/golden/gen_golden.py g16): REFERENCE-INTENT, parity unpinned
    upstream.  The oracle's compression — every run compressed to q-1 tokens on its last image positions, the global feature g carried
    from run to run into the text rows that follow — must reproduce its compressed K/V, real_len, final g and the consumed output row."""
    g = load(golden_dir, "g16_multi_image_repaired.npz")
    d, _ = oracle_draft(num_q=int(g[f"{tag}_q"]), seed=16)
    out, kv, pos = d.forward_prefill(g[f"{tag}_hidden"], g[f"{tag}_embeds"], g[f"{tag}_mask"])
    close(kv[0], g[f"{tag}_k"])
    close(kv[1], g[f"{tag}_v"])
    assert kv[2] == int(g[f"{tag}_real_len"])
    close(d.last_img_hidden, g[f"{tag}_g"])
    close(out[-1], g[f"{tag}_out_last"])


def test_g17_whole_loop_on_a_multi_image_prompt(golden_dir):
    """Three image runs through the whole draft-and-verify loop: token stream and per-round accept lengths of the reference's loop body
    with its draft forward repaired (fixture g17, reference-intent: the published code crashes on the second run)."""
    g = load(golden_dir, "g17_multi_image_loop.npz")
    t, tw = oracle_target(seed=70, structured=True)
    d, _ = oracle_draft(seed=71, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    out, new_token, idx, acc = vo.specgenerate(t, d, g["ids"], inputs_embeds=g["emb"], image_mask=g["mask"], max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(out, g["out"])
    np.testing.assert_array_equal(acc, g["acc"])
    assert max(acc) == 4 and d.stable_kv[0].shape[1] < len(out) - 40  # three runs compressed to q-1 rows each
