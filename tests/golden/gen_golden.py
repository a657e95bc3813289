"""
Golden-vector generator.  Runs ONLY in the build container (needs /root/reference); the fixtures it
writes (tests/golden/*.npz) are committed and are what travels to the GPU box.

It imports the reference's own modules read-only (no reference source is copied into this repo) with the three
shims of SURVEY.md §8(c): (1) transformers-5.x config normalisation (`rope_scaling=None`, `rope_theta`),
(2) `SpecModel.__new__` constructor bypass (no tokenizer / hub access), (3) pre-seeded CPU KVCache so
`specgenerate` never reaches the CUDA assert.  Weights are NOT stored: they are re-created bit-identically from
numpy PCG64 seeds by vispec_amd/synth.py; only inputs that are not seed-derived and the reference's outputs are saved.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
from torch import nn  # noqa: E402
from transformers import LlamaConfig  # noqa: E402
from vispec.model import cnets_ours, modeling_llama_kv, utils  # noqa: E402
from vispec.model.configs import EConfig  # noqa: E402
from vispec.model.kv_cache import KVCache  # noqa: E402
from vispec.model.spec_model_ours import SpecModel  # noqa: E402

from vispec_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T = synth.TINY
torch.set_grad_enabled(False)
torch.set_num_threads(4)


def t(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def build_draft(num_q=2, seed=1, structured=False, target_embed=None, dtype=torch.float32, rho=0.115):
    ec = EConfig(vocab_size=T["V"], hidden_size=T["D"], intermediate_size=T["I"], num_hidden_layers=1,
                 num_attention_heads=T["H"], num_key_value_heads=T["H"], max_position_embeddings=T["max_pos"],
                 rms_norm_eps=1e-5, pad_token_id=0)
    ec.rope_scaling = None  # shim 1
    ec.rope_theta = 10000.0
    m = cnets_ours.Model(ec, bias=True, total_tokens=30, depth=3, top_k=8, num_q=num_q).eval()
    w = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], num_q=num_q, seed=seed, structured=structured,
                                 target_embed=target_embed, rho=rho)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in w.items()}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    m = m.to(dtype)
    m.diff_device = False
    m.init_tree()
    return m, w


def build_target(seed=0, structured=False, dtype=torch.float32):
    tc = LlamaConfig(vocab_size=T["V"], hidden_size=T["D"], intermediate_size=T["I"], num_hidden_layers=T["NL"],
                     num_attention_heads=T["H"], num_key_value_heads=T["H"], max_position_embeddings=T["max_pos"],
                     rms_norm_eps=1e-5, pad_token_id=0, hidden_act="silu")
    tc.rope_scaling = None
    tc.rope_theta = 10000.0
    tc.pretraining_tp = 1
    tc.architectures = ["LlamaForCausalLM"]
    base = modeling_llama_kv.LlamaForCausalLM(tc).eval()
    w = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=seed, structured=structured)
    sd = {k: t(v) for k, v in w.items()}
    missing, unexpected = base.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary_emb" in k for k in missing), missing
    return base.to(dtype), w


def make_kv(base):
    c = base.config
    NL = c.num_hidden_layers
    data = torch.zeros(2 * NL, 1, c.num_key_value_heads, c.max_position_embeddings, c.hidden_size // c.num_attention_heads,
                       dtype=base.dtype)
    cur = torch.zeros(2 * NL, dtype=torch.long)
    pkv = [[KVCache(data[2 * i + j], cur[2 * i + j]) for j in (0, 1)] for i in range(NL)]
    return pkv, [data], cur


def build_spec(base, draft):
    sm = SpecModel.__new__(SpecModel)  # shim 2
    nn.Module.__init__(sm)
    sm.base_model, sm.config, sm.spec_layer = base, base.config, draft
    sm.tokenizer = SimpleNamespace(eos_token_id=2)
    sm.past_key_values, sm.past_key_values_data, sm.current_length_data = make_kv(base)  # shim 3
    return sm


def f32(x):
    return x.detach().to(torch.float32).cpu().numpy()


# ---------------------------------------------------------------------------------------------------
def g1_imgadaptor():
    out = {}
    for q in (2, 5):
        m, _ = build_draft(num_q=q, seed=11)
        rng = np.random.default_rng(100 + q)
        x = synth.bf16_grid(rng.standard_normal((1, 37, T["D"]), dtype=np.float32) * 0.05)
        out[f"x_q{q}"] = x
        out[f"y_q{q}"] = f32(m.imadpt(t(x)))[0]
    np.savez_compressed(os.path.join(OUT, "g1_imgadaptor.npz"), **out)


def g2_prefill():
    """Model.forward prefill: single image run (q in {2,5}) and text-only; full out, compressed KV, real_len."""
    out = {}
    L = 48
    for tag, q, n_pre, n_img in (("img_q2", 2, 6, 21), ("img_q5", 5, 9, 17), ("txt", 2, 48, 0)):
        m, _ = build_draft(num_q=q, seed=12)
        rng = np.random.default_rng(200 + q + n_img)
        hidden = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32))
        embeds = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32) * 0.05)
        mask = np.zeros((1, L), bool)
        mask[0, n_pre : n_pre + n_img] = True
        o, kv = m(t(hidden), inputs_embeds=t(embeds), use_cache=True, image_mask=(torch.from_numpy(mask) if n_img else None))
        out[f"{tag}_hidden"], out[f"{tag}_embeds"], out[f"{tag}_mask"] = hidden[0], embeds[0], mask[0]
        out[f"{tag}_out"] = f32(o)[0]
        out[f"{tag}_k"] = f32(kv[0][0])[0]
        out[f"{tag}_v"] = f32(kv[0][1])[0]
        out[f"{tag}_real_len"] = np.int64(int(kv[0][2]))
        out[f"{tag}_g"] = f32(m.last_img_hidden)
    np.savez_compressed(os.path.join(OUT, "g2_prefill.npz"), **out)


def g3_decode():
    """decode-branch forward with explicit positions + tree mask on top of a prefill KV."""
    m, _ = build_draft(num_q=2, seed=13)
    rng = np.random.default_rng(300)
    L = 40
    hidden = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32))
    embeds = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32) * 0.05)
    mask = np.zeros((1, L), bool)
    mask[0, 5:25] = True
    _, kv = m(t(hidden), inputs_embeds=t(embeds), use_cache=True, image_mask=torch.from_numpy(mask))
    # catch-up of 3 tokens (positions follow real_len)
    h2 = synth.bf16_grid(rng.standard_normal((1, 3, T["D"]), dtype=np.float32))
    ids2 = rng.integers(3, T["V"], size=(1, 3))
    o2, kv2 = m(t(h2), input_ids=torch.from_numpy(ids2), past_key_values=kv, use_cache=True)
    # one tree level: 8 tokens, same position, eye mask ; then a second level with parent selection
    h3 = synth.bf16_grid(rng.standard_normal((1, 8, T["D"]), dtype=np.float32))
    ids3 = rng.integers(3, T["V"], size=(1, 8))
    m.tree_mask = m.tree_mask_init
    pos3 = torch.full((8,), L + 3, dtype=torch.long)
    o3, kv3 = m(t(h3), input_ids=torch.from_numpy(ids3), past_key_values=kv2, position_ids=pos3, use_cache=True)
    out_ids = np.array([0, 0, 3, 5, 5, 5, 1, 7])
    tm = torch.cat((m.tree_mask_init[:, :, out_ids], m.tree_mask_init), dim=3)
    m.tree_mask = tm
    h4 = synth.bf16_grid(rng.standard_normal((1, 8, T["D"]), dtype=np.float32))
    ids4 = rng.integers(3, T["V"], size=(1, 8))
    pos4 = torch.full((8,), L + 4, dtype=torch.long)
    o4, kv4 = m(t(h4), input_ids=torch.from_numpy(ids4), past_key_values=kv3, position_ids=pos4, use_cache=True)
    np.savez_compressed(
        os.path.join(OUT, "g3_decode.npz"), hidden=hidden[0], embeds=embeds[0], mask=mask[0],
        h2=h2[0], ids2=ids2[0], o2=f32(o2)[0], h3=h3[0], ids3=ids3[0], o3=f32(o3)[0],
        h4=h4[0], ids4=ids4[0], o4=f32(o4)[0], out_ids=out_ids, tm4=f32(tm)[0, 0],
        k4=f32(kv4[0][0])[0], v4=f32(kv4[0][1])[0], real_len4=np.int64(int(kv4[0][2])),
    )


def g4_topk():
    """topK_genrate full outputs: prefill call (image) then a decode call; greedy and sampling row order."""
    out = {}
    base, _ = build_target(seed=20)
    head = base.lm_head
    for tag, lp in (("greedy", None), ("sampling", object())):
        m, _ = build_draft(num_q=2, seed=14)
        m.reset_kv()
        rng = np.random.default_rng(400)
        L = 44
        hidden = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32))
        embeds = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32) * 0.05)
        mask = np.zeros((1, L), bool)
        mask[0, 4:30] = True
        ids = rng.integers(3, T["V"], size=(1, L + 1))
        r = m.topK_genrate(t(hidden), torch.from_numpy(ids), head, lp, inputs_embeds=t(embeds),
                           image_mask=torch.from_numpy(mask))
        h2 = synth.bf16_grid(rng.standard_normal((1, 3, T["D"]), dtype=np.float32))
        ids2 = np.concatenate([ids, rng.integers(3, T["V"], size=(1, 3))], axis=1)
        r2 = m.topK_genrate(t(h2), torch.from_numpy(ids2), head, lp)
        if tag == "greedy":
            out.update(hidden=hidden[0], embeds=embeds[0], mask=mask[0], ids=ids[0], h2=h2[0], ids2=ids2[0])
        for nm, rr in (("a", r), ("b", r2)):
            out[f"{tag}_{nm}_tokens"] = rr[0][0].numpy()
            out[f"{tag}_{nm}_retrieve"] = rr[1].numpy()
            out[f"{tag}_{nm}_mask"] = rr[2][0, 0].numpy()
            out[f"{tag}_{nm}_pos"] = rr[3].numpy()
    np.savez_compressed(os.path.join(OUT, "g4_topk.npz"), **out)


def g14_tree_levels():
    """Float outputs of every draft forward inside topK_genrate (prefill call with an image run, then a decode call): the hidden
    rows of each tree level depend on how the level mask EVOLVES (cnets_ours.py:1163-1165, `tree_mask[:, :, out_ids]` = the parent's
    ROW), which the integer outputs of g4 are not sensitive to.  Captured with a forward hook on the reference's draft."""
    out = {}
    base, _ = build_target(seed=20)
    head = base.lm_head
    m, _ = build_draft(num_q=2, seed=14)
    m.reset_kv()
    rec = []
    hook = m.register_forward_hook(lambda mod, inp, res: rec.append(f32(res[0])[0]))
    rng = np.random.default_rng(401)
    L = 37
    hidden = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32))
    embeds = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32) * 0.05)
    mask = np.zeros((1, L), bool)
    mask[0, 6:27] = True
    ids = rng.integers(3, T["V"], size=(1, L + 1))
    r = m.topK_genrate(t(hidden), torch.from_numpy(ids), head, None, inputs_embeds=t(embeds), image_mask=torch.from_numpy(mask))
    n_a = len(rec)
    h2 = synth.bf16_grid(rng.standard_normal((1, 4, T["D"]), dtype=np.float32))
    ids2 = np.concatenate([ids, rng.integers(3, T["V"], size=(1, 4))], axis=1)
    r2 = m.topK_genrate(t(h2), torch.from_numpy(ids2), head, None)
    hook.remove()
    assert n_a == 4 and len(rec) == 8  # 1 prefill/catch-up forward + depth level forwards per call
    out.update(hidden=hidden[0], embeds=embeds[0], mask=mask[0], ids=ids[0], h2=h2[0], ids2=ids2[0])
    for c, (nm, rr) in enumerate((("a", r), ("b", r2))):
        out[f"{nm}_first_last_row"] = rec[4 * c][-1]
        for lvl in range(3):
            out[f"{nm}_level{lvl}_out"] = rec[4 * c + 1 + lvl]
        out[f"{nm}_tokens"] = rr[0][0].numpy()
        out[f"{nm}_mask"] = rr[2][0, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "g14_tree_levels.npz"), **out)


def g5_verify():
    """target prefill then tree verify with a tree mask; logits, hidden, KV lengths."""
    base, _ = build_target(seed=21)
    pkv, data, cur = make_kv(base)
    rng = np.random.default_rng(500)
    L = 23
    ids = rng.integers(3, T["V"], size=(1, L))
    o = base(input_ids=torch.from_numpy(ids), past_key_values=pkv, output_hidden_states=True, return_dict=True)
    # a hand-made 9-node tree
    parents = [-1, 0, 0, 1, 1, 2, 3, 3, 6]
    Tn = len(parents)
    tm = np.zeros((Tn, Tn), np.float32)
    for i, p in enumerate(parents):
        tm[i, i] = 1
        while p >= 0:
            tm[i, p] = 1
            p = parents[p]
    pos = tm.sum(1).astype(np.int64) - 1
    cand = rng.integers(3, T["V"], size=(1, Tn))
    base.model.tree_mask = t(tm)[None, None]
    o2 = base(input_ids=torch.from_numpy(cand), past_key_values=pkv, position_ids=torch.from_numpy(pos + L),
              output_hidden_states=True, return_dict=True)
    base.model.tree_mask = None
    np.savez_compressed(
        os.path.join(OUT, "g5_verify.npz"), ids=ids[0], prefill_logits=f32(o.logits)[0], prefill_hidden=f32(o.hidden_states[-1])[0],
        tree_mask=tm, tree_pos=pos, cand=cand[0], logits=f32(o2.logits)[0], hidden=f32(o2.hidden_states[-1])[0],
        cur=cur.numpy().copy(), k0=f32(data[0][0, 0, :, : L + Tn]), v1=f32(data[0][3, 0, :, : L + Tn]),
    )


def g6_posterior():
    """evaluate_posterior greedy: random, zero-accept, multi-max ties (first max wins), full accept."""
    rng = np.random.default_rng(600)
    out = {}
    V = 50
    cases = []
    for ci in range(6):
        n_leaf, m = int(rng.integers(2, 9)), int(rng.integers(2, 6))
        logits = rng.standard_normal((n_leaf, m, V)).astype(np.float32)
        am = logits.argmax(-1)
        cand = rng.integers(0, V, size=(n_leaf, m))
        if ci >= 1:  # plant accepted prefixes
            for r in range(n_leaf):
                a = int(rng.integers(0, m))
                cand[r, 1 : 1 + a] = am[r, :a]
        if ci == 2:  # tie: two rows share the max accept length
            cand[1, 1:] = am[1, : m - 1]
            cand[0, 1:] = am[0, : m - 1]
        if ci == 3:  # zero accept everywhere
            cand[:, 1] = (am[:, 0] + 1) % V
        if ci == 4:  # padding -1 inside candidates
            cand[:, -1] = -1
        cases.append((logits, cand))
    for i, (logits, cand) in enumerate(cases):
        b, a, p = utils.evaluate_posterior(t(logits), torch.from_numpy(cand), None)
        out[f"logits{i}"], out[f"cand{i}"] = logits, cand
        out[f"best{i}"], out[f"acc{i}"], out[f"p{i}"] = np.int64(int(b)), np.int64(int(a)), f32(p)
    out["n"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "g6_posterior.npz"), **out)


def g7_posterior_sampling():
    """evaluate_posterior, sampling branch (utils.py:453-493), with torch.rand_like replaced by a recorded uniform tensor and the
    processor list the reference builds for --temperature T (prepare_logits_processor, utils.py:39-55)."""
    rng = np.random.default_rng(700)
    out = {}
    V = 40
    n = 0
    real_rand_like = torch.rand_like
    for T_, K_ in ((1.0, 0), (0.7, 0), (1.5, 0), (1.0, 5), (0.8, 3), (1.3, 12)):  # K_ > 0: + TopKLogitsWarper (utils.py:52-53)
        lp = utils.prepare_logits_processor(temperature=T_, top_k=K_)
        for rep in range(6):
            n_leaf, m = int(rng.integers(2, 9)), int(rng.integers(2, 6))
            # leaves of a small prefix tree: rows share prefixes so that is_eq / the candidate set logic is exercised
            cand = np.zeros((n_leaf, m), np.int64)
            cand[:, 0] = 7
            for c in range(1, m):
                for j in range(n_leaf):
                    cand[j, c] = cand[j - 1, c] if (j > 0 and rng.random() < 0.5 and (cand[j, :c] == cand[j - 1, :c]).all()) else rng.integers(0, V)
            if rep % 3 == 2:
                cand[rng.integers(0, n_leaf), -1] = -1
            logits = (rng.standard_normal((n_leaf, m, V)) * 2).astype(np.float32)
            # rows with equal prefixes must carry the same distribution (they are gathers of the same tree node)
            for c in range(m):
                for j in range(1, n_leaf):
                    for j2 in range(j):
                        if (cand[j, : c + 1] == cand[j2, : c + 1]).all():
                            logits[j, c] = logits[j2, c]
                            break
            if rep % 2 == 0:  # make acceptance likely along row 0
                for c in range(m - 1):
                    if cand[0, c + 1] >= 0:
                        logits[:, c, cand[0, c + 1]] += 6
                for c in range(m):
                    for j in range(1, n_leaf):
                        for j2 in range(j):
                            if (cand[j, : c + 1] == cand[j2, : c + 1]).all():
                                logits[j, c] = logits[j2, c]
                                break
            u = rng.random((n_leaf, m)).astype(np.float32)
            if rep % 2 == 0:
                u[0, : max(2, m - rep // 2)] *= 0.05  # accepted for sure on the first levels of row 0
            torch.rand_like = lambda t_, dtype=None, _u=u: torch.from_numpy(_u).to(dtype or t_.dtype)
            try:
                b, a, p_ = utils.evaluate_posterior(t(logits), torch.from_numpy(cand), lp)
            finally:
                torch.rand_like = real_rand_like
            out[f"logits{n}"], out[f"cand{n}"], out[f"u{n}"], out[f"T{n}"], out[f"K{n}"] = logits, cand, u, np.float32(T_), np.int64(K_)
            out[f"best{n}"], out[f"acc{n}"], out[f"p{n}"] = np.int64(int(b)), np.int64(int(a)), f32(p_)
            n += 1
    out["n"] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, "g7_posterior_sampling.npz"), **out)


def g8_loop():
    """Whole-loop token streams (+ per-round accept lengths, greedy-AR equality):
    text-only random pair, structured (successor) pair, and image-path structured pair."""
    out = {}
    # (a) fully random pair, text-only — SpecModel.specgenerate end to end
    for si, seed in enumerate((0, 1, 2)):
        base, _ = build_target(seed=30 + seed)
        draft, _ = build_draft(seed=40 + seed)
        sm = build_spec(base, draft)
        rng = np.random.default_rng(800 + seed)
        ids = rng.integers(3, T["V"], size=(1, 20))
        o, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids), temperature=0.0, max_new_tokens=24, log=True,
                                                 return_acceptance_len=True)
        base.model.tree_mask = None
        out[f"rand{si}_ids"], out[f"rand{si}_out"] = ids[0], o[0].numpy()
        out[f"rand{si}_new_token"], out[f"rand{si}_idx"], out[f"rand{si}_acc"] = np.int64(int(new_token)), np.int64(idx), np.array(acc)
        out[f"rand{si}_ar"] = greedy_ar(base, ids, o.shape[1] - ids.shape[1])
    # (b) structured pair, text-only
    for si, seed in enumerate((0, 1)):
        base, tw = build_target(seed=50 + seed, structured=True)
        draft, _ = build_draft(seed=60 + seed, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
        sm = build_spec(base, draft)
        rng = np.random.default_rng(900 + seed)
        ids = rng.integers(3, T["V"], size=(1, 16))
        o, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids), temperature=0.0, max_new_tokens=40, log=True,
                                                 return_acceptance_len=True)
        base.model.tree_mask = None
        out[f"succ{si}_ids"], out[f"succ{si}_out"] = ids[0], o[0].numpy()
        out[f"succ{si}_new_token"], out[f"succ{si}_idx"], out[f"succ{si}_acc"] = np.int64(int(new_token)), np.int64(idx), np.array(acc)
        out[f"succ{si}_ar"] = greedy_ar(base, ids, o.shape[1] - ids.shape[1])
    # (c) image path, structured pair: drive the loop body by hand, restating spec_model_ours.py:455-547
    base, tw = build_target(seed=70, structured=True)
    draft, _ = build_draft(seed=71, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    sm = build_spec(base, draft)
    ids, emb, mask = synth.make_request(T["V"], T["D"], 5, 30, 9, seed=7, embed=tw["model.embed_tokens.weight"])
    o, acc = image_loop(sm, ids, emb, mask, max_new_tokens=30)
    out["img_ids"], out["img_emb"], out["img_mask"], out["img_out"], out["img_acc"] = ids, emb, mask, o, np.array(acc)
    np.savez_compressed(os.path.join(OUT, "g8_loop.npz"), **out)


def greedy_ar(base, ids, n_new):
    pkv, _, _ = make_kv(base)
    base.model.tree_mask = None
    cur = torch.from_numpy(ids)
    o = base(input_ids=cur, past_key_values=pkv, return_dict=True)
    toks = []
    for _ in range(n_new):
        tok = int(torch.argmax(o.logits[0, -1]))
        toks.append(tok)
        o = base(input_ids=torch.tensor([[tok]]), past_key_values=pkv, return_dict=True)
    return np.array(toks, dtype=np.int64)


def image_loop(sm, ids, emb, mask, max_new_tokens):
    base, draft = sm.base_model, sm.spec_layer
    pkv, pkv_data, cur = sm.past_key_values, sm.past_key_values_data, sm.current_length_data
    cur.zero_()
    draft.reset_kv()
    utils.reset_tree_mode(sm)
    input_ids = torch.from_numpy(ids)[None]
    inputs_embeds = t(emb)[None]
    _, orig, hidden = sm(None, past_key_values=pkv, output_orig=True, inputs_embeds=inputs_embeds)
    token = torch.argmax(orig[:, -1])[None, None]
    input_ids = torch.cat((input_ids, token), dim=1)
    dt, ri, tm, tp = draft.topK_genrate(hidden, input_ids, base.lm_head, None, inputs_embeds=inputs_embeds,
                                        image_mask=torch.from_numpy(mask)[None])
    input_ids = input_ids[:, :-1]
    padding = torch.zeros(1, 1, dtype=torch.long) - 1
    new_token, acc = 0, []
    for _ in range(200):
        base.model.tree_mask = tm
        logits, hidden_new, _ = utils.tree_decoding(sm, dt, pkv, tp, input_ids, ri)
        dt2 = torch.cat((dt, padding), dim=1)
        cand = dt2[0, ri]
        best, al, sample_p = utils.evaluate_posterior(logits, cand, None)
        acc.append(int(al))
        input_ids, dt, ri, tm, tp, new_token, _, _ = utils.update_inference_inputs(
            input_ids, cand, best, al, ri, None, new_token, pkv_data, cur, sm, hidden_new, sample_p)
        if new_token > max_new_tokens:
            break
    base.model.tree_mask = None
    return input_ids[0].numpy(), acc


def g9_bf16():
    """Reference run in torch-bf16 on CPU: pins the oracle's bf16 rounding-point emulation (tolerance-level)."""
    out = {}
    m, _ = build_draft(num_q=2, seed=15, dtype=torch.bfloat16)
    rng = np.random.default_rng(950)
    L = 36
    hidden = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32))
    embeds = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32) * 0.05)
    mask = np.zeros((1, L), bool)
    mask[0, 3:27] = True
    o, kv = m(t(hidden, torch.bfloat16), inputs_embeds=t(embeds, torch.bfloat16), use_cache=True, image_mask=torch.from_numpy(mask))
    h2 = synth.bf16_grid(rng.standard_normal((1, 4, T["D"]), dtype=np.float32))
    ids2 = rng.integers(3, T["V"], size=(1, 4))
    o2, _ = m(t(h2, torch.bfloat16), input_ids=torch.from_numpy(ids2), past_key_values=kv, use_cache=True)
    out.update(d_hidden=hidden[0], d_embeds=embeds[0], d_mask=mask[0], d_out_last=f32(o)[0, -1], d_k=f32(kv[0][0])[0],
               d_h2=h2[0], d_ids2=ids2[0], d_o2=f32(o2)[0])
    base, _ = build_target(seed=22, dtype=torch.bfloat16)
    pkv, data, cur = make_kv(base)
    ids = rng.integers(3, T["V"], size=(1, 19))
    o = base(input_ids=torch.from_numpy(ids), past_key_values=pkv, output_hidden_states=True, return_dict=True)
    out.update(t_ids=ids[0], t_logits=f32(o.logits)[0], t_hidden=f32(o.hidden_states[-1])[0])
    np.savez_compressed(os.path.join(OUT, "g9_bf16.npz"), **out)


def g10_qwen():
    """Qwen2.5-VL text model of the reference (modeling_qwen2_5_vl_kv.Qwen2_5_VLModel: KVCache.cat, tree mask, multimodal rotary,
    GQA, q/k/v bias, SDPA) on CPU: prefill with an image block in the 3-component positions, then a tree verify at delta-shifted
    positions.  Extra shims for transformers 5.x: text_config carries the flat attributes the reference reads, rope_scaling in
    the 4.49 form, and the 'default' rope-init function the reference looks up."""
    from vispec.model import modeling_qwen2_5_vl_kv as q
    Q = synth.QWEN_TINY

    def default_rope(config, device=None, seq_len=None, **kw):
        dim = config.hidden_size // config.num_attention_heads
        return 1.0 / (config.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim)), 1.0

    q.ROPE_INIT_FUNCTIONS = dict(q.ROPE_INIT_FUNCTIONS)
    q.ROPE_INIT_FUNCTIONS["default"] = default_rope
    cfg = q.Qwen2_5_VLConfig(text_config=dict(vocab_size=Q["V"], hidden_size=Q["D"], intermediate_size=Q["I"], num_hidden_layers=Q["NL"],
                                              num_attention_heads=Q["H"], num_key_value_heads=Q["Hkv"], max_position_embeddings=Q["max_pos"],
                                              rms_norm_eps=Q["eps"], rope_theta=Q["theta"], bos_token_id=1, eos_token_id=2))
    tc = cfg.text_config
    tc.pad_token_id = 0
    tc.rope_scaling = {"type": "mrope", "rope_type": "default", "mrope_section": list(Q["mrope_section"])}
    tc.rope_theta = Q["theta"]
    tc._attn_implementation = "sdpa"
    tc.sliding_window, tc.use_sliding_window, tc.max_window_layers = 32768, False, Q["NL"]
    m = q.Qwen2_5_VLModel(tc).eval()
    w = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=90, qkv_bias=True, H_kv=Q["Hkv"])
    sd = {k[len("model."):]: t(v) for k, v in w.items() if k.startswith("model.")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in k for k in missing), (missing, unexpected)
    head = t(w["lm_head.weight"])
    NL, hd = Q["NL"], Q["D"] // Q["H"]
    data = torch.zeros(2 * NL, 1, Q["Hkv"], Q["max_pos"], hd)
    cur = torch.zeros(2 * NL, dtype=torch.long)
    pkv = [[KVCache(data[2 * i + j], cur[2 * i + j]) for j in (0, 1)] for i in range(NL)]
    rng = np.random.default_rng(1000)
    IMG = Q["V"] - 1
    ids = np.concatenate([rng.integers(3, IMG, 5), np.full(12, IMG), rng.integers(3, IMG, 7)])  # one 1x6x8-patch image -> 3x4 tokens
    pos3, delta = synth.qwen_rope_index(ids, IMG, [(1, 6, 8)])
    L = len(ids)
    emb = synth.bf16_grid(rng.standard_normal((L, Q["D"]), dtype=np.float32) * 0.05)
    o = m(inputs_embeds=t(emb)[None], position_ids=torch.from_numpy(pos3)[:, None], past_key_values=pkv, use_cache=True, return_dict=True)
    hid = o.last_hidden_state[0]
    parents = [-1, 0, 0, 1, 1, 2, 3, 3, 6]
    Tn = len(parents)
    tm = np.zeros((Tn, Tn), np.float32)
    for i, p_ in enumerate(parents):
        tm[i, i] = 1
        while p_ >= 0:
            tm[i, p_] = 1
            p_ = parents[p_]
    tpos = tm.sum(1).astype(np.int64) - 1
    cand = rng.integers(3, IMG, Tn)
    m.tree_mask = t(tm)[None, None]
    p1 = torch.from_numpy(tpos + L + delta)[None, None].expand(3, 1, Tn)  # utils.py:397-402
    o2 = m(input_ids=torch.from_numpy(cand)[None], position_ids=p1, past_key_values=pkv, use_cache=True, return_dict=True)
    m.tree_mask = None
    hid2 = o2.last_hidden_state[0]
    np.savez_compressed(os.path.join(OUT, "g10_qwen.npz"), ids=ids, emb=emb, pos3=pos3, delta=np.int64(delta), prefill_hidden=f32(hid),
                        prefill_logits=f32(torch.nn.functional.linear(hid, head)), tree_mask=tm, tree_pos=tpos, cand=cand,
                        hidden=f32(hid2), logits=f32(torch.nn.functional.linear(hid2, head)), cur=cur.numpy().copy(),
                        k0=f32(data[0, 0, :, : L + Tn]), v1=f32(data[3, 0, :, : L + Tn]))


def g15_qwen_rope_index():
    """The reference's own Qwen2_5_VLForConditionalGeneration.get_rope_index (modeling_qwen2_5_vl_kv.py:1789-1975) on prompts with image
    runs, video runs (temporal index scaled by second_per_grid_ts * tokens_per_second) and both: pins vispec_amd.synth.qwen_rope_index,
    the function the product's Qwen prefill positions and rope_delta come from.  Stored: prompts, grids, the reference's outputs."""
    from types import SimpleNamespace as NS
    from vispec.model import modeling_qwen2_5_vl_kv as q
    IMG, VID, VS = 900, 901, 902
    rng = np.random.default_rng(1500)
    txt = lambda n: rng.integers(3, 800, n)
    run = lambda tok, g: np.concatenate([[VS], np.full(g[0] * (g[1] // 2) * (g[2] // 2), tok)])
    cases = [
        dict(parts=[("t", 5), ("i", (1, 6, 8)), ("t", 7)], sec=None, tps=2),
        dict(parts=[("t", 3), ("i", (1, 4, 4)), ("t", 2), ("i", (1, 8, 6)), ("t", 4)], sec=None, tps=2),
        dict(parts=[("t", 4), ("v", (3, 4, 6)), ("t", 5)], sec=[1.0], tps=2),
        dict(parts=[("t", 2), ("v", (4, 4, 4)), ("t", 3)], sec=None, tps=2),          # second_per_grid_ts absent -> 1.0
        dict(parts=[("t", 2), ("i", (1, 4, 6)), ("t", 1), ("v", (5, 6, 4)), ("t", 6)], sec=[0.5], tps=25),
        dict(parts=[("v", (2, 4, 4)), ("t", 3), ("v", (3, 4, 4)), ("i", (1, 4, 4)), ("t", 2)], sec=[2.0, 0.3], tps=2),
        dict(parts=[("t", 9)], sec=None, tps=2),                                        # text only
    ]
    out = {"n_cases": np.int64(len(cases)), "ids_image": np.int64(IMG), "ids_video": np.int64(VID)}
    for ci, c in enumerate(cases):
        ids, ig, vg = [], [], []
        for kind, v in c["parts"]:
            if kind == "t":
                ids.append(txt(v))
            elif kind == "i":
                ids.append(run(IMG, v)); ig.append(v)
            else:
                ids.append(run(VID, v)); vg.append(v)
        ids = np.concatenate(ids).astype(np.int64)
        fake = NS(config=NS(vision_config=NS(spatial_merge_size=2, tokens_per_second=c["tps"]), image_token_id=IMG, video_token_id=VID,
                            vision_start_token_id=VS))
        pos, delta = q.Qwen2_5_VLForConditionalGeneration.get_rope_index(
            fake, torch.from_numpy(ids)[None], torch.tensor(ig) if ig else None, torch.tensor(vg) if vg else None,
            torch.tensor(c["sec"], dtype=torch.float32) if c["sec"] is not None else None)
        out[f"c{ci}_ids"] = ids
        out[f"c{ci}_image_grids"] = np.asarray(ig, np.int64).reshape(-1, 3)
        out[f"c{ci}_video_grids"] = np.asarray(vg, np.int64).reshape(-1, 3)
        out[f"c{ci}_sec"] = np.asarray(c["sec"] if c["sec"] is not None else [], np.float32)
        out[f"c{ci}_has_sec"] = np.int64(c["sec"] is not None)
        out[f"c{ci}_tps"] = np.float32(c["tps"])
        out[f"c{ci}_pos"] = pos[:, 0].numpy().astype(np.int64)
        out[f"c{ci}_delta"] = np.int64(int(delta.reshape(-1)[0]))
    np.savez_compressed(os.path.join(OUT, "g15_qwen_rope_index.npz"), **out)


def g16_multi_image_repaired():
    """MULTI-IMAGE draft prefill.  The reference as published cannot run it: the scatter matrix of Model.forward takes its row counts from
    the FIRST run's blocks (`h_s[0]`, `h_s[1]`, cnets_ours.py:939-940) and crashes on the second image.  The rows it
    meant are the current run's (`h_s[-2]`, `h_s[-1]`): the reference's own forward is re-compiled here with exactly those two index
    expressions replaced — nothing else — and run on prompts with two and three image runs (one of them touching the end of the prompt).
    Fixtures from this function are REFERENCE-INTENT (repaired), parity unpinned upstream.  Stored: inputs, the compressed K/V, real_len,
    the last global feature g and the output row topK_genrate consumes (out[:, -1])."""
    import inspect
    import textwrap
    src = textwrap.dedent(inspect.getsource(cnets_ours.Model.forward))
    a, b = "eye_m[img_id_start : img_id_start + h_s[0].shape[0], :]", "eye_m[img_id_end - h_s[1].shape[0] : img_id_end, :]"
    assert src.count(a) == 1 and src.count(b) == 1, "the reference's scatter-matrix lines moved: re-derive the repair"
    src = src.replace(a, a.replace("h_s[0]", "h_s[-2]")).replace(b, b.replace("h_s[1]", "h_s[-1]"))
    ns = {}
    exec(compile(src, "<repaired Model.forward>", "exec"), vars(cnets_ours), ns)
    repaired = ns["forward"]
    out = {}
    cases = {"two": (2, [(4, 13), (6, 9)], 7), "three_q3": (3, [(3, 8), (2, 11), (5, 6)], 4), "image_last": (2, [(5, 10), (4, 12)], 0),
             "one_q5": (5, [(9, 17)], 6)}  # a single run through the same (repaired) code path
    for tag, (q, runs, n_tail) in cases.items():
        m, w16 = build_draft(num_q=q, seed=16)
        rng = np.random.default_rng(1600 + q + len(runs) + n_tail)
        mask = np.concatenate([np.concatenate([np.zeros(nt, bool), np.ones(ni, bool)]) for nt, ni in runs] + [np.zeros(n_tail, bool)])[None]
        L = mask.shape[1]
        hidden = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32))
        embeds = synth.bf16_grid(rng.standard_normal((1, L, T["D"]), dtype=np.float32) * 0.05)
        first_tok = int(rng.integers(3, T["V"] - 1))
        embeds[0, -1] = w16["embed_tokens.weight"][first_tok]  # topK_genrate's last shifted row is the draft's embedding of the sampled token (:1081-1082)
        out[f"{tag}_first_tok"] = np.int64(first_tok)
        o, kv = repaired(m, t(hidden), inputs_embeds=t(embeds), use_cache=True, image_mask=torch.from_numpy(mask))
        out[f"{tag}_hidden"], out[f"{tag}_embeds"], out[f"{tag}_mask"], out[f"{tag}_q"] = hidden[0], embeds[0], mask[0], np.int64(q)
        out[f"{tag}_out_last"] = f32(o)[0, -1]
        out[f"{tag}_k"], out[f"{tag}_v"] = f32(kv[0][0])[0], f32(kv[0][1])[0]
        out[f"{tag}_real_len"] = np.int64(int(kv[0][2]))
        out[f"{tag}_g"] = f32(m.last_img_hidden)
    np.savez_compressed(os.path.join(OUT, "g16_multi_image_repaired.npz"), **out)


def repaired_forward():
    """The reference's Model.forward with the two scatter-matrix index expressions of cnets_ours.py:939-940 taken from the CURRENT run
    (see g16_multi_image_repaired); compiled from the reference's own source at generation time."""
    import inspect
    import textwrap
    src = textwrap.dedent(inspect.getsource(cnets_ours.Model.forward))
    a, b = "eye_m[img_id_start : img_id_start + h_s[0].shape[0], :]", "eye_m[img_id_end - h_s[1].shape[0] : img_id_end, :]"
    assert src.count(a) == 1 and src.count(b) == 1, "the reference's scatter-matrix lines moved: re-derive the repair"
    src = src.replace(a, a.replace("h_s[0]", "h_s[-2]")).replace(b, b.replace("h_s[1]", "h_s[-1]"))
    ns = {}
    exec(compile(src, "<repaired Model.forward>", "exec"), vars(cnets_ours), ns)
    return ns["forward"]


def g17_multi_image_loop():
    """Whole draft-and-verify loop on a MULTI-IMAGE prompt (three image runs, structured pair, the loop body of g8's image case) with the
    repaired draft forward installed for the duration: token stream + accept lengths.  REFERENCE-INTENT (the published code crashes on the
    second image run), parity unpinned upstream."""
    base, tw = build_target(seed=70, structured=True)
    draft, _ = build_draft(seed=71, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    sm = build_spec(base, draft)
    rng = np.random.default_rng(1700)
    E = tw["model.embed_tokens.weight"]
    ids, mask = [], []
    for n_txt, n_img in ((4, 14), (3, 22), (6, 9)):
        ids += rng.integers(3, T["V"] - 1, size=n_txt).tolist() + [T["V"] - 1] * n_img
        mask += [False] * n_txt + [True] * n_img
    ids += rng.integers(3, T["V"] - 1, size=7).tolist()
    mask += [False] * 7
    ids, mask = np.array(ids, np.int64), np.array(mask)
    emb = E[ids].copy()
    emb[mask] = synth.bf16_grid(rng.standard_normal((int(mask.sum()), T["D"]), dtype=np.float32) * 0.05)
    orig = cnets_ours.Model.forward
    cnets_ours.Model.forward = repaired_forward()
    try:
        o, acc = image_loop(sm, ids, emb, mask, max_new_tokens=30)
    finally:
        cnets_ours.Model.forward = orig
    np.savez_compressed(os.path.join(OUT, "g17_multi_image_loop.npz"), ids=ids, emb=emb, mask=mask, out=o, acc=np.array(acc))


def g11_update_inference_inputs():
    """utils.update_inference_inputs in isolation (SURVEY §8c "G7"): accepted ids appended, KV rows gathered from the tree
    slots into [n, n+a+1) of EVERY cache tensor, lengths, the hidden rows handed to the draft, next token — greedy and
    multinomial.  The draft call is intercepted (its own behaviour is pinned by g4)."""
    rng = np.random.default_rng(1100)
    out = {}
    cases = [(0, 0, False), (2, 3, False), (1, 1, False), (3, 2, True)]  # (best, accept_length, sampling)
    for ci, (best, acc, sampling) in enumerate(cases):
        n, Tn, V, D = 11, 9, 40, 8
        ids = rng.integers(3, V, n)
        ri = np.array([[0, 1, 3, 7], [0, 1, 4, -1], [0, 2, 5, 8], [0, 2, 6, -1]], np.int64)
        draft_tokens = rng.integers(3, V, Tn)
        ext = np.concatenate([draft_tokens, [-1]])
        cand = ext[ri]
        data = rng.standard_normal((4, 1, 2, 32, 4)).astype(np.float32)
        data2 = rng.standard_normal((2, 1, 2, 32, 4)).astype(np.float32)  # a second device's tensor (kv_cache.py:121-141)
        cur = np.full(6, n + Tn, np.int64)
        hid = rng.standard_normal((1, Tn, D)).astype(np.float32)
        sp = np.abs(rng.standard_normal(V)).astype(np.float32)
        sp /= sp.sum()
        seen = {}

        class Draft:
            def topK_genrate(self, hidden, input_ids=None, head=None, logits_processor=None):
                seen["hidden"], seen["ids"] = f32(hidden), input_ids.numpy().copy()
                return "dt", "ri", "tm", "tp"

        model = SimpleNamespace(spec_layer=Draft(), base_model=SimpleNamespace(lm_head="head"))
        td, td2, tc = t(data.copy()), t(data2.copy()), torch.from_numpy(cur.copy())  # t() aliases its argument
        u = 0.37
        orig = torch.multinomial
        torch.multinomial = lambda prob, k: torch.searchsorted(torch.cumsum(prob.double(), 0), torch.tensor([u], dtype=torch.float64) * prob.double().sum()).clamp(max=prob.numel() - 1)
        try:
            r = utils.update_inference_inputs(torch.from_numpy(ids)[None], torch.from_numpy(cand), torch.tensor(best), acc, torch.from_numpy(ri),
                                              (object() if sampling else None), 5, [td, td2], tc, model, t(hid), t(sp))
        finally:
            torch.multinomial = orig
        assert r[1:5] == ("dt", "ri", "tm", "tp")
        out.update({f"ids{ci}": ids, f"ri{ci}": ri, f"cand{ci}": cand, f"best{ci}": np.int64(best), f"acc{ci}": np.int64(acc),
                    f"sampling{ci}": np.int64(sampling), f"u{ci}": np.float64(u), f"data{ci}": data, f"data2_{ci}": data2, f"hid{ci}": hid, f"sp{ci}": sp,
                    f"o_ids{ci}": r[0][0].numpy(), f"o_data{ci}": td.numpy(), f"o_data2_{ci}": td2.numpy(), f"o_cur{ci}": tc.numpy(),
                    f"o_new_token{ci}": np.int64(r[5]), f"o_token{ci}": np.int64(int(r[7])), f"o_hidden{ci}": seen["hidden"][0],
                    f"o_draft_ids{ci}": seen["ids"][0]})
    out["n"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "g11_update.npz"), **out)


def g13_real_dims():
    """Single ops at the REAL LLaVA-7B dims (D=4096, V=32064; SURVEY §8c "G9"): LM-head -> LogSoftmax -> top-k as topK_genrate does it
    (cnets_ours.py:1109-1123), and the draft's input fusion fc(cat(emb, img_fc(cat(h, g)))) (cnets_ours.py:982-988) — torch fp32 on CPU.
    Weights are seed-derived (numpy PCG64) and NOT stored; only inputs and results are."""
    D, V, k = 4096, 32064, 8
    rng = np.random.default_rng(1300)
    W = rng.standard_normal((V, D), dtype=np.float32) * np.float32(0.02)
    h = synth.bf16_grid(rng.standard_normal((8, D), dtype=np.float32))
    logits = torch.nn.functional.linear(t(h), t(W))
    logp = nn.LogSoftmax(dim=-1)(logits)
    top = torch.topk(logp, k, dim=-1)
    rng2 = np.random.default_rng(1301)
    fc_w, fc_b = rng2.standard_normal((D, 2 * D), dtype=np.float32) * np.float32(0.02), rng2.standard_normal(D, dtype=np.float32) * np.float32(0.02)
    ifc_w, ifc_b = rng2.standard_normal((D, 2 * D), dtype=np.float32) * np.float32(0.02), rng2.standard_normal(D, dtype=np.float32) * np.float32(0.02)
    emb = synth.bf16_grid(rng2.standard_normal((5, D), dtype=np.float32) * 0.05)
    hid = synth.bf16_grid(rng2.standard_normal((5, D), dtype=np.float32))
    g = synth.bf16_grid(rng2.standard_normal((1, D), dtype=np.float32))
    fc, ifc = nn.Linear(2 * D, D), nn.Linear(2 * D, D)
    fc.weight.data, fc.bias.data, ifc.weight.data, ifc.bias.data = t(fc_w), t(fc_b), t(ifc_w), t(ifc_b)
    fused = fc(torch.cat((t(emb), ifc(torch.cat((t(hid), t(g).expand(5, D)), dim=-1))), dim=-1))
    np.savez_compressed(os.path.join(OUT, "g13_real_dims.npz"), h=h, top_idx=top.indices.numpy(), top_logp=top.values.numpy(),
                        lse=torch.logsumexp(logits, -1).numpy(), emb=emb, hid=hid, g=g, fused=fused.detach().numpy())


def g12_kvcache():
    """KVCache.cat / .copy / .shape / current_length (kv_cache.py:4-66) on a CPU slab: a cat of the prompt rows, a cat of T tree
    rows, then the compaction copy of an accepted path."""
    rng = np.random.default_rng(1200)
    data = torch.zeros(1, 2, 16, 4)
    cur = torch.zeros((), dtype=torch.long)
    kv = KVCache(data, cur)
    a = t(rng.standard_normal((1, 2, 5, 4)).astype(np.float32))
    b = t(rng.standard_normal((1, 2, 6, 4)).astype(np.float32))
    v1 = kv.cat(a).numpy().copy()  # the returned views alias the slab: snapshot them before the next in-place op
    s1 = tuple(kv.shape)
    v2 = kv.cat(b).numpy().copy()
    s2 = tuple(kv.shape)
    d2 = data.clone()
    idx = torch.tensor([5, 7, 10])
    kv.copy(idx, 5)
    np.savez_compressed(os.path.join(OUT, "g12_kvcache.npz"), a=a.numpy(), b=b.numpy(), v1=v1, s1=np.array(s1), v2=v2,
                        s2=np.array(s2), d2=d2.numpy(), idx=idx.numpy(), d3=data.numpy().copy(), len3=np.int64(int(cur)), s3=np.array(tuple(kv.shape)))


def g18_prompts():
    """The reference's prompt front-ends (vispec/evaluation/*_prompt.py): every `build_prompt` is run with a RECORDING processor (a stand-in
    for transformers.AutoProcessor that stores the conversation handed to apply_chat_template and the keyword arguments of the processor call)
    on seeded sample data; the fixture holds what each benchmark's function built — conversation, processor-construction arguments, call
    arguments — as JSON.  Strings stand in for the images / videos (the functions only pass them through)."""
    import importlib
    import json
    import types
    from types import SimpleNamespace
    import transformers

    class Rec:
        log = []

        def __init__(self, ctor):
            self.ctor, self.calls = ctor, []

        @classmethod
        def from_pretrained(cls, *a, **kw):
            r = cls(dict(args=list(a), kwargs=kw))
            cls.log.append(r)
            return r

        def apply_chat_template(self, conv, **kw):
            self.conv, self.tmpl_kw = conv, kw
            return "<PROMPT>"

        def __call__(self, **kw):
            self.calls.append(kw)
            return SimpleNamespace(to=lambda dev: dict(device=dev))

    qv = types.ModuleType("qwen_vl_utils")
    qv.process_vision_info = lambda conv, return_video_kwargs=False: (None, ["<FRAMES>"], {"fps": [2.0]})
    sys.modules["qwen_vl_utils"] = qv
    real = transformers.AutoProcessor  # (resolving the lazy attribute may replace sys.modules["transformers"]: patch the live object)
    transformers = sys.modules["transformers"]
    transformers.AutoProcessor = Rec
    out = {}
    try:
        data = dict(image="<IMAGE>", text="What is on the table?", question="What colour is the car?", video_name="<VIDEO>", video="<VIDEO>")
        for task in ("coco_caption", "synthdog", "gqa", "mmbench", "mme", "seed_bench", "vqav2", "mmvet", "vizwiz", "hr_bench", "textvqa", "msvd_qa", "mvbench"):
            mod = importlib.import_module(f"vispec.evaluation.{task}_prompt")
            for model in ("llava-hf/llava-v1.6-vicuna-7b-hf", "Qwen/Qwen2.5-VL-7B-Instruct"):
                Rec.log.clear()
                ret = mod.build_prompt(dict(data), SimpleNamespace(model=model))
                r = Rec.log[-1]
                out[f"{task}|{model}"] = dict(ctor=r.ctor, conversation=r.conv, template_kwargs=r.tmpl_kw, call=r.calls[-1], returned=ret)
        sqa = importlib.import_module("vispec.evaluation.scienceqa_prompt")
        rng = np.random.default_rng(1800)
        problems = {}
        for q in range(5):
            n_ch = int(rng.integers(2, 5))
            problems[str(q)] = dict(question=f"Question number {q}?", hint="" if q % 2 else f"A hint for {q}.", caption=f"a caption {q}",
                                    choices=[f"choice {q}.{i}" for i in range(n_ch)], answer=int(rng.integers(0, n_ch)), lecture=f"Lecture {q}.",
                                    solution="" if q == 3 else f"Solution {q}.", image=f"<IMAGE {q}>")
        for fmt in ("CQM-A", "QCM-LEA", "QCM-AL", "QCM-AE", "QCMLE-A", "QCLEM-A", "QCEM-A", "CQM-ELA", "QCML-AE", "QCM-ALE"):
            for use_caption in (False, True):
                Rec.log.clear()
                args = SimpleNamespace(model="Qwen/Qwen2.5-VL-7B-Instruct", use_caption=use_caption, options=["A", "B", "C", "D", "E"], prompt_format=fmt)
                ret = sqa.build_prompt(problems, ["0", "3", "1"], "4", args)
                r = Rec.log[-1]
                out[f"scienceqa|{fmt}|{int(use_caption)}"] = dict(ctor=r.ctor, conversation=r.conv, template_kwargs=r.tmpl_kw, call=r.calls[-1], returned=ret)
        out["scienceqa_problems"] = problems
    finally:
        transformers.AutoProcessor = real
        del sys.modules["qwen_vl_utils"]
    json.dump(out, open(os.path.join(OUT, "g18_prompts.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18"]
    fns = dict(g14=g14_tree_levels, g1=g1_imgadaptor, g2=g2_prefill, g3=g3_decode, g4=g4_topk, g5=g5_verify, g6=g6_posterior, g7=g7_posterior_sampling,
               g8=g8_loop, g9=g9_bf16, g10=g10_qwen, g11=g11_update_inference_inputs, g12=g12_kvcache, g13=g13_real_dims, g15=g15_qwen_rope_index, g16=g16_multi_image_repaired, g17=g17_multi_image_loop, g18=g18_prompts)
    for k in which:
        print("generating", k, flush=True)
        fns[k]()
    print("done")
