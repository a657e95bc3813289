"""The reference's STAGE fixtures that until round 5 pinned only the oracle — g1 (ImgAdaptor.forward), g2 (Model.forward, prefill / compression
branch), g3 (Model.forward, decode branch), g5 (KV-Llama prefill + tree verify), g10 (Qwen2.5-VL text model: multimodal rotary prefill + tree
verify) — fed to the HIP path through the C-ABI and compared with the FIXTURE itself (the reference's fp32 CPU run, captured by
tests/golden/gen_golden.py); tests/test_fixtures_gpu.py does the same for g4, g6, g7, g11, g13, g14.

Bars: floats within 2^-6 of the tensor's largest magnitude (the kernels hold bf16 weights and activations, like the reference on a GPU; the
fixtures are fp32), integers exact.  What each test can reach through the product's entry points is said in its docstring — the library has
no call that runs ONE draft forward on caller-chosen rows, so a stage is driven through vispec_draft_prefill / vispec_draft_round /
vispec_target_forward with inputs arranged so that those calls compute exactly the fixture's forward."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import synth  # noqa: E402
from vispec_amd.engine import DraftConfig, TargetConfig  # noqa: E402
from vispec_amd.model import SpecModel  # noqa: E402

from test_loop_gpu import IMG_TOK  # noqa: E402

D = T["D"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def close(got, want, frac=2.0 ** -6):
    np.testing.assert_allclose(np.asarray(got, np.float32), want, rtol=0, atol=frac * float(np.abs(want).max()))


def t_(a, dtype=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def build_pair(seed_t, seed_d, num_q=2, embed_row=None):
    """tiny LLaVA-NeXT-shaped pair with the FIXTURE generator's weights (tests/golden/gen_golden.py: synth.make_*_weights with these seeds).
    embed_row = (token, vector): the draft's embedding of `token` is replaced by `vector` — Model.forward receives inputs_embeds already shifted
    by one with the sampled token's embedding as last row (cnets_ours.py:1081-1082), the fixtures hold an arbitrary last row; with this the
    library's own shift reproduces the fixture's input exactly."""
    tw = synth.make_target_weights(D, T["H"], T["I"], T["V"], T["NL"], seed=seed_t)
    dw = synth.make_draft_weights(D, T["H"], T["I"], T["V"], num_q=num_q, seed=seed_d)
    if embed_row is not None:
        dw = dict(dw)
        e = np.array(dw["embed_tokens.weight"], copy=True)
        e[embed_row[0]] = embed_row[1]
        dw["embed_tokens.weight"] = e
    tcfg = TargetConfig(hidden_size=D, num_heads=T["H"], num_kv_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"], num_layers=T["NL"],
                        max_position_embeddings=T["max_pos"], architectures=("LlavaNextForConditionalGeneration",), image_token_index=IMG_TOK)
    dcfg = DraftConfig(hidden_size=D, num_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"], max_position_embeddings=T["max_pos"])
    return SpecModel.from_weights(tcfg, dcfg, tw, dw, num_q=num_q)


def dev_state(eng):
    return eng.buffer("state", (24,), torch.int32).cpu().numpy()  # DevState: [10] draft_len, [11] draft_real_len (csrc/kernels.h)


def run_prefill(sm, hidden_fwd, embeds_fwd, mask, first_tok=7):
    """vispec_draft_prefill on exactly the tensors Model.forward saw in the fixture: forward's inputs_embeds row i = the library's embeds row
    i + 1 (the library shifts), its last row = embed(first token) (build_pair's embed_row)."""
    L = hidden_fwd.shape[0]
    eng = sm.engine
    E = np.concatenate([np.zeros((1, D), np.float32), embeds_fwd[:-1]], 0)
    eng.begin_request(np.arange(3, 3 + L, dtype=np.int32), 200)
    sm.spec_layer.reset_kv()
    eng.draft_prefill(t_(hidden_fwd), t_(E), None if mask is None else np.asarray(mask, bool), torch.tensor([first_tok], dtype=torch.int32).cuda())
    torch.cuda.synchronize()
    return eng


def draft_kv_rows(eng, n):
    kv = eng.draft_kv.float().cpu().numpy()  # [2, H, max_pos, 128]
    return kv[0][:, :n], kv[1][:, :n]


@pytest.mark.parametrize("q", [2, 5])
def test_g1_imgadaptor_fixture_through_the_draft_prefill(golden_dir, q):
    """ImgAdaptor.forward (cnets_ours.py:603-661) on the fixture's 37 image embeddings: a prompt whose (shifted) mask has one 37-row image run
    carrying exactly those rows; the adaptor's q outputs are the run's q - 1 compressed tokens (the draft's compressed input rows) and the global
    feature g."""
    g = load(golden_dir, "g1_imgadaptor.npz")
    x = synth.bf16_grid(g[f"x_q{q}"][0])
    N, pre, post = x.shape[0], 3, 4
    L = pre + N + post
    sm = build_pair(20, 11, num_q=q)
    rng = np.random.default_rng(5)
    hidden = synth.bf16_grid(rng.standard_normal((L, D), dtype=np.float32))
    E = synth.bf16_grid(rng.standard_normal((L, D), dtype=np.float32) * 0.05)
    mask = np.zeros(L, bool)
    mask[pre + 1:pre + 1 + N] = True  # the draft sees mask[1:] and embeds[1:] (cnets_ours.py:880, 1081): image rows pre .. pre + N - 1 of the shifted sequence
    E[mask] = x
    eng = sm.engine
    eng.begin_request(np.arange(3, 3 + L, dtype=np.int32), 200)
    sm.spec_layer.reset_kv()
    eng.draft_prefill(t_(hidden), t_(E), mask, torch.tensor([7], dtype=torch.int32).cuda())
    torch.cuda.synchronize()
    y = g[f"y_q{q}"]
    close(eng.buffer("draft_g", (D,)).float().cpu().numpy(), y[-1])                      # last output = the new global feature (:930)
    xc = eng.buffer("draft_xc", (L, D)).float().cpu().numpy()
    close(xc[pre:pre + q - 1], y[:q - 1])                                                  # first q - 1 outputs = the compressed tokens (:928-929)
    assert eng.state()["draft_len"] == L - N + (q - 1)


@pytest.mark.parametrize("tag,q", [("img_q2", 2), ("img_q5", 5), ("txt", 2)])
def test_g2_prefill_fixture_through_the_draft_prefill(golden_dir, tag, q):
    """Model.forward, prefill / compression branch (cnets_ours.py:879-975, 1020-1023): last hidden row (the only row topK_genrate consumes,
    :1109), the compressed K / V cache, real_len and the global feature against the fixture."""
    g = load(golden_dir, "g2_prefill.npz")
    hidden, emb = synth.bf16_grid(g[f"{tag}_hidden"]), synth.bf16_grid(g[f"{tag}_embeds"])
    mask = g[f"{tag}_mask"] if tag != "txt" else None
    sm = build_pair(20, 12, num_q=q, embed_row=(7, emb[-1]))
    eng = run_prefill(sm, hidden, emb, mask)
    k, v = g[f"{tag}_k"], g[f"{tag}_v"]
    n_c = k.shape[1]
    st = dev_state(eng)
    assert st[10] == n_c and st[11] == int(g[f"{tag}_real_len"])
    gk, gv = draft_kv_rows(eng, n_c)
    close(gk, k)
    close(gv, v)
    close(eng.buffer("draft_g", (D,)).float().cpu().numpy(), g[f"{tag}_g"][0])
    close(eng.buffer("draft_last", (16, D))[0].float().cpu().numpy(), g[f"{tag}_out"][-1])


def test_g3_decode_fixture_catch_up_forward_through_the_draft_round(golden_dir):
    """Model.forward, decode branch (cnets_ours.py:976-988 + the layer): the fixture's first decode call — three accepted hidden rows with their
    token ids at explicit positions real_len .. real_len + 2, causal over the compressed cache — is the catch-up forward of vispec_draft_round:
    the rows are staged the way the accept step stages them (a three-node chain tree whose nodes carry h2, all accepted), then the round runs.
    Checked against the fixture: the K / V rows the catch-up appends behind the compressed prompt, and its last hidden row.  (The fixture's two
    tree-level forwards take rows the library chooses itself — its top-k — so they are not reachable with the fixture's inputs; the level
    forwards are held to the reference by g14 in tests/test_fixtures_gpu.py.)"""
    g = load(golden_dir, "g3_decode.npz")
    hidden, emb, mask = synth.bf16_grid(g["hidden"]), synth.bf16_grid(g["embeds"]), g["mask"]
    sm = build_pair(20, 13, embed_row=(7, emb[-1]))
    eng = run_prefill(sm, hidden, emb, mask)
    n_c = int(dev_state(eng)[10])
    h2, ids2 = synth.bf16_grid(g["h2"]), g["ids2"].astype(np.int32)
    a1 = h2.shape[0]
    assert n_c == g["k4"].shape[1] - a1 - 2 * 8
    k_pref, v_pref = draft_kv_rows(eng, n_c)
    close(k_pref, g["k4"][:, :n_c])
    close(v_pref, g["v4"][:, :n_c])
    # accepted rows: node j carries h2[j]; the token each row pairs with is ids2[j] (cnets_ours.py:1084: ids shifted by one)
    eng.set_total_token(a1)
    chain = np.arange(a1, dtype=np.int32)
    eng.set_tree(np.concatenate([[7], ids2[:-1]]).astype(np.int32), chain, np.array([(1 << (j + 1)) - 1 for j in range(a1)], np.uint64), chain[None])
    hn = eng.buffer("hidden_new", (32, D))
    hn[:a1] = t_(h2)
    am = eng.buffer("am", (32,), torch.int32)
    am[:a1] = torch.from_numpy(ids2).to(am.device)
    eng.accept()
    assert eng.last_accept() == (0, a1 - 1)
    eng.set_total_token(30)
    eng.draft_round()
    torch.cuda.synchronize()
    close(eng.buffer("draft_last", (16, D))[0].float().cpu().numpy(), g["o2"][-1])
    gk, gv = draft_kv_rows(eng, n_c + a1)
    close(gk[:, n_c:], g["k4"][:, n_c:n_c + a1])
    close(gv[:, n_c:], g["v4"][:, n_c:n_c + a1])
    assert int(dev_state(eng)[11]) == hidden.shape[0] + a1  # real_len advanced by the catch-up rows (:416-418)


def tree_bits(tm):
    return np.array([sum(1 << j for j in range(tm.shape[1]) if tm[i, j] > 0) for i in range(tm.shape[0])], np.uint64)


def test_g5_verify_fixture_through_the_target_forward(golden_dir):
    """KV-Llama prefill (PyTorch-ROCm prefill of the product) and the tree verify forward (vispec_target_forward: all skinny GEMMs, tree
    attention with the fixture's 9-node mask, positions tree_pos + L) against the fixture: prefill logits / hidden, verify logits / hidden,
    the K rows of layer 0 and V rows of layer 1 over prompt + tree, and the KV lengths."""
    g = load(golden_dir, "g5_verify.npz")
    sm = build_pair(21, 14)
    ids = g["ids"]
    L = len(ids)
    eng = sm.engine
    emb = torch.nn.functional.embedding(torch.from_numpy(ids).cuda(), sm.base_model.w.embed).to(torch.bfloat16).contiguous()
    logits, hidden = sm.base_model.prefill(emb, all_logits=True)
    close(hidden.float().cpu().numpy(), g["prefill_hidden"])
    close(logits.float().cpu().numpy()[:, :g["prefill_logits"].shape[1]], g["prefill_logits"])
    sm._start_request(torch.from_numpy(ids)[None], None, {}, max_new_tokens=64)
    cand, Tn = g["cand"], len(g["cand"])
    eng.set_total_token(Tn)
    eng.set_tree(cand.astype(np.int32), g["tree_pos"].astype(np.int32), tree_bits(g["tree_mask"]), None)
    eng.target_forward()
    torch.cuda.synchronize()
    V = T["V"]
    close(eng.buffer("hidden_new", (32, D))[:Tn].float().cpu().numpy(), g["hidden"])
    close(eng.buffer("logits", (32, V))[:Tn].float().cpu().numpy()[:, :g["logits"].shape[1]], g["logits"])
    kv = eng.target_kv.float().cpu().numpy()  # [2 * NL, 1, H, max_pos, hd]
    n = L + Tn
    close(kv[0, 0, :, :n], g["k0"])
    close(kv[3, 0, :, :n], g["v1"])
    assert (g["cur"] == n).all()  # the reference's KVCache lengths after the verify; the library's rows [L, n) are exactly the tree's (checked above)


def test_g10_qwen_fixture_through_the_target_forward(golden_dir):
    """Qwen2.5-VL text model (GQA 4 / 2, q/k/v bias, theta 1e6, multimodal rotary with an image block): the product's prefill on the fixture's
    inputs_embeds at the fixture's 3-component positions, then the tree verify at rope_delta-shifted positions, against the fixture."""
    Q = synth.QWEN_TINY
    g = load(golden_dir, "g10_qwen.npz")
    tw = synth.make_target_weights(Q["D"], Q["H"], Q["I"], Q["V"], Q["NL"], seed=90, qkv_bias=True, H_kv=Q["Hkv"])
    dw = synth.make_draft_weights(Q["D"], Q["H"], Q["I"], Q["V"], seed=92, qkv_bias=True)
    IMG = Q["V"] - 1
    tcfg = TargetConfig(hidden_size=Q["D"], num_heads=Q["H"], num_kv_heads=Q["Hkv"], intermediate_size=Q["I"], vocab_size=Q["V"], num_layers=Q["NL"],
                        max_position_embeddings=Q["max_pos"], rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True,
                        architectures=("Qwen2_5_VLForConditionalGeneration",), image_token_index=IMG, attn_impl="sdpa", mrope_section=Q["mrope_section"])
    dcfg = DraftConfig(hidden_size=Q["D"], num_heads=Q["H"], intermediate_size=Q["I"], vocab_size=Q["V"], max_position_embeddings=Q["max_pos"],
                       rms_norm_eps=Q["eps"], rope_theta=Q["theta"], qkv_bias=True)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw)
    eng = sm.engine
    ids, emb = g["ids"], g["emb"]
    L = len(ids)
    logits, hidden = sm.base_model.prefill(t_(emb), all_logits=True, position_ids=torch.from_numpy(g["pos3"]))
    close(hidden.float().cpu().numpy(), g["prefill_hidden"])
    close(logits.float().cpu().numpy(), g["prefill_logits"])
    sm._start_request(torch.from_numpy(ids)[None], t_(emb)[None], dict(image_grid_thw=torch.tensor([(1, 6, 8)])), max_new_tokens=64)
    assert sm._rope_delta == int(g["delta"])
    cand, Tn = g["cand"], len(g["cand"])
    eng.set_total_token(Tn)
    eng.set_tree(cand.astype(np.int32), g["tree_pos"].astype(np.int32), tree_bits(g["tree_mask"]), None)
    eng.target_forward()
    torch.cuda.synchronize()
    close(eng.buffer("hidden_new", (32, Q["D"]))[:Tn].float().cpu().numpy(), g["hidden"])
    close(eng.buffer("logits", (32, Q["V"]))[:Tn].float().cpu().numpy(), g["logits"])
    kv = eng.target_kv.float().cpu().numpy()
    n = L + Tn
    close(kv[0, 0, :, :n], g["k0"])
    close(kv[3, 0, :, :n], g["v1"])
    assert (g["cur"] == n).all()
