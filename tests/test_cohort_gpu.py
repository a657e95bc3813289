"""Cohorts: two independent requests run their draft-and-verify rounds in lockstep on ONE weight pass (every GEMM of a round is
launched once on 64 activation rows, tile t = request t; trees, accept decisions, KV caches, attention and round state stay per
request).  Each request must produce exactly what it produces alone — tokens, accept lengths, round counts — and that is what the
oracle / reference fixtures say."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import synth  # noqa: E402
from vispec_amd.model.spec_model_ours import specgenerate_cohort  # noqa: E402

from test_loop_gpu import IMG_TOK, build  # noqa: E402


def single(sm, ids, kw, **gen):
    return sm.specgenerate(ids, log=True, return_acceptance_len=True, **gen, **kw)


@pytest.mark.parametrize("structured", [True, False])
def test_cohort_equals_two_single_requests(golden_dir, structured):
    sm, ot, od = build(50, 60, structured, arch="LlavaNextForConditionalGeneration")
    mb = sm.make_cohort_member()
    assert mb.engine.tw is sm.engine.tw and mb.engine.target_kv.data_ptr() != sm.engine.target_kv.data_ptr()
    rng = np.random.default_rng(71)
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids_a = g["succ0_ids"]
    n_img = 23
    ids_b = np.concatenate([rng.integers(3, IMG_TOK, 5), np.full(n_img, IMG_TOK), rng.integers(3, IMG_TOK, 9)])
    feats = synth.bf16_grid(rng.standard_normal((n_img, T["D"]), dtype=np.float32) * 0.05)
    req_a = (torch.from_numpy(ids_a)[None], {})
    req_b = (torch.from_numpy(ids_b)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda()))
    want_a = single(sm, *req_a, max_new_tokens=30)
    want_b = single(sm, *req_b, max_new_tokens=30)
    got = specgenerate_cohort([sm, mb], [req_a, req_b], max_new_tokens=30)
    for (toks, new_token, idx, acc), want in zip(got, (want_a, want_b)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), want[0][0].cpu().numpy())
        assert (new_token, idx, acc) == (want[1], want[2], want[3])
    # ... and that is the oracle's / the reference fixture's stream
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids_a, max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
    assert got[0][3] == o_acc
    if structured:  # the committed output of the reference itself (max_new_tokens = 40 there): same stream
        n = min(len(o_out), len(g["succ0_out"]))
        np.testing.assert_array_equal(o_out[:n], g["succ0_out"][:n])
    # swapped roles and a second pair on the same contexts (state fully reset per request)
    got2 = specgenerate_cohort([sm, mb], [req_b, req_a], max_new_tokens=30)
    np.testing.assert_array_equal(got2[0][0][0].cpu().numpy(), want_b[0][0].cpu().numpy())
    np.testing.assert_array_equal(got2[1][0][0].cpu().numpy(), want_a[0][0].cpu().numpy())
    # the leader still serves single requests
    again = single(sm, *req_a, max_new_tokens=30)
    np.testing.assert_array_equal(again[0][0].cpu().numpy(), want_a[0][0].cpu().numpy())


def test_cohort_request_that_finishes_early_is_frozen(golden_dir):
    """One request stops long before the other (a second stop token, spec_model_ours.py:540-542): it is frozen on the device while the
    other completes, and both results equal the single-request runs."""
    from types import SimpleNamespace
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    ids_a, ids_b = g["succ0_ids"], g["succ1_ids"]
    sm, ot, od = build(50, 60, True)
    mb = sm.make_cohort_member()
    full_a = sm.specgenerate(torch.from_numpy(ids_a)[None], max_new_tokens=60)[0].cpu().numpy()
    eot = int(full_a[len(ids_a) + 9])
    sm.tokenizer = SimpleNamespace(eos_token_id=2, convert_tokens_to_ids=lambda t: eot)
    mb.tokenizer = SimpleNamespace(eos_token_id=2, convert_tokens_to_ids=lambda t: -1)
    req_a, req_b = (torch.from_numpy(ids_a)[None], {}), (torch.from_numpy(ids_b)[None], {})
    want_a = single(sm, *req_a, max_new_tokens=60, is_llama3=True)
    want_b = single(mb, *req_b, max_new_tokens=60, is_llama3=True)
    assert want_a[2] + 5 < want_b[2]  # a stops many rounds before b
    got = specgenerate_cohort([sm, mb], [req_a, req_b], max_new_tokens=60, is_llama3=True)
    for (toks, new_token, idx, acc), want in zip(got, (want_a, want_b)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), want[0][0].cpu().numpy())
        assert (new_token, idx, acc) == (want[1], want[2], want[3])
    assert sm.engine.state()["n_ctx"] == want_a[0].shape[1]  # frozen: nothing moved after its last round


def test_cohort_on_a_side_stream_replays_graphs(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    sm, _, _ = build(50, 60, True)
    mb = sm.make_cohort_member()
    req_a, req_b = (torch.from_numpy(g["succ0_ids"])[None], {}), (torch.from_numpy(g["succ1_ids"])[None], {})
    want = [single(sm, *req_a, max_new_tokens=24), single(sm, *req_b, max_new_tokens=24)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got = specgenerate_cohort([sm, mb], [req_a, req_b], max_new_tokens=24)
        s.synchronize()
    for (toks, new_token, idx, acc), w in zip(got, want):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy())
        assert acc == w[3]
    gs = sm.engine.graph_stats()
    assert gs["replays"] > 0 and gs["captures"] >= 2


def test_member_requires_its_leader():
    sm, _, _ = build(50, 60, True)
    other, _, _ = build(50, 60, True)
    mb = sm.make_cohort_member()
    with pytest.raises(RuntimeError, match="member of the first"):
        other.engine.cohort_round(mb.engine)


@pytest.mark.parametrize("fp8", [False, True])
def test_cohort_qwen_tiny_gqa_bias_mrope_and_fp8(fp8):
    """Qwen2.5-VL-shaped pair (GQA 4/2, q/k/v bias, two-kernel q|k|v + rotary path below the fused threshold, rope_delta per request,
    multimodal rotary prefill) and its fp8 variant: the cohort reproduces the two single runs."""
    from test_loop_gpu import build_qwen, build_qwen_fp8
    sm, ot, od, IMG = build_qwen_fp8() if fp8 else build_qwen()
    mb = sm.make_cohort_member()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(81)
    reqs = []
    for grids, segs in (([(1, 6, 8), (1, 4, 4)], (4, 3, 6)), ([(1, 4, 8)], (7, 5))):
        parts = []
        for gi, g3 in enumerate(grids):
            parts += [rng.integers(3, IMG, segs[gi]), np.full(g3[1] * g3[2] // 4, IMG)]
        parts.append(rng.integers(3, IMG, segs[-1]))
        ids = np.concatenate(parts)
        n_img = int((ids == IMG).sum())
        feats = synth.bf16_grid(rng.standard_normal((n_img, Q["D"]), dtype=np.float32) * 0.05)
        reqs.append((torch.from_numpy(ids)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(), image_grid_thw=torch.tensor(grids))))
    want = [single(sm, *r, max_new_tokens=24) for r in reqs]
    got = specgenerate_cohort([sm, mb], reqs, max_new_tokens=24)
    for (toks, new_token, idx, acc), w in zip(got, want):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy())
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    assert max(max(g[3]) for g in got) >= 3


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("n_req", [2, 3])
def test_cohort_qwen_tiny_with_wide_trees(fp8, n_req):
    """The same Qwen2.5-VL-shaped pair with 44-node trees (two activation tiles per request, round 6): the two-kernel q|k|v + rotary path with
    64-row request strides, GQA, rope_delta per request, W8A16 on the wide (4 tiles) and cohort-8 (6 tiles) kernels: == the single runs."""
    from test_loop_gpu import build_qwen, build_qwen_fp8
    sm, ot, od, IMG = build_qwen_fp8() if fp8 else build_qwen()
    members = [sm.make_cohort_member() for _ in range(n_req - 1)]
    sm.spec_layer.total_tokens = 43
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(83)
    reqs = []
    for grids, segs in (([(1, 6, 8), (1, 4, 4)], (4, 3, 6)), ([(1, 4, 8)], (7, 5)), ([(1, 4, 4)], (9, 4)))[:n_req]:
        parts = []
        for gi, g3 in enumerate(grids):
            parts += [rng.integers(3, IMG, segs[gi]), np.full(g3[1] * g3[2] // 4, IMG)]
        parts.append(rng.integers(3, IMG, segs[-1]))
        ids = np.concatenate(parts)
        n_img = int((ids == IMG).sum())
        feats = synth.bf16_grid(rng.standard_normal((n_img, Q["D"]), dtype=np.float32) * 0.05)
        reqs.append((torch.from_numpy(ids)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(), image_grid_thw=torch.tensor(grids))))
    want = [single(sm, *r, max_new_tokens=24) for r in reqs]
    got = specgenerate_cohort([sm] + members, reqs, max_new_tokens=24)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    for m in members:
        m.engine.close()


def test_cohort_with_sampling_uses_each_requests_seed():
    sm, ot, od = build(50, 60, True)
    mb = sm.make_cohort_member()
    rng = np.random.default_rng(82)
    reqs = [(torch.from_numpy(rng.integers(3, T["V"], size=14))[None], {}), (torch.from_numpy(rng.integers(3, T["V"], size=19))[None], {})]
    want = [sm.specgenerate(r[0], temperature=6.0, top_k=8, seed=sd, max_new_tokens=20, log=True, return_acceptance_len=True) for r, sd in zip(reqs, (3, 5))]
    got = specgenerate_cohort([sm, mb], reqs, temperature=6.0, top_k=8, seeds=[3, 5], max_new_tokens=20)
    for (toks, new_token, idx, acc), w in zip(got, want):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy())
        assert acc == w[3]
