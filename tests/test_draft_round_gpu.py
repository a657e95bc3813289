"""GPU parity of the FLOAT half of the draft (cnets_ours.py:1090-1165) against the numpy oracle, level by level.

tests/test_loop_gpu.py checks the draft's integer logic exactly and its floats only through the token streams they produce.
Here the values themselves are compared: after the draft prefill and after one decode round the device's last hidden row, the
per-level LM-head -> log-softmax -> top-k outputs (`scores_all` / `tokens_all`: bf16 cumulative log-probs and token ids) and the last
level's hidden rows are checked against the oracle.  Near-ties make a top-k SELECTION ambiguous between two correct bf16
implementations (a different accumulation order moves a logit by an ulp), so the oracle is replayed along the DEVICE's own
selections (recovered from `parents_all`) and every selection is checked to be optimal up to the tolerance:
  * value of every kept (token, score) pair == oracle log-prob of that token (+ the oracle score of its parent), within TOL;
  * no token the oracle ranks above the device's k-th pick by more than TOL is missing (sets equal where the oracle separates
    neighbours by more than the tolerance);
  * the frontier the device carries to the next level is, by the oracle's values, within TOL of the best k of the k*k children.
TOL = 2^-6 of the row's largest |log-prob| (two bf16 ulps at the top of the range; a bf16 log-prob of magnitude 8..16 has ulp 2^-4).
Models: LLaVA-tiny (random and structured pair, image prompt), Qwen2.5-VL-tiny (q/k/v bias, two image runs), fp8 target weights
(the draft's head is the target's fp8 lm_head)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import synth  # noqa: E402
from vispec_amd.engine import DraftConfig, TargetConfig  # noqa: E402
from vispec_amd.model import SpecModel  # noqa: E402

from test_loop_gpu import build, check_tree_exact  # noqa: E402


def tol_of(row):
    return 2.0 ** -6 * float(np.max(np.abs(row)))


def check_topk_row(lp_row, dev_tok, dev_val, add, k, worst):
    """One top-k row: lp_row = oracle log-softmax [V]; dev_tok/dev_val = the device's k picks and their (cumulative) scores."""
    tol = tol_of(lp_row) + (2.0 ** -6 * abs(add))
    assert len(set(dev_tok.tolist())) == k
    want = vo.bf16_round(lp_row[dev_tok] + np.float32(add))
    err = np.abs(dev_val - want)
    worst[0] = max(worst[0], float(err.max() / tol))
    np.testing.assert_allclose(dev_val, want, rtol=0, atol=tol)
    assert np.all(np.diff(dev_val) <= 0), "device picks are sorted by value, descending"
    kth = float(np.min(lp_row[dev_tok]))
    missing = np.setdiff1d(np.nonzero(lp_row > kth + tol)[0], dev_tok)
    assert missing.size == 0, f"tokens {missing[:5]} beat the device's k-th pick by more than the tolerance"


def replay_levels(od, head_w, last, kv, len_posi, eng, worst):
    """Replay topK_genrate's tree growth (cnets_ours.py:1109-1165) in the oracle along the device's selections."""
    k, depth = eng.top_k, eng.depth
    n_all = k + depth * k * k
    sc = eng.buffer("scores_all", (n_all,), torch.float32).cpu().numpy()
    tk = eng.buffer("tokens_all", (n_all,), torch.int32).cpu().numpy().astype(np.int64)
    pa = eng.buffer("parents_all", (1 + depth * k,), torch.int32).cpu().numpy().astype(np.int64)
    o = od.ops
    lp0 = o.log_softmax(o.linear(last, head_w))[0]
    check_topk_row(lp0, tk[:k], sc[:k], 0.0, k, worst)
    scores = vo.bf16_round(lp0[tk[:k]])
    in_ids, in_h, tmask = tk[:k], np.repeat(last, k, axis=0), np.eye(k, dtype=bool)
    assert pa[0] == 0
    out = None
    for i in range(depth):
        bias = 1 + k * k * max(0, i - 1) + (k if i > 0 else 0)
        cs_idx = pa[1 + i * k: 1 + (i + 1) * k] - bias  # the frontier this level expands = the previous level's selection
        if i == 0:
            np.testing.assert_array_equal(cs_idx, np.arange(k))
        pos = np.full(k, len_posi + i, np.int64)
        out, kv = od.forward_decode(in_h, in_ids, kv, pos=pos, tree_mask=tmask)
        lp = o.log_softmax(o.linear(out, head_w))
        base = k + i * k * k
        dtok, dcu = tk[base: base + k * k].reshape(k, k), sc[base: base + k * k].reshape(k, k)
        for r in range(k):
            check_topk_row(lp[r], dtok[r], dcu[r], float(scores[r]), k, worst)
        cu_o = vo.bf16_round(np.take_along_axis(lp, dtok, axis=1) + scores[:, None]).reshape(-1)
        if i + 1 < depth:
            nbias = 1 + k * k * i + k
            nxt = pa[1 + (i + 1) * k: 1 + (i + 2) * k] - nbias
            assert len(set(nxt.tolist())) == k and nxt.min() >= 0 and nxt.max() < k * k
            tol = tol_of(cu_o)
            kth_best = np.sort(cu_o)[-k]
            assert cu_o[nxt].min() >= kth_best - tol, "the carried frontier is within the tolerance of the best k children"
            scores, in_ids = cu_o[nxt], dtok.reshape(-1)[nxt]
            out_ids = nxt // k
            in_h = out[out_ids]
            tmask = np.concatenate([tmask[out_ids], np.eye(k, dtype=bool)], axis=1)  # a node inherits its PARENT'S row (cnets_ours.py:1163-1165)
    dout = eng.buffer("draft_out", (64, eng.dcfg.hidden_size))[:k].float().cpu().numpy()
    np.testing.assert_allclose(dout, out, rtol=0, atol=2.0 ** -6 * np.abs(out).max())


def run_case(sm, ot, od, ids, head_w, feats=None, grids=None, mask=None):
    eng = sm.engine
    D = eng.tcfg.hidden_size
    kw = {}
    if feats is not None:
        kw["pixel_values"] = torch.from_numpy(feats).to(torch.bfloat16).cuda()
    if grids is not None:
        kw["image_grid_thw"] = torch.tensor(grids)
    hidden, demb, mask_np, first = sm._start_request(torch.from_numpy(ids)[None], None, kw, max_new_tokens=64)
    L = len(ids)
    first_tok = int(first.cpu()[0])
    # ---- draft prefill: oracle on the DEVICE's target hidden states, so that only the draft's arithmetic is compared
    h_np, e_np = hidden.float().cpu().numpy(), demb.float().cpu().numpy()
    e_shift = np.concatenate([e_np[1:], od.ops.rd(od.w["embed_tokens.weight"][[first_tok]])], 0)  # cnets_ours.py:1081-1082
    od.reset_kv()
    out_c, kv, _ = od.forward_prefill(h_np, e_shift, None if mask_np is None else mask_np.astype(bool))
    last = out_c[-1:]
    dlast = eng.buffer("draft_last", (16, D))[:1].float().cpu().numpy()
    np.testing.assert_allclose(dlast, last, rtol=0, atol=2.0 ** -6 * np.abs(last).max())
    worst = [0.0]
    replay_levels(od, head_w, last, kv, L, eng, worst)
    check_tree_exact(eng)
    # ---- one verify + accept, then the DECODE round of the draft (catch-up rows + tree), again fed with the device's own inputs
    eng.verify_accept(-1)
    st = eng.state()
    a, n = st["accept_len"], st["n_ctx"]
    acc_h = eng.buffer("accept_hidden", (16, D))[: a + 1].float().cpu().numpy()
    dids = eng.buffer("draft_ids", (16,), torch.int32)[: a + 1].cpu().numpy().astype(np.int64)
    eng.draft_round()
    out2, kv2 = od.forward_decode(acc_h, dids, kv)
    assert kv2[2] == n and kv2[0].shape[1] == kv[0].shape[1] + a + 1  # real length / compressed length after the catch-up
    last2 = out2[-1:]
    dlast2 = eng.buffer("draft_last", (16, D))[:1].float().cpu().numpy()
    np.testing.assert_allclose(dlast2, last2, rtol=0, atol=2.0 ** -6 * np.abs(last2).max())
    replay_levels(od, head_w, last2, kv2, n, eng, worst)
    check_tree_exact(eng)
    assert eng.state()["draft_len"] == kv2[0].shape[1]
    return worst[0]


@pytest.mark.parametrize("structured,seeds", [(False, (21, 13)), (True, (50, 60))])
def test_llava_tiny_draft_floats(structured, seeds):
    sm, ot, od = build(seeds[0], seeds[1], structured, arch="LlavaNextForConditionalGeneration")
    rng = np.random.default_rng(31)
    IMG = T["V"] - 1
    ids = np.concatenate([rng.integers(3, IMG, 7), np.full(19, IMG), rng.integers(3, IMG, 11)])
    feats = synth.bf16_grid(rng.standard_normal((19, T["D"]), dtype=np.float32) * 0.05)
    worst = run_case(sm, ot, od, ids, ot.lm_head, feats=feats)
    assert worst <= 1.0
    print(f"worst draft log-prob error: {worst:.2f} of the tolerance")


def test_llava_tiny_text_only_draft_floats():
    sm, ot, od = build(21, 13, False)
    ids = np.random.default_rng(32).integers(3, T["V"], size=26)
    assert run_case(sm, ot, od, ids, ot.lm_head) <= 1.0


def test_qwen_tiny_draft_floats():
    from test_loop_gpu import build_qwen
    sm, ot, od, IMG = build_qwen()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(33)
    grids = [(1, 6, 8), (1, 4, 4)]
    ids = np.concatenate([rng.integers(3, IMG, 5), np.full(12, IMG), rng.integers(3, IMG, 4), np.full(4, IMG), rng.integers(3, IMG, 7)])
    feats = synth.bf16_grid(rng.standard_normal((16, Q["D"]), dtype=np.float32) * 0.05)
    assert run_case(sm, ot, od, ids, ot.lm_head, feats=feats, grids=grids) <= 1.0


def test_fp8_target_draft_floats():
    """fp8 target: the draft's LM head is the target's e4m3 lm_head (W8A16 GEMM with per-row scales)."""
    from test_loop_gpu import build_qwen_fp8
    sm, ot, od, IMG = build_qwen_fp8()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(34)
    grids = [(1, 6, 8)]
    ids = np.concatenate([rng.integers(3, IMG, 6), np.full(12, IMG), rng.integers(3, IMG, 8)])
    feats = synth.bf16_grid(rng.standard_normal((12, Q["D"]), dtype=np.float32) * 0.05)
    assert run_case(sm, ot, od, ids, ot.lm_head, feats=feats, grids=grids) <= 1.0
