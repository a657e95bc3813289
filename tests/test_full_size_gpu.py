"""Full BASELINE sizes — every model of BASELINE.json's configs (bench.MODELS): LLaVA-v1.6-vicuna-7B (L = 2704 = 48 + 2144 image +
512 text tokens), LLaVA-v1.6-vicuna-13B (40 layers, 40 heads: the 8-GPU config's replica), Qwen2.5-VL-7B with a 4-image multi-turn
prompt (GQA 28/4, q/k/v bias, V = 152 064, multimodal rotary prefill, rope_delta) and Qwen2.5-VL-7B with fp8 target weights on a
1280x960 image.  The numpy oracle cannot run at these sizes in seconds, so parity here is size-independent properties: (1) the
reference's own invariant — speculative output == greedy AR output of the same target, token for token; (2) exact integer logic
replayed by the oracle on the device's buffers; (3) bookkeeping identities of the compressed draft KV and of the accept log."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import vo  # noqa: E402


# model -> (prompt length, image tokens, image runs)
SHAPES = {"llava7b": (2704, 2144, 1), "llava13b": (2704, 2144, 1), "qwen7b": (1584, 1024, 4), "qwen7b-fp8": (2124, 1564, 1)}


@pytest.fixture(scope="module", params=list(SHAPES))
def model_full(request):
    import gc
    import bench
    bench.MODEL = request.param
    sms, tcfg, _ = bench.build_models(torch.device("cuda:0"), 0, 0, 1, 1)
    yield sms[0], tcfg, request.param
    bench.MODEL = "llava7b"
    del sms
    gc.collect()
    torch.cuda.empty_cache()


def test_speculative_equals_greedy_ar_at_full_size(model_full):
    import bench
    sm, tcfg, name = model_full
    bench.MODEL = name
    ids, pix = bench.make_request(tcfg, 3, torch.device("cuda:0"))
    out, new_token, idx, acc = sm.specgenerate(ids, max_new_tokens=96, log=True, return_acceptance_len=True, **pix)
    L = ids.shape[1]
    L_want, n_img, n_runs = SHAPES[name]
    assert L == L_want and int((ids == tcfg.image_token_index).sum()) == n_img
    assert new_token > 96 and len(acc) == idx + 1
    assert out.shape[1] == L + sum(a + 1 for a in acc) == L + new_token            # accept log <-> token count
    assert 0 <= min(acc) and max(acc) <= sm.engine.depth + 1
    st = sm.engine.state()
    assert st["n_ctx"] == out.shape[1]
    assert st["draft_len"] == st["n_ctx"] - n_img + n_runs * (sm.engine.num_q - 1)  # every image run compressed to num_q-1 draft rows
    # exact tree logic on the device's own candidate lists at full vocabulary
    k, d = sm.engine.top_k, sm.engine.depth
    n_all = k + d * k * k
    sc = sm.engine.buffer("scores_all", (n_all,), torch.float32).cpu().numpy()
    tk = sm.engine.buffer("tokens_all", (n_all,), torch.int32).cpu().numpy()
    pa = sm.engine.buffer("parents_all", (1 + d * k,), torch.int32).cpu().numpy()
    tok, pos, mask, ret = sm.engine.tree()
    w_tok, w_ret, w_mask, w_pos = vo.build_tree(sc, tk.astype(np.int64), pa.astype(np.int64), tok[0], sm.engine.total_token - 1, k)
    np.testing.assert_array_equal(tok, w_tok)
    np.testing.assert_array_equal(mask, w_mask)
    np.testing.assert_array_equal(ret, w_ret)
    # greedy invariance: same target, same kernels at T = 1
    ar = sm.baseline_generate(ids, max_new_tokens=96, max_steps=97, **pix)
    n = min(ar.shape[1], out.shape[1])
    assert n >= L + 96
    np.testing.assert_array_equal(ar[0, :n].cpu().numpy(), out[0, :n].cpu().numpy())
    assert np.mean(acc) > 1.5  # the structured synthetic pair really exercises multi-token acceptance


def test_idempotent_and_request_independent(model_full):
    """Running the same request twice (KV buffers reused, state reset on the device) gives identical tokens; an interleaved
    different request does not leak into it."""
    import bench
    sm, tcfg, name = model_full
    bench.MODEL = name
    dev = torch.device("cuda:0")
    a_ids, a_pix = bench.make_request(tcfg, 5, dev)
    b_ids, b_pix = bench.make_request(tcfg, 6, dev)
    a1 = sm.specgenerate(a_ids, max_new_tokens=48, **a_pix)
    b1 = sm.specgenerate(b_ids, max_new_tokens=48, **b_pix)
    a2 = sm.specgenerate(a_ids, max_new_tokens=48, **a_pix)
    assert torch.equal(a1, a2) and not torch.equal(a1[:, -40:], b1[:, -40:])


def test_cohort_of_two_equals_the_single_requests_at_full_size(model_full):
    """Two requests on one weight pass (64-row GEMMs, m_tile mode, fused two-request attention) at the real model sizes: token for token
    the single-request results."""
    import bench
    from vispec_amd.model.spec_model_ours import specgenerate_cohort
    sm, tcfg, name = model_full
    bench.MODEL = name
    dev = torch.device("cuda:0")
    reqs = [bench.make_request(tcfg, 7, dev), bench.make_request(tcfg, 8, dev)]
    want = [sm.specgenerate(ids, max_new_tokens=64, log=True, return_acceptance_len=True, **pix) for ids, pix in reqs]
    mb = sm.make_cohort_member()
    got = specgenerate_cohort([sm, mb], reqs, max_new_tokens=64)
    for (toks, new_token, idx, acc), w in zip(got, want):
        assert torch.equal(toks, w[0]) and (new_token, idx, acc) == (w[1], w[2], w[3])
    del mb
