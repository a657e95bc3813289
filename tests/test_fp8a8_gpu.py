"""W8A8 (round 4): fp8 (e4m3) weights AND fp8 activations for the target's q|k|v, gate|up and down GEMMs (o_proj keeps bf16 activations), multiplied on v_mfma_scale_f32_32x32x64_f8f6f4 —
what BASELINE config 5 literally names ("fp8 weights (CDNA4 fp8 MFMA)").  The reference has no fp8 path; the arithmetic is defined by the
oracle's `Ops.linear(..., a8=True)` (per-row dynamic scale sx = max|x| / 448, q = e4m3(x / sx), exact products, fp32 accumulation) and SURVEY.md
§7.1 step 8's bar: same accepted tokens as that oracle / documented divergence (tests/test_fp8_activation_study.py prices W8A8 against W8A16).
Checked here: (1) the GEMM at unit level against the oracle, every epilogue, split-K + fused norm, one and two activation tiles; (2) cohort rows
BIT-IDENTICAL to single-request rows (the cohort contract of every other dtype); (3) the whole draft-and-verify loop of the Qwen2.5-VL-shaped tiny
model: token streams and accept lengths == the oracle's A8 loop, == greedy AR with the same kernels, cohort == single."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import vo  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402

from test_kernels_gpu import assert_bf16_close, dev, fn, lib, p, stream, tb  # noqa: E402,F401
from test_loop_gpu import build_qwen_fp8  # noqa: E402


@pytest.fixture(scope="module")
def eng8():
    """An engine whose quantisation scratch is wide enough for the test shapes (K up to 18944): Qwen-tiny with a wide MLP."""
    from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights
    D, H, I, V, NL = 512, 4, 18944, 1024, 1
    tcfg = TargetConfig(hidden_size=D, num_heads=H, num_kv_heads=2, intermediate_size=I, vocab_size=V, num_layers=NL, max_position_embeddings=512)
    dcfg = DraftConfig(hidden_size=D, num_heads=H, intermediate_size=704, vocab_size=V, max_position_embeddings=512)
    tw = TargetWeights.from_state_dict(tcfg, synth.make_target_weights(D, H, I, V, NL, seed=0, H_kv=2), dev())
    dw = DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(D, H, 704, V, seed=1), 2, dev())
    return Engine(tcfg, dcfg, tw, dw)


def _weights(N, K, epi, rng):
    from vispec_amd.engine import pack_weight_fp8, quantize_fp8, swiglu_order
    rows = 2 * N if epi == 2 else N
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    q_u8, sc = quantize_fp8(tb(w))
    P8 = pack_weight_fp8(swiglu_order(q_u8) if epi == 2 else q_u8)
    codes = q_u8.view(torch.float8_e4m3fn).float().cpu().numpy()
    return P8, sc, (codes, sc.cpu().numpy()), rows


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (30, 4608, 3584), (30, 1024, 18944), (7, 96, 11008), (30, 256, 704), (60, 4096, 3584), (33, 1008, 704)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_fp8a8_gemm_against_the_oracle(lib, eng8, M, N, K, epi):
    if epi == 2 and N % 16:
        N = N // 16 * 16
    rng = np.random.default_rng(M + N + K + epi)
    P8, sc, Wt, rows = _weights(N, K, epi, rng)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32) * 0.1)
    r = synth.bf16_grid(rng.standard_normal((M, N), dtype=np.float32))
    X, B, R = tb(x), tb(b), tb(r)
    Y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_fp8a8(eng8.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, 1, 0, M, N, K, epi, None, None, C.c_float(0)))
    torch.cuda.synchronize()
    o = vo.Ops(bf16=True)
    if epi == 2:
        gu = o.linear(x, Wt, b, a8=True)
        want = o.silu_mul(gu[:, :N], gu[:, N:])
    else:
        want = o.linear(x, Wt, b, a8=True)
        if epi == 1:
            want = o.add(r, want)
    # tolerance: one bf16 ulp at the element's own magnitude, but never below one ulp at 1/16 of the row's largest output — the fp8 MFMA sums
    # its 64 products per instruction in its own order, so an output that is a near-cancellation of K terms carries the absolute error of the
    # row's scale, not of its own (tiny) value (measured: 8.6e-5 on a -4e-4 output in a row of magnitude 8)
    floor = np.abs(want).max(axis=-1, keepdims=True) / 16
    # (a chained epilogue — + residual, SwiGLU — doubles a first-rounding flip: 2 ulp, as for every other dtype's chained epilogues)
    assert_bf16_close(fn(Y), want, min_exact=0.9, ulps=1 if epi == 0 else 2, scale=floor if epi != 1 else np.maximum(floor, np.maximum(np.abs(r), np.abs(want))))


@pytest.mark.parametrize("M,N,K", [(30, 3584, 3584), (30, 512, 18944), (5, 256, 704)])
def test_fp8a8_split_k_gemm_with_fused_norm_against_the_oracle(lib, eng8, M, N, K):
    """The o_proj / down_proj form: split-K partials (scaled per row and per channel), reduce + bias + residual + the RMSNorm that follows."""
    rng = np.random.default_rng(M + N + K)
    P8, sc, Wt, _ = _weights(N, K, 0, rng)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((M, N), dtype=np.float32))
    nw = synth.bf16_grid(1.0 + 0.1 * rng.standard_normal(N, dtype=np.float32))
    X, R, NW = tb(x), tb(r), tb(nw)
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    Yn = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_fp8a8(eng8.h, stream(), p(X), K, p(P8), p(sc), None, p(Y), N, p(R), N, 1, 0, M, N, K, 1, p(NW), p(Yn), C.c_float(1e-6)))
    torch.cuda.synchronize()
    o = vo.Ops(bf16=True)
    h = o.add(r, o.linear(x, Wt, None, a8=True))
    assert_bf16_close(fn(Y), h, min_exact=0.9, ulps=2, scale=np.maximum(np.abs(h).max(axis=-1, keepdims=True) / 16, np.maximum(np.abs(r), np.abs(h))))
    assert_bf16_close(fn(Yn), o.rmsnorm(fn(Y), nw, 1e-6), min_exact=0.9, ulps=2)
    # the reduce also left the normed rows QUANTISED in the ctx's scratch (the next GEMM's input: target_forward's fused form of
    # quant_rows_e4m3_kernel): exactly the oracle's quantisation of the normed rows the device wrote
    codes = torch.zeros(M, N, dtype=torch.uint8, device=dev())
    scales = torch.zeros(M, dtype=torch.float32, device=dev())
    L.check(lib.vispec_a8_scratch_read(eng8.h, stream(), p(codes), p(scales), M, N))
    torch.cuda.synchronize()
    yn = fn(Yn)
    sx = (np.maximum(np.abs(yn).max(axis=-1, keepdims=True), np.float32(1e-12)) / np.float32(448.0)).astype(np.float32)
    np.testing.assert_array_equal(scales.cpu().numpy(), sx[:, 0])
    np.testing.assert_array_equal(codes.view(torch.float8_e4m3fn).float().cpu().numpy(), vo.e4m3_round((yn / sx).astype(np.float32)))


@pytest.mark.parametrize("N,K", [(4608, 3584), (3584, 18944), (256, 704), (1008, 256), (96, 11008), (256, 256)])
@pytest.mark.parametrize("n_req,m_tile,rb", [(4, 30, 4), (3, 30, 4), (4, 1, 4), (2, 30, 4), (4, 30, 8), (3, 30, 8), (4, 1, 8)])  # (pairs: the paired-tile kernel)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_fp8a8_cohort_rows_are_bit_identical_to_the_single_request_rows(lib, eng8, N, K, n_req, m_tile, epi, rb):
    """rb = 8: the eight-row-block form (gemm_w32_wide8_kernel<.., 2, ..>; K = 256 has quarters shorter than a group and falls back to four)."""
    if epi == 2 and N % 16:
        N = N // 16 * 16
    eng8.set_wide_row_blocks(rb)
    rng = np.random.default_rng(N + K + n_req + m_tile + epi)
    P8, sc, _, rows = _weights(N, K, epi, rng)
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32) * 0.1)
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    X, B, R = tb(x), tb(b), tb(r)
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_fp8a8(eng8.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, n_req, m_tile, 0, N, K, epi, None, None, C.c_float(0)))
    for t in range(n_req):
        Y1 = torch.full((32, N), 7.0, dtype=torch.bfloat16, device=dev())
        Xt, Rt = X[32 * t:32 * t + 32].contiguous(), R[32 * t:32 * t + 32].contiguous()
        L.check(lib.vispec_gemm_fp8a8(eng8.h, stream(), p(Xt), K, p(P8), p(sc), p(B), p(Y1), N, p(Rt), N, 1, 0, m_tile, N, K, epi, None, None, C.c_float(0)))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(Y[32 * t:32 * t + m_tile].view(torch.int16).cpu().numpy(), Y1[:m_tile].view(torch.int16).cpu().numpy(), err_msg=f"request {t}")
        assert (Y[32 * t + m_tile:32 * t + 32].float() == 7.0).all(), "padding rows of a tile must stay untouched"
    eng8.set_wide_row_blocks(4)


@pytest.mark.parametrize("M,N,K,bias", [(200, 4608, 3584, True), (75, 1408, 704, False), (33, 512, 18944, False)])
def test_fp8a8_prefill_linear_against_the_oracle(lib, M, N, K, bias):
    """The PREFILL form of the W8A8 linears (vispec_amd/model/target.py: scaled_linear(q8=...)): activations quantised by the library's kernel
    (vispec_quant_rows_e4m3 == the decode path's), product on the library's fp8 x fp8 GEMM (torch._scaled_mm, row-wise scales) -> the oracle's
    Ops.linear(a8=True); the quantisation itself exact (codes and scales)."""
    from vispec_amd.engine import quantize_fp8
    from vispec_amd.model.target import scaled_linear
    rng = np.random.default_rng(M + N + K)
    w = synth.bf16_grid(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    b = synth.bf16_grid(rng.standard_normal(N, dtype=np.float32) * 0.1) if bias else None
    q_u8, sc = quantize_fp8(tb(w))
    codes = q_u8.view(torch.float8_e4m3fn).float().cpu().numpy()
    X = tb(x)
    qx = torch.zeros(M, K, dtype=torch.uint8, device=dev())
    sx = torch.zeros(M, dtype=torch.float32, device=dev())
    L.check(lib.vispec_quant_rows_e4m3(None, stream(), p(X), K, p(qx), K, p(sx), M, K))
    torch.cuda.synchronize()
    want_sx = (np.maximum(np.abs(x).max(axis=-1, keepdims=True), np.float32(1e-12)) / np.float32(448.0)).astype(np.float32)
    np.testing.assert_array_equal(sx.cpu().numpy(), want_sx[:, 0])
    np.testing.assert_array_equal(qx.view(torch.float8_e4m3fn).float().cpu().numpy(), vo.e4m3_round((x / want_sx).astype(np.float32)))
    y = scaled_linear(X, q_u8.view(torch.float8_e4m3fn).to(torch.bfloat16), None if b is None else tb(b), sc, q_u8)
    want = vo.Ops(bf16=True).linear(x, (codes, sc.cpu().numpy()), b, a8=True)
    assert_bf16_close(fn(y), want, min_exact=0.9, ulps=2, scale=np.abs(want).max(axis=-1, keepdims=True) / 16)


def _a8_model():
    """The Qwen2.5-VL-shaped tiny model of tests/test_loop_gpu.py with fp8 weights, switched to fp8 activations (product and oracle alike)."""
    sm, ot, od, IMG = build_qwen_fp8()
    L.check(sm.engine.lib.vispec_set_fp8_activations(sm.engine.h, 1))
    sm.engine.target_weight_dtype = "fp8a8"
    ot.a8_decode = True
    return sm, ot, od, IMG


def test_fp8a8_loop_matches_the_oracles_a8_loop_and_greedy_ar():
    sm, ot, od, IMG = _a8_model()
    Q = synth.QWEN_TINY
    rng = np.random.default_rng(19)
    grids = [(1, 6, 8)]
    ids = np.concatenate([rng.integers(3, IMG, 6), np.full(12, IMG), rng.integers(3, IMG, 9)])
    mask = ids == IMG
    feats = synth.bf16_grid(rng.standard_normal((int(mask.sum()), Q["D"]), dtype=np.float32) * 0.05)
    kw = dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda(), image_grid_thw=torch.tensor(grids))
    out, new_token, idx, acc = sm.specgenerate(torch.from_numpy(ids)[None], max_new_tokens=32, log=True, return_acceptance_len=True, **kw)
    pos3, delta = synth.qwen_rope_index(ids, IMG, grids)
    emb = ot.w["model.embed_tokens.weight"][ids].copy()
    emb[mask] = feats
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, ids, inputs_embeds=emb, image_mask=mask, max_new_tokens=32, max_pos=Q["max_pos"], position_ids=pos3,
                                                 rope_delta=delta)
    np.testing.assert_array_equal(out[0].cpu().numpy(), o_out)
    assert acc == o_acc and max(acc) >= 2
    ar = sm.baseline_generate(torch.from_numpy(ids)[None], max_new_tokens=28, **kw)
    n = min(ar.shape[1], len(o_out))
    np.testing.assert_array_equal(ar[0, :n].cpu().numpy(), o_out[:n])


def test_fp8a8_cohort_of_four_equals_the_single_requests():
    from vispec_amd.model.spec_model_ours import specgenerate_cohort
    sm, ot, od, IMG = _a8_model()
    members = [sm.make_cohort_member() for _ in range(3)]
    for m in members:
        L.check(m.engine.lib.vispec_set_fp8_activations(m.engine.h, 1))
    rng = np.random.default_rng(23)
    reqs = [(torch.from_numpy(rng.integers(3, IMG, n))[None], {}) for n in (17, 9, 26, 12)]
    budgets = [24, 30, 16, 21]
    want = [sm.specgenerate(ids, max_new_tokens=b, log=True, return_acceptance_len=True, **kw) for (ids, kw), b in zip(reqs, budgets)]
    got = specgenerate_cohort([sm] + members, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])


@pytest.mark.parametrize("n_req", [2, 4])
def test_fp8a8_cohort_with_wide_trees_equals_the_single_requests(n_req):
    """W8A8 with trees of 44 nodes inside cohorts (round 6: two activation tiles per request — the quantisation scratch, the fp8 q|k|v epilogue
    with per-tile row counts, the split-K reduce that re-quantises the normed rows): token for token the single-request results."""
    from vispec_amd.model.spec_model_ours import specgenerate_cohort
    sm, ot, od, IMG = _a8_model()
    members = [sm.make_cohort_member() for _ in range(n_req - 1)]
    for m in members:
        L.check(m.engine.lib.vispec_set_fp8_activations(m.engine.h, 1))
    sm.spec_layer.total_tokens = 43
    rng = np.random.default_rng(29)
    reqs = [(torch.from_numpy(rng.integers(3, IMG, n))[None], {}) for n in (17, 9, 26, 12)[:n_req]]
    budgets = [24, 30, 16, 21][:n_req]
    want = [sm.specgenerate(ids, max_new_tokens=b, log=True, return_acceptance_len=True, **kw) for (ids, kw), b in zip(reqs, budgets)]
    got = specgenerate_cohort([sm] + members, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    for m in members:
        m.engine.close()
