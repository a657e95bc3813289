"""Cohorts of five to eight requests (csrc/gemm_c8.h: eight activation tiles per weight pass, ONE accumulator chain per output element).
The c8 kernel gives up the bit-identity of a cohort row with the single-request kernel (whose K range is summed as four separately rounded
quarters): its rows differ from the single-request rows in fp32 rounding.  What these tests hold instead:
  * every epilogue against the fp64 product, rounded where the epilogues round, within the float bar of every other kernel (2^-6 of scale);
  * a request's rows are BIT-IDENTICAL whatever shares its weight pass: other tile, other cohort size, other live-row count;
  * whole loops: a request of a cohort of 5..8 returns the tokens it returns in any other cohort of 5..8, the speculative and the AR form of
    such a cohort agree token for token, and on the confident models of the loop tests those are also the single-request tokens and the
    oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402
from vispec_amd.model.spec_model_ours import baseline_generate_cohort, specgenerate_cohort, specgenerate_stream  # noqa: E402

from test_kernels_gpu import dev, engine, lib, p, packed, stream, tb  # noqa: E402,F401
from test_loop_gpu import IMG_TOK, build  # noqa: E402

# whole groups / an odd number of groups per split (11008 = 172 groups, S = 4: 43) / one group (K = 64) / a ragged last workgroup
# (1008 rows = 31.5 tiles, 96 rows = 3 tiles) / K not a multiple of a group (the fragment-shaped kernel) / the real layer shapes
C8_SHAPES = [(256, 256), (256, 704), (1008, 256), (96, 11008), (64, 64), (256, 144), (4096, 4096), (12288, 4096), (4096, 11008), (22016 // 2, 4096),
             (4608, 3584), (3584, 18944), (32064, 512)]


def bf(t):
    return t.to(torch.bfloat16).double()


def want_of(X, Wn, B, R, N, epi):
    acc = X.double() @ Wn.double().T + B.double()
    if epi == 0:
        return bf(acc)
    if epi == 1:
        return bf(R.double() + bf(acc))
    y, u = bf(acc[:, :N]), bf(acc[:, N:])
    return bf(bf(y / (1 + torch.exp(-y))) * u)


def X_lin(X, Wn, B):
    return torch.nan_to_num(X).double() @ Wn.double().T + B.double()


def assert_per_element(Y, want, epi, scale, fp8=False):
    """tests/test_kernels_gpu.py's per-element bar (assert_bf16_close with the arguments test_gemm_skinny / test_gemm_skinny_fp8 pass): 1 bf16 ulp
    of the ELEMENT (2 behind the SwiGLU chain), >= 97 % (90 %) of the elements bit-identical to the fp64 product rounded where the epilogue
    rounds.  One difference, measured in round 6: the reference here is the fp64 product (the single-request tests compare with the oracle's
    own fp32 accumulation), and the c8 order is ONE fp32 chain over the whole K — its accumulation noise, sqrt(K) x 2^-24 x the running sum's
    magnitude (6.7e-6 observed at K = 4096 on outputs of sigma 3), is an ABSOLUTE error that no summation order avoids and that only shows on
    elements which cancel to less than 2^-11 of the tensor's scale.  The absolute floor is therefore 2^-19 of the scale (a 1/500 of one bf16
    ulp of a typical element) instead of the single-request tests' fixed 1e-6."""
    from test_kernels_gpu import assert_bf16_close, fn
    w = want.float().cpu().numpy()
    floor = float(np.abs(w).max()) * 2.0 ** -19
    # (outlier_frac: at least ONE element of a small tile may sit in the tail — a 8 x 96 x 8 SwiGLU tile is 6 144 elements, 5e-5 of it is 0.3)
    kw = dict(outlier_frac=1e-3, outlier_mult=8) if (fp8 and epi == 2) else (dict(outlier_frac=max(5e-5, 1.01 / w.size), outlier_mult=3) if epi == 2 else {})
    assert_bf16_close(fn(Y), w, min_exact=0.90 if epi == 2 else 0.97, ulps=2 if epi == 2 else 1,
                      scale=None if scale is None else scale.float().cpu().numpy(), atol=max(floor, 1e-4 if epi == 2 else 1e-6), **kw)


def c8_cases():
    out = []
    for N, K in C8_SHAPES:
        for n_req, m_tile in [(8, 30), (5, 30), (7, 1), (6, 32), (8, 8)]:
            for epi in (0, 1, 2):
                if epi == 2 and N % 16:
                    continue
                if N * K > 2e7 and ((n_req, m_tile) != (8, 30) or epi == 1):
                    continue
                out.append((N, K, n_req, m_tile, epi))
    return out


@pytest.mark.parametrize("N,K,n_req,m_tile,epi", c8_cases())
def test_c8_gemm_against_the_fp64_product(lib, engine, N, K, n_req, m_tile, epi):
    rng = np.random.default_rng(N + 3 * K + 17 * n_req + m_tile + epi)
    rows = 2 * N if epi == 2 else N
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    X, W, B, R = tb(x), packed(w, swiglu=(epi == 2)), tb(b), tb(r)
    for t in range(n_req):
        X[32 * t + m_tile:32 * t + 32] = float("nan")  # rows outside a request's live rows may hold anything
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(W), None, p(B), p(Y), N, p(R), N, n_req, m_tile, N, K, epi))
    torch.cuda.synchronize()
    live = torch.zeros(32 * n_req, dtype=torch.bool, device=dev())
    for t in range(n_req):
        live[32 * t:32 * t + m_tile] = True
    want = want_of(torch.nan_to_num(X), tb(w), B, R, N, epi)
    err = (Y.double() - want)[live].abs().max().item()
    scale = want[live].abs().max().item()
    assert err <= scale * 2.0 ** -6, (err, scale)  # one bf16 ulp of the tensor's largest magnitude: the float bar of every GEMM test
    # ... and PER ELEMENT the bar the single-request kernel is held to (tests/test_kernels_gpu.py test_gemm_skinny): 1 bf16 ulp of the element
    # (2 behind the SwiGLU chain), >= 97 % (90 %) of the elements bit-identical to the fp64 product rounded where the epilogue rounds
    assert_per_element(Y[live], want[live], epi, None if epi != 1 else torch.maximum(R.double().abs(), bf(X_lin(X, tb(w), B)).abs())[live])
    assert (Y[~live].float() == 7.0).all(), "padding rows of a tile must stay untouched"


@pytest.mark.parametrize("N,K", [(256, 704), (1008, 256), (4096, 4096), (512, 11008), (64, 64), (256, 144)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_c8_rows_do_not_depend_on_what_shares_the_weight_pass(lib, engine, N, K, epi):
    """The same 30 rows as tile 2 of eight, as tile 4 of five next to other neighbours, and its first row as a one-row tile 6 of seven
    (the AR step's shape): bit for bit the same outputs."""
    if epi == 2 and N % 16:
        pytest.skip("SwiGLU needs N % 16 == 0")
    rng = np.random.default_rng(N + K + epi)
    rows = 2 * N if epi == 2 else N
    mine = synth.bf16_grid(rng.standard_normal((32, K), dtype=np.float32))
    rmine = synth.bf16_grid(rng.standard_normal((32, N), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32))
    W, B = packed(w, swiglu=(epi == 2)), tb(b)
    outs = []
    for n_req, tile, m_tile in [(8, 2, 30), (5, 4, 30), (7, 6, 1), (8, 7, 30)]:
        x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
        r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
        x[32 * tile:32 * tile + 32] = mine
        r[32 * tile:32 * tile + 32] = rmine
        X, R = tb(x), tb(r)
        Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
        L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(W), None, p(B), p(Y), N, p(R), N, n_req, m_tile, N, K, epi))
        torch.cuda.synchronize()
        outs.append(Y[32 * tile:32 * tile + m_tile].view(torch.int16).cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[3])
    np.testing.assert_array_equal(outs[0][:1], outs[2])


@pytest.mark.parametrize("N,K,epi", [(N, K, e) for N, K in [(256, 704), (4096, 3584), (1024, 18944), (96, 11008), (4608, 3584), (256, 256), (256, 160)]
                                     for e in (0, 1, 2) if not (e == 2 and N % 16)])
@pytest.mark.parametrize("n_req", [5, 8])
def test_c8_gemm_fp8_weights_against_the_fp64_product(lib, engine, N, K, n_req, epi):
    """W8A16 (e4m3 weights up-converted exactly, per-output-channel scale on the fp32 accumulator): against the fp64 product of the
    DEQUANTISED weights."""
    from vispec_amd.engine import pack_weight_fp8, quantize_fp8, swiglu_order
    m_tile = 30
    rng = np.random.default_rng(N + K + n_req + epi)
    rows = 2 * N if epi == 2 else N
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    q_u8, sc = quantize_fp8(tb(w))
    P8 = pack_weight_fp8(swiglu_order(q_u8) if epi == 2 else q_u8)
    X, B, R = tb(x), tb(b), tb(r)
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, n_req, m_tile, N, K, epi))
    torch.cuda.synchronize()
    wq = q_u8.view(torch.float8_e4m3fn).double() * sc.double()[:, None]
    acc = (X.double() @ wq.T) + B.double()
    if epi == 0:
        want = bf(acc)
    elif epi == 1:
        want = bf(R.double() + bf(acc))
    else:
        y, u = bf(acc[:, :N]), bf(acc[:, N:])
        want = bf(bf(y / (1 + torch.exp(-y))) * u)
    live = torch.zeros(32 * n_req, dtype=torch.bool, device=dev())
    for t in range(n_req):
        live[32 * t:32 * t + m_tile] = True
    err = (Y.double() - want)[live].abs().max().item()
    scale = want[live].abs().max().item()
    assert err <= scale * 2.0 ** -6, (err, scale)
    # per element, the bar of the single-request fp8 kernel (tests/test_kernels_gpu.py test_gemm_skinny_fp8)
    assert_per_element(Y[live], want[live], epi, None if epi != 1 else torch.maximum(R.double().abs(), bf(acc).abs())[live], fp8=True)
    assert (Y[~live].float() == 7.0).all()


def single(sm, ids, kw, **gen):
    return sm.specgenerate(ids, log=True, return_acceptance_len=True, **gen, **kw)


def make_requests(golden_dir, n, seed=91):
    rng = np.random.default_rng(seed)
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    reqs = [(torch.from_numpy(g["succ0_ids"])[None], {}), (torch.from_numpy(g["succ1_ids"])[None], {})]
    for i, ln in enumerate((17, 9, 23, 12, 20, 7, 15, 11, 14, 19)):
        if i % 3 == 1:  # an image prompt every third request
            n_img = 11 + i
            ids = np.concatenate([rng.integers(3, IMG_TOK, 4), np.full(n_img, IMG_TOK), rng.integers(3, IMG_TOK, ln)])
            feats = synth.bf16_grid(rng.standard_normal((n_img, T["D"]), dtype=np.float32) * 0.05)
            reqs.append((torch.from_numpy(ids)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda())))
        else:
            reqs.append((torch.from_numpy(rng.integers(3, IMG_TOK, size=ln))[None], {}))
    return reqs[:n], g


@pytest.mark.parametrize("n_req", [5, 8])
def test_cohort_of_five_and_eight_whole_loops(golden_dir, n_req):
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    models = [sm] + [sm.make_cohort_member() for _ in range(7)]
    reqs, g = make_requests(golden_dir, n_req)
    budgets = [30, 22, 41, 17, 26, 35, 12, 29][:n_req]  # ragged: the requests finish in different rounds and freeze one after the other
    got = specgenerate_cohort(models[:n_req], reqs, max_new_tokens=budgets)
    # (1) the cohort's composition does not matter: the requests in reverse order, through other tiles, next to other neighbours
    rev = specgenerate_cohort(models[:n_req], reqs[::-1], max_new_tokens=budgets[::-1])[::-1]
    for t, (a, b) in enumerate(zip(got, rev)):
        np.testing.assert_array_equal(a[0][0].cpu().numpy(), b[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert a[1:] == b[1:], f"request {t}"
    if n_req == 8:  # ... and the first five of them in a cohort of five
        five = specgenerate_cohort(models[:5], reqs[:5], max_new_tokens=budgets[:5])
        for t, (a, b) in enumerate(zip(got[:5], five)):
            np.testing.assert_array_equal(a[0][0].cpu().numpy(), b[0][0].cpu().numpy(), err_msg=f"request {t}")
            assert a[1:] == b[1:]
    # (2) speculative == greedy AR at the same batching, token for token
    ar = baseline_generate_cohort(models[:n_req], reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), a) in enumerate(zip(got, ar)):
        n = min(toks.shape[1], a.shape[1])
        assert n >= reqs[t][0].shape[1] + budgets[t]
        np.testing.assert_array_equal(toks[0, :n].cpu().numpy(), a[0, :n].cpu().numpy(), err_msg=f"request {t}: speculative != AR")
    # (3) on these (confident) models the c8 rounding changes no decision: the single-request tokens, accept lengths, round counts
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    # ... and request 0 is the oracle's / the reference fixture's stream
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, g["succ0_ids"], max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
    assert got[0][3] == o_acc


@pytest.mark.parametrize("n_slots,temperature", [(8, 0.0), (6, 0.0), (8, 6.0)])
def test_request_stream_through_eight_slots(golden_dir, n_slots, temperature):
    """Continuous batching over 6 / 8 slots: twelve ragged requests (text and image prompts), greedy and sampling; each returns what it returns
    alone, and the graphs are replayed."""
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    models = [sm] + [sm.make_cohort_member() for _ in range(n_slots - 1)]
    reqs, g = make_requests(golden_dir, 12, seed=97)
    budgets = [30, 12, 41, 8, 25, 33, 5, 19, 27, 16, 22, 9]
    seeds = list(range(40, 52))
    gen = dict(temperature=temperature, top_k=8) if temperature > 0 else {}
    want = [single(sm, *r, max_new_tokens=b, seed=sd, **gen) for r, b, sd in zip(reqs, budgets, seeds)]
    st = {}
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        got = specgenerate_stream(models, reqs, max_new_tokens=budgets, seeds=seeds, stats=st, **gen)
        side.synchronize()
    assert sm.engine.graph_stats()["replays"] > 0
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3]), f"request {t}"


@pytest.mark.parametrize("N,K", [(256, 256), (256, 704), (1008, 256), (96, 11008), (4096, 4096), (64, 64), (32064, 512)])
@pytest.mark.parametrize("n_req,rows", [(8, 8), (5, 5), (7, 1), (6, 8)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_two_tile_slab_rows_are_bit_identical_to_the_single_request_kernel(lib, engine, N, K, n_req, rows, epi):
    """The draft's GEMMs of a cohort of 5..8 (gemm_w32_kernel SLAB with two activation tiles: tile row 8 t + i is row 32 t + i of X / Y / R):
    the slab form keeps the single-request kernel's summation order, so its rows are that kernel's rows bit for bit."""
    if epi == 2 and N % 16:
        pytest.skip("SwiGLU needs N % 16 == 0")
    rng = np.random.default_rng(N + 3 * K + 17 * n_req + rows + epi)
    nrows = 2 * N if epi == 2 else N
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((nrows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(nrows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    X, W, B, R = tb(x), packed(w, swiglu=(epi == 2)), tb(b), tb(r)
    for t in range(n_req):
        X[32 * t + rows:32 * t + 32] = float("nan")
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(W), None, p(B), p(Y), N, p(R), N, n_req, -rows, N, K, epi))
    for t in range(n_req):
        Y1 = torch.full((32, N), 7.0, dtype=torch.bfloat16, device=dev())
        Xt, Rt = X[32 * t:32 * t + 32].contiguous(), R[32 * t:32 * t + 32].contiguous()
        L.check(lib.vispec_gemm_skinny(engine.h, stream(), p(Xt), K, p(W), p(B), p(Y1), N, p(Rt), N, rows, N, K, epi))
        torch.cuda.synchronize()
        got, want = Y[32 * t:32 * t + 32].view(torch.int16).cpu().numpy(), Y1.view(torch.int16).cpu().numpy()
        np.testing.assert_array_equal(got[:rows], want[:rows], err_msg=f"request {t}")
        assert (Y[32 * t + rows:32 * t + 32].float() == 7.0).all(), "rows outside the live rows must stay untouched"


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


@pytest.mark.parametrize("N,K", [(4608, 3584), (3584, 18944), (256, 704), (1008, 256), (96, 11008), (256, 256), (256, 128), (256, 192)])
@pytest.mark.parametrize("n_req,m_tile", [(8, 30), (5, 30), (7, 1)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_c8_gemm_fp8_activations_against_the_oracle(lib, eng8, N, K, n_req, m_tile, epi):
    """W8A8 on the cohort-8 kernel (v_mfma_scale_f32_32x32x64_f8f6f4, one accumulator chain): the oracle's Ops.linear(a8=True) per request,
    with the tolerance of tests/test_fp8a8_gpu.py's unit test."""
    from test_fp8a8_gpu import _weights
    from test_kernels_gpu import assert_bf16_close, fn
    if epi == 2 and N % 16:
        N = N // 16 * 16
    rng = np.random.default_rng(N + K + n_req + epi)
    P8, sc, Wt, rows = _weights(N, K, epi, rng)
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32) * 0.1)
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    X, B, R = tb(x), tb(b), tb(r)
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_fp8a8(eng8.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, n_req, m_tile, 0, N, K, epi, None, None, C.c_float(0)))
    torch.cuda.synchronize()
    o = vo.Ops(bf16=True)
    for t in range(n_req):
        xt, rt = x[32 * t:32 * t + m_tile], r[32 * t:32 * t + m_tile]
        if epi == 2:
            gu = o.linear(xt, Wt, b, a8=True)
            want = o.silu_mul(gu[:, :N], gu[:, N:])
        else:
            want = o.linear(xt, Wt, b, a8=True)
            if epi == 1:
                want = o.add(rt, want)
        floor = np.abs(want).max(axis=-1, keepdims=True) / 16
        # (outlier_frac: one element of a 30 x 1008 SwiGLU tile where the gate's and the up value's roundings both flip is 3.3e-5 of the tile)
        assert_bf16_close(fn(Y[32 * t:32 * t + m_tile]), want, min_exact=0.9, ulps=1 if epi == 0 else 2, outlier_frac=1e-4,
                          scale=floor if epi != 1 else np.maximum(floor, np.maximum(np.abs(rt), np.abs(want))))
        assert (Y[32 * t + m_tile:32 * t + 32].float() == 7.0).all()


@pytest.mark.parametrize("tree", [dict(total_token=20, depth=5, top_k=4), dict(total_token=12, depth=2, top_k=3), dict(total_token=30, depth=2, top_k=10)])
@pytest.mark.parametrize("n_req", [6, 8])
def test_cohorts_of_six_and_eight_with_other_tree_shapes(n_req, tree):
    """The draft's two-tile slab GEMMs with row counts other than the default's (top_k = 4 / 3 rows per level, depth + 2 = 7 / 4 catch-up rows)
    and a tree whose levels do not fit a slab (top_k = 10: the draft's GEMMs — lm_head included — then take the tile-per-request form, i.e.
    the cohort-8 kernel at m_tile = 10): composition independence, == the single requests, == the oracle."""
    import dataclasses
    sm, ot, od = build(50, 60, True, **tree)
    models = [sm] + [sm.make_cohort_member() for _ in range(n_req - 1)]
    rng = np.random.default_rng(193 + n_req)
    reqs = [(torch.from_numpy(rng.integers(3, IMG_TOK, size=n))[None], {}) for n in (17, 11, 23, 14, 9, 20, 12, 16)[:n_req]]
    budgets = [26, 19, 33, 12, 22, 15, 28, 18][:n_req]
    got = specgenerate_cohort(models, reqs, max_new_tokens=budgets)
    rev = specgenerate_cohort(models, reqs[::-1], max_new_tokens=budgets[::-1])[::-1]
    for t, (a, b) in enumerate(zip(got, rev)):
        np.testing.assert_array_equal(a[0][0].cpu().numpy(), b[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert a[1:] == b[1:]
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    od.cfg = dataclasses.replace(od.cfg, **tree)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, reqs[0][0][0].numpy(), max_new_tokens=budgets[0], max_pos=T["max_pos"])
    np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
    assert got[0][3] == o_acc


@pytest.mark.parametrize("total_token", [40, 60])
@pytest.mark.parametrize("n_req", [2, 3, 4])
def test_cohorts_with_trees_of_more_than_one_tile(golden_dir, n_req, total_token):
    """Trees of 33..64 nodes (the reference's `total_token = -1` autotune picks 40..60, spec_model_ours.py:179-201) INSIDE cohorts (round 6): a
    request then owns two activation tiles of the target-side workspaces (csrc: retarget_views), two / three / four such requests run their
    verify GEMMs on the wide (4 tiles) / cohort-8 (6, 8 tiles) kernels.  Ragged budgets; == the single-request runs == the oracle,
    composition-independent, speculative == AR; the tree size can be switched after the members exist and back."""
    import dataclasses
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    models = [sm] + [sm.make_cohort_member() for _ in range(n_req - 1)]  # built with 30-node trees ...
    sm.spec_layer.total_tokens = total_token - 1                         # ... then the leader's tree grows (autotune does this)
    reqs, g = make_requests(golden_dir, n_req)
    budgets = [30, 22, 41, 17][:n_req]
    got = specgenerate_cohort(models, reqs, max_new_tokens=budgets)      # (the members follow the leader's tree size)
    assert all(m.engine.total_token == total_token for m in models)
    rev = specgenerate_cohort(models, reqs[::-1], max_new_tokens=budgets[::-1])[::-1]
    for t, (a, b) in enumerate(zip(got, rev)):
        np.testing.assert_array_equal(a[0][0].cpu().numpy(), b[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert a[1:] == b[1:], f"request {t}"
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    ar = baseline_generate_cohort(models, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), a) in enumerate(zip(got, ar)):
        n = min(toks.shape[1], a.shape[1])
        assert n >= reqs[t][0].shape[1] + budgets[t]
        np.testing.assert_array_equal(toks[0, :n].cpu().numpy(), a[0, :n].cpu().numpy(), err_msg=f"request {t}: speculative != AR")
    od.cfg = dataclasses.replace(od.cfg, total_token=total_token)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, g["succ0_ids"], max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
    assert got[0][3] == o_acc and max(o_acc) >= 3
    # back to one tile per request: the same contexts, the 30-node results of the loop tests
    sm.spec_layer.total_tokens = 29
    back = specgenerate_cohort(models, reqs, max_new_tokens=budgets)
    od.cfg = dataclasses.replace(od.cfg, total_token=30)
    o30, _, _, acc30 = vo.specgenerate(ot, od, g["succ0_ids"], max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(back[0][0][0].cpu().numpy(), o30)
    assert back[0][3] == acc30


def test_wide_tree_cohort_with_sampling_seeds():
    """T > 0 (device sequential rejection with per-request counter-based uniforms) on 40-node trees in a cohort of three: every request's
    stream is its single-request stream for its own seed."""
    sm, ot, od = build(50, 60, True)
    models = [sm] + [sm.make_cohort_member() for _ in range(2)]
    sm.spec_layer.total_tokens = 39
    rng = np.random.default_rng(197)
    reqs = [(torch.from_numpy(rng.integers(3, T["V"], size=n))[None], {}) for n in (14, 19, 9)]
    seeds = [3, 5, 7]
    want = [sm.specgenerate(r[0], temperature=6.0, top_k=8, seed=sd, max_new_tokens=20, log=True, return_acceptance_len=True) for r, sd in zip(reqs, seeds)]
    got = specgenerate_cohort(models, reqs, temperature=6.0, top_k=8, seeds=seeds, max_new_tokens=20)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert acc == w[3]


def test_request_stream_with_wide_trees(golden_dir):
    """Continuous batching over three / four slots whose requests carry 48-node trees (two activation tiles each): every request returns what it
    returns alone; the graphs of the two-tile rounds are replayed."""
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    sm.spec_layer.total_tokens = 47
    reqs, g = make_requests(golden_dir, 9, seed=97)
    budgets = [30, 12, 41, 8, 25, 33, 5, 19, 27]
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    for n_slots in (3, 4):
        models = [sm] + [sm.make_cohort_member() for _ in range(n_slots - 1)]
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            got = specgenerate_stream(models, reqs, max_new_tokens=budgets)
            side.synchronize()
        for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
            np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"{n_slots} slots, request {t}")
            assert (new_token, idx, acc) == (w[1], w[2], w[3]), f"{n_slots} slots, request {t}"
        for m in models[1:]:
            m.engine.close()
    assert sm.engine.graph_stats()["replays"] > 0


def test_wide_tree_cohorts_are_at_most_four_requests():
    """Two tiles per request: a fifth request does not fit the eight tiles of a weight pass — refused loudly, by the library and by the host side."""
    sm, _, _ = build(50, 60, True)
    members = [sm.make_cohort_member() for _ in range(7)]
    members[2].engine.set_total_token(40)  # slots 1..3 can hold two tiles
    with pytest.raises(RuntimeError, match="first four request slots"):
        members[3].engine.set_total_token(33)  # slot 4
    sm.spec_layer.total_tokens = 47
    with pytest.raises(ValueError, match="at most four requests"):
        specgenerate_cohort([sm] + members[:4], [(torch.zeros(1, 4, dtype=torch.long), {})] * 5, max_new_tokens=4)
