"""Wide cohorts: three or four independent requests share one weight pass (csrc/gemm_wide.h: 16 waves = 4 weight row blocks x 4
K-quarters sharing each staged activation group).  The bar is the cohort bar of round 2: every request's rows come out BIT-IDENTICAL to
the single-request kernel's, so a request in a cohort of four produces exactly the tokens it produces alone — and those are the
oracle's / the reference fixtures'."""
import ctypes as C
import dataclasses
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402
from vispec_amd.model.spec_model_ours import specgenerate_cohort  # noqa: E402

from test_kernels_gpu import dev, engine, lib, p, packed, stream, tb  # noqa: E402,F401
from test_loop_gpu import IMG_TOK, build  # noqa: E402


# shapes: whole groups only / leftover k-steps (704 = 44 steps: quarters of 11) / split-K with uneven quarters (11008) / a ragged last
# workgroup (1008 rows = 31.5 tiles; 96 rows = 3 tiles) / one k-step per quarter (K = 64) / the real layer shapes (LLaVA-7B, Qwen2.5-VL-7B)
WIDE_SHAPES = [(256, 256), (256, 704), (1008, 256), (96, 11008), (4096, 4096), (12288, 4096), (4096, 11008), (64, 64), (32064, 512),
               (22016 // 2, 4096), (4608, 3584), (3584, 18944)]


def wide_cases():
    """(N, K, n_req, m_tile, epi, row_blocks): every launch shape of vispec_set_wide_row_blocks (8 = csrc/gemm_wide.h's eight-row-block kernel;
    shapes whose K-quarters hold less than one 64-k group fall back to four row blocks inside the library) — small shapes with every row
    count and epilogue, the large (bench) shapes at the bench row count, 2 / 3 row blocks at the bench row counts and a one-row tile."""
    out = []
    for N, K in WIDE_SHAPES:
        for n_req, m_tile in [(3, 30), (4, 30), (4, 8), (3, 1), (4, 32)]:
            for epi in (0, 1, 2):
                for rb in (4, 3, 2, 8, 84):
                    if epi == 2 and N % 16:
                        continue
                    if N * K > 2e7 and (m_tile != 30 or epi == 1):
                        continue
                    if rb in (2, 3) and (n_req, m_tile) not in ((4, 30), (3, 30), (3, 1)):
                        continue
                    if rb == 8 and (n_req, m_tile) not in ((4, 30), (3, 30), (3, 1), (4, 32)):
                        continue
                    if rb == 84 and ((n_req, m_tile) != (4, 30) or epi == 1):  # the per-shape policy: the bench row count only
                        continue
                    out.append((N, K, n_req, m_tile, epi, rb))
    return out


@pytest.mark.parametrize("N,K,n_req,m_tile,epi,row_blocks", wide_cases())
def test_wide_gemm_rows_are_bit_identical_to_the_single_request_kernel(lib, engine, N, K, n_req, m_tile, epi, row_blocks):
    L.check(lib.vispec_set_wide_row_blocks(engine.h, row_blocks))
    rng = np.random.default_rng(N + 3 * K + 17 * n_req + m_tile + epi)
    rows = 2 * N if epi == 2 else N
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    X, W, B, R = tb(x), packed(w, swiglu=(epi == 2)), tb(b), tb(r)
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(W), None, p(B), p(Y), N, p(R), N, n_req, m_tile, N, K, epi))
    for t in range(n_req):
        Y1 = torch.full((32, N), 7.0, dtype=torch.bfloat16, device=dev())
        Xt, Rt = X[32 * t:32 * t + 32].contiguous(), R[32 * t:32 * t + 32].contiguous()
        L.check(lib.vispec_gemm_skinny(engine.h, stream(), p(Xt), K, p(W), p(B), p(Y1), N, p(Rt), N, m_tile, N, K, epi))
        torch.cuda.synchronize()
        got, want = Y[32 * t:32 * t + 32].view(torch.int16).cpu().numpy(), Y1.view(torch.int16).cpu().numpy()
        np.testing.assert_array_equal(got[:m_tile], want[:m_tile], err_msg=f"request {t}")
        assert (Y[32 * t + m_tile:32 * t + 32].float() == 7.0).all(), "padding rows of a tile must stay untouched"
    L.check(lib.vispec_set_wide_row_blocks(engine.h, 4))


@pytest.mark.parametrize("N,K,epi", [(N, K, e) for N, K in [(256, 704), (4096, 3584), (1024, 18944), (96, 11008), (4608, 3584), (256, 256)]
                                     for e in (0, 1, 2) if not (e == 2 and N % 16)])
@pytest.mark.parametrize("n_req", [3, 4])
@pytest.mark.parametrize("row_blocks", [4, 0, 8, 84])
def test_wide_gemm_fp8_rows_are_bit_identical_to_the_single_request_kernel(lib, engine, N, K, n_req, epi, row_blocks):
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
    L.check(lib.vispec_set_wide_row_blocks(engine.h, row_blocks))
    L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, n_req, m_tile, N, K, epi))
    L.check(lib.vispec_set_wide_row_blocks(engine.h, 4))
    for t in range(n_req):
        Y1 = torch.full((32, N), 7.0, dtype=torch.bfloat16, device=dev())
        Xt, Rt = X[32 * t:32 * t + 32].contiguous(), R[32 * t:32 * t + 32].contiguous()
        L.check(lib.vispec_gemm_skinny_fp8(engine.h, stream(), p(Xt), K, p(P8), p(sc), p(B), p(Y1), N, p(Rt), N, m_tile, N, K, epi))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(Y[32 * t:32 * t + m_tile].view(torch.int16).cpu().numpy(), Y1[:m_tile].view(torch.int16).cpu().numpy())


# The draft's GEMMs of a cohort (csrc/kernels.h, gemm_w32_kernel SLAB): the requests have at most 8 live rows each (top_k rows of a tree
# level, depth + 2 catch-up rows, the root row) and share ONE activation tile — tile row 8 t + i is row 32 t + i of X / Y / R.  Same bar:
# row for row bit-identical to the single-request launch; nothing outside the live rows is written.
def slab_cases():
    out = []
    for N, K in [(256, 256), (256, 704), (1008, 256), (96, 11008), (4096, 4096), (4096, 8192), (12288, 4096), (4096, 11008), (64, 64), (32064, 512),
                 (22016 // 2, 4096)]:
        for n_req, rows in [(4, 8), (4, 5), (3, 8), (2, 8), (4, 1), (2, 3)]:
            for epi in (0, 1, 2):
                if epi == 2 and N % 16:
                    continue
                if N * K > 2e7 and ((n_req, rows) not in ((4, 8), (4, 5)) or epi == 1):  # large shapes: the bench row counts only
                    continue
                out.append((N, K, n_req, rows, epi))
    return out


@pytest.mark.parametrize("N,K,n_req,rows,epi", slab_cases())
def test_slab_gemm_rows_are_bit_identical_to_the_single_request_kernel(lib, engine, N, K, n_req, rows, epi):
    rng = np.random.default_rng(N + 3 * K + 17 * n_req + rows + epi)
    nrows = 2 * N if epi == 2 else N
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((nrows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(nrows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    X, W, B, R = tb(x), packed(w, swiglu=(epi == 2)), tb(b), tb(r)
    for t in range(n_req):
        X[32 * t + rows:32 * t + 32] = float("nan")  # rows outside a request's live rows may hold anything
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


@pytest.mark.parametrize("N,K,epi", [(N, K, e) for N, K in [(256, 704), (4096, 3584), (1024, 18944), (96, 11008), (152064 // 8, 512)]
                                     for e in (0, 1, 2) if not (e == 2 and N % 16)])
@pytest.mark.parametrize("n_req,rows", [(4, 8), (3, 1)])
def test_slab_gemm_fp8_rows_are_bit_identical_to_the_single_request_kernel(lib, engine, N, K, n_req, rows, epi):
    from vispec_amd.engine import pack_weight_fp8, quantize_fp8, swiglu_order
    rng = np.random.default_rng(N + K + n_req + epi)
    nrows = 2 * N if epi == 2 else N
    x = synth.bf16_grid(rng.standard_normal((32 * n_req, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((nrows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(nrows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((32 * n_req, N), dtype=np.float32))
    q_u8, sc = quantize_fp8(tb(w))
    P8 = pack_weight_fp8(swiglu_order(q_u8) if epi == 2 else q_u8)
    X, B, R = tb(x), tb(b), tb(r)
    Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_cohort(engine.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, n_req, -rows, N, K, epi))
    for t in range(n_req):
        Y1 = torch.full((32, N), 7.0, dtype=torch.bfloat16, device=dev())
        Xt, Rt = X[32 * t:32 * t + 32].contiguous(), R[32 * t:32 * t + 32].contiguous()
        L.check(lib.vispec_gemm_skinny_fp8(engine.h, stream(), p(Xt), K, p(P8), p(sc), p(B), p(Y1), N, p(Rt), N, rows, N, K, epi))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(Y[32 * t:32 * t + rows].view(torch.int16).cpu().numpy(), Y1[:rows].view(torch.int16).cpu().numpy())
        assert (Y[32 * t + rows:32 * t + 32].float() == 7.0).all()


def single(sm, ids, kw, **gen):
    return sm.specgenerate(ids, log=True, return_acceptance_len=True, **gen, **kw)


@pytest.mark.parametrize("n_req", [3, 4])
def test_cohort_of_three_and_four_equals_the_single_requests(golden_dir, n_req):
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    sm.engine.set_wide_row_blocks(0 if n_req == 3 else 4)  # both launch shapes of the wide GEMMs go through a whole request loop
    members = [sm.make_cohort_member() for _ in range(n_req - 1)]
    rng = np.random.default_rng(91)
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    n_img = 23
    ids_img = np.concatenate([rng.integers(3, IMG_TOK, 5), np.full(n_img, IMG_TOK), rng.integers(3, IMG_TOK, 9)])
    feats = synth.bf16_grid(rng.standard_normal((n_img, T["D"]), dtype=np.float32) * 0.05)
    reqs = [(torch.from_numpy(g["succ0_ids"])[None], {}),
            (torch.from_numpy(ids_img)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda())),
            (torch.from_numpy(g["succ1_ids"])[None], {}),
            (torch.from_numpy(rng.integers(3, IMG_TOK, 21))[None], {})][:n_req]
    budgets = [30, 22, 41, 17][:n_req]  # ragged: the requests finish in different rounds and freeze one after the other
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    models = [sm] + members
    got = specgenerate_cohort(models, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    # ... and request 0 is the oracle's / the reference fixture's stream
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, g["succ0_ids"], max_new_tokens=30, max_pos=T["max_pos"])
    np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
    assert got[0][3] == o_acc
    n = min(len(o_out), len(g["succ0_out"]))
    np.testing.assert_array_equal(o_out[:n], g["succ0_out"][:n])
    # a smaller round on the same contexts afterwards (the graph cache is keyed by the member set), then the leader alone
    got2 = specgenerate_cohort(models[:2], [reqs[1], reqs[0]], max_new_tokens=[budgets[1], budgets[0]])
    np.testing.assert_array_equal(got2[0][0][0].cpu().numpy(), want[1][0][0].cpu().numpy())
    np.testing.assert_array_equal(got2[1][0][0].cpu().numpy(), want[0][0][0].cpu().numpy())
    again = single(sm, *reqs[0], max_new_tokens=budgets[0])
    np.testing.assert_array_equal(again[0][0].cpu().numpy(), want[0][0][0].cpu().numpy())


@pytest.mark.parametrize("tree", [dict(total_token=20, depth=5, top_k=4), dict(total_token=12, depth=2, top_k=3), dict(total_token=30, depth=2, top_k=10)])
@pytest.mark.parametrize("n_req", [2, 4])
def test_cohorts_with_other_tree_shapes_equal_the_single_requests(n_req, tree):
    """The draft's slab GEMMs with row counts other than the default's (top_k = 4 / 3 rows per level, depth + 2 = 7 / 4 catch-up rows), and
    a tree whose levels do not fit a slab (top_k = 10: the tile-per-request form): a cohort request == the same request alone == the oracle."""
    sm, ot, od = build(50, 60, True, **tree)
    models = [sm] + [sm.make_cohort_member() for _ in range(n_req - 1)]
    rng = np.random.default_rng(93 + n_req)
    reqs = [(torch.from_numpy(rng.integers(3, IMG_TOK, size=n))[None], {}) for n in (17, 11, 23, 14)[:n_req]]
    budgets = [26, 19, 33, 12][:n_req]
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    got = specgenerate_cohort(models, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    od.cfg = dataclasses.replace(od.cfg, **tree)
    o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, reqs[0][0][0].numpy(), max_new_tokens=budgets[0], max_pos=T["max_pos"])
    np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
    assert got[0][3] == o_acc


@pytest.mark.parametrize("n_slots,temperature", [(4, 0.0), (3, 0.0), (2, 0.0), (4, 6.0)])
def test_request_stream_through_the_slots_of_a_cohort_equals_the_single_requests(golden_dir, monkeypatch, n_slots, temperature):
    """Continuous batching (specgenerate_stream): nine requests of ragged lengths and budgets — text and image prompts — through 2..4 request
    slots; a finished request's slot takes the next request while the others are mid-flight.  Every request returns what it returns alone
    (tokens, new_token, round count, accept lengths), greedy and sampling; fewer lockstep rounds than cohort-by-cohort execution."""
    from vispec_amd.model.spec_model_ours import specgenerate_stream
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    models = [sm] + [sm.make_cohort_member() for _ in range(n_slots - 1)]
    rng = np.random.default_rng(97)
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    reqs = [(torch.from_numpy(g["succ0_ids"])[None], {})]
    for i, n in enumerate((17, 9, 23, 12, 20, 7, 15, 11)):
        if i % 3 == 1:  # an image prompt every third request
            n_img = 11 + i
            ids = np.concatenate([rng.integers(3, IMG_TOK, 4), np.full(n_img, IMG_TOK), rng.integers(3, IMG_TOK, n)])
            feats = synth.bf16_grid(rng.standard_normal((n_img, T["D"]), dtype=np.float32) * 0.05)
            reqs.append((torch.from_numpy(ids)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda())))
        else:
            reqs.append((torch.from_numpy(rng.integers(3, IMG_TOK, size=n))[None], {}))
    budgets = [30, 12, 41, 8, 25, 33, 5, 19, 27]
    seeds = list(range(40, 49))
    gen = dict(temperature=temperature, top_k=8) if temperature > 0 else {}
    want = [single(sm, *r, max_new_tokens=b, seed=sd, **gen) for r, b, sd in zip(reqs, budgets, seeds)]
    st = {}
    g0 = sm.engine.graph_stats()
    side = torch.cuda.Stream()  # (the legacy default stream cannot be captured: graphs are replayed on side streams only)
    with torch.cuda.stream(side):
        got = specgenerate_stream(models, reqs, max_new_tokens=budgets, seeds=seeds, stats=st, **gen)
        side.synchronize()
    g1 = sm.engine.graph_stats()
    if temperature == 0:  # a refill does not re-capture: the requests' context hints fall in one 512-key bucket, the graph key holds their maximum
        assert g1["captures"] - g0["captures"] <= 2 and g1["replays"] - g0["replays"] >= 2 * st["rounds"] - 2
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3]), f"request {t}"
    per_request = [w[2] + 1 for w in want]
    assert st["request_rounds"] == sum(per_request)
    by_cohort = sum(max(per_request[lo:lo + n_slots]) for lo in range(0, len(per_request), n_slots))
    assert st["rounds"] <= by_cohort and st["rounds"] >= -(-sum(per_request) // n_slots)
    if temperature == 0:  # ... and request 0 is the oracle's / the reference fixture's stream
        o_out, o_new, o_idx, o_acc = vo.specgenerate(ot, od, g["succ0_ids"], max_new_tokens=30, max_pos=T["max_pos"])
        np.testing.assert_array_equal(got[0][0][0].cpu().numpy(), o_out)
        assert got[0][3] == o_acc
    # short queues fall back to the plain loops
    two = specgenerate_stream(models, reqs[:2], max_new_tokens=budgets[:2], seeds=seeds[:2], **gen)
    np.testing.assert_array_equal(two[1][0][0].cpu().numpy(), want[1][0][0].cpu().numpy())
    one = specgenerate_stream(models, reqs[2:3], max_new_tokens=budgets[2:3], seeds=seeds[2:3], **gen)
    np.testing.assert_array_equal(one[0][0][0].cpu().numpy(), want[2][0][0].cpu().numpy())
    if n_slots == 4:  # the loop without its round of lookahead (VISPEC_STREAM_LOOKAHEAD=0: launch, read the states, then launch) returns the same
        monkeypatch.setenv("VISPEC_STREAM_LOOKAHEAD", "0")
        with torch.cuda.stream(side):
            again = specgenerate_stream(models, reqs, max_new_tokens=budgets, seeds=seeds, **gen)
            side.synchronize()
        for t, (a, b) in enumerate(zip(again, got)):
            np.testing.assert_array_equal(a[0][0].cpu().numpy(), b[0][0].cpu().numpy(), err_msg=f"request {t}")
            assert a[1:] == b[1:], f"request {t}"


def test_cohort_of_four_on_a_side_stream_with_sampling_seeds():
    sm, ot, od = build(50, 60, True)
    models = [sm] + [sm.make_cohort_member() for _ in range(3)]
    rng = np.random.default_rng(92)
    reqs = [(torch.from_numpy(rng.integers(3, T["V"], size=n))[None], {}) for n in (14, 19, 9, 25)]
    seeds = [3, 5, 7, 11]
    want = [sm.specgenerate(r[0], temperature=6.0, top_k=8, seed=sd, max_new_tokens=20, log=True, return_acceptance_len=True) for r, sd in zip(reqs, seeds)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got = specgenerate_cohort(models, reqs, temperature=6.0, top_k=8, seeds=seeds, max_new_tokens=20)
        s.synchronize()
    for (toks, new_token, idx, acc), w in zip(got, want):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy())
        assert acc == w[3]
    assert sm.engine.graph_stats()["replays"] > 0


def test_member_slots_and_cohort_shape_errors():
    sm, _, _ = build(50, 60, True)
    other, _, _ = build(50, 60, True)
    m1, m2, m3, *_rest = [sm.make_cohort_member() for _ in range(7)]
    with pytest.raises(RuntimeError, match="seven members"):
        sm.make_cohort_member()
    with pytest.raises(RuntimeError, match="member of the first"):
        other.engine.cohort_round([m1.engine])
    with pytest.raises(RuntimeError, match="tiles 1 .. n-1"):
        sm.engine.cohort_round([m3.engine])  # a two-request round needs the member that owns tile 1
    with pytest.raises(RuntimeError, match="same member twice|tiles 1"):
        sm.engine.cohort_round([m1.engine, m1.engine])
    # trees of 33..64 nodes take two activation tiles per request (round 6): request slots 0..3 only, at most four requests per round
    with pytest.raises(RuntimeError, match="first four request slots"):
        _rest[0].engine.set_total_token(40)  # slot 4
    for m in (sm, m1, m2, m3, _rest[0]):
        m.engine.set_total_token(40 if m is not _rest[0] else 30)
    with pytest.raises(RuntimeError, match="at most four requests per round"):
        sm.engine.cohort_round([m1.engine, m2.engine, m3.engine, _rest[0].engine])
    m3.engine.set_total_token(30)
    with pytest.raises(RuntimeError, match="same tree size"):
        sm.engine.cohort_round([m1.engine, m2.engine, m3.engine])
    for m in (sm, m1, m2, m3):
        m.engine.set_total_token(30)


def test_step_api_keeps_stepping_after_done_outside_cohorts():
    """Round-2 advice: the freeze of a finished request belongs to cohort rounds only.  The single-request entry points (vispec_verify_accept,
    vispec_accept via utils.update_inference_inputs, vispec_ar_step) keep producing fresh results when a caller with its own stop rule goes
    on after the library's `done` flags are set (here: the token budget, bit 1)."""
    sm, _, _ = build(50, 60, True)
    rng = np.random.default_rng(93)
    ids = torch.from_numpy(rng.integers(3, T["V"], size=12))[None]
    eng = sm.engine
    sm._start_request(ids, None, {}, max_new_tokens=3)
    seen = []
    for _ in range(8):
        eng.verify_accept()
        eng.draft_round()
        st = eng.state()
        seen.append((st["n_ctx"], st["done"], st["rounds"]))
    assert seen[-1][1] & 2, "the budget flag must be set by now"
    assert all(b[0] > a[0] and b[2] == a[2] + 1 for a, b in zip(seen, seen[1:])), f"every step must advance the request, done or not: {seen}"


def test_closing_a_member_frees_its_tile_and_one_sync_reads_all_states():
    """Engine.close() destroys a member's context at once: its activation tile goes back to the leader (the next member gets the same
    tile), the leader's cohort graphs that bake the member in are dropped, and cohort rounds keep working with the new member.
    Engine.cohort_states() returns what state() returns for every request, with one synchronisation."""
    sm, _, _ = build(50, 60, True)
    m1 = sm.make_cohort_member()
    m2 = sm.make_cohort_member()
    rng = np.random.default_rng(94)
    reqs = [(torch.from_numpy(rng.integers(3, T["V"], size=n))[None], {}) for n in (13, 17, 11)]
    want = [single(sm, *r, max_new_tokens=18) for r in reqs]
    got = specgenerate_cohort([sm, m1, m2], reqs, max_new_tokens=18)
    for (toks, new_token, idx, acc), w in zip(got, want):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy())
    states = sm.engine.cohort_states([m1.engine, m2.engine])
    assert states == [sm.engine.state(), m1.engine.state(), m2.engine.state()]
    m1.engine.close()
    m1.engine.close()  # idempotent
    with pytest.raises(Exception):
        sm.engine.cohort_round([m1.engine, m2.engine])  # a closed context cannot take part any more
    m1b = sm.make_cohort_member()  # takes tile 1 again
    got = specgenerate_cohort([sm, m1b, m2], reqs, max_new_tokens=18)
    for (toks, new_token, idx, acc), w in zip(got, want):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy())


def test_cohort_at_long_context_equals_the_single_requests():
    """Contexts of 9 000 - 33 000 keys in 40 960-row caches (more than 64 x 512 rows: the attention's key range per workgroup doubles to
    1024; 9 - 33 key splits per request): the split boundaries come from the cache capacity, never from the requests of a launch, so a
    cohort request stays bit-identical to the same request alone at every context length — and the speculative stream is still the
    greedy stream of the same target."""
    from vispec_amd.engine import DraftConfig, TargetConfig
    from vispec_amd.model import SpecModel
    max_pos = 40960
    tw = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=50, structured=True)
    dw = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], num_q=2, seed=60, structured=True, target_embed=tw["model.embed_tokens.weight"], rho=0.25)
    tcfg = TargetConfig(hidden_size=T["D"], num_heads=T["H"], num_kv_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"],
                        num_layers=T["NL"], max_position_embeddings=max_pos, architectures=("LlamaForCausalLM",), image_token_index=IMG_TOK)
    dcfg = DraftConfig(hidden_size=T["D"], num_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"], max_position_embeddings=max_pos)
    sm = SpecModel.from_weights(tcfg, dcfg, tw, dw, num_q=2)
    models = [sm] + [sm.make_cohort_member() for _ in range(2)]
    rng = np.random.default_rng(97)
    reqs = [(torch.from_numpy(rng.integers(3, IMG_TOK, size=n))[None], {}) for n in (33000, 9000, 17000)]
    budgets = [24, 31, 18]
    want = [single(sm, *r, max_new_tokens=b) for r, b in zip(reqs, budgets)]
    got = specgenerate_cohort(models, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), w) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(toks[0].cpu().numpy(), w[0][0].cpu().numpy(), err_msg=f"request {t}")
        assert (new_token, idx, acc) == (w[1], w[2], w[3])
    ar = sm.baseline_generate(reqs[1][0], max_new_tokens=budgets[1])[0].cpu().numpy()
    out = want[1][0][0].cpu().numpy()
    n = min(len(ar), len(out))
    assert n > 9000 + 20
    np.testing.assert_array_equal(ar[:n], out[:n])


@pytest.mark.parametrize("n_req", [2, 3, 4])
def test_cohort_ar_baseline_equals_the_single_request_ar_runs(golden_dir, n_req):
    """vispec_cohortn_ar_step (bench.py's AR denominator at the cohort's batching): every request's greedy AR tokens are the tokens of
    SpecModel.baseline_generate for that request alone — and, greedy decoding being what speculative decoding preserves, the tokens of its
    speculative run (gen_baseline_answer_coco_caption.py:34-133 vs spec_model_ours.py:247-582).  Ragged budgets: requests freeze one by one."""
    from vispec_amd.model.spec_model_ours import baseline_generate_cohort
    sm, ot, od = build(50, 60, True, arch="LlavaNextForConditionalGeneration")
    members = [sm.make_cohort_member() for _ in range(n_req - 1)]
    rng = np.random.default_rng(191)
    g = np.load(os.path.join(golden_dir, "g8_loop.npz"))
    n_img = 23
    ids_img = np.concatenate([rng.integers(3, IMG_TOK, 5), np.full(n_img, IMG_TOK), rng.integers(3, IMG_TOK, 9)])
    feats = synth.bf16_grid(rng.standard_normal((n_img, T["D"]), dtype=np.float32) * 0.05)
    reqs = [(torch.from_numpy(g["succ0_ids"])[None], {}),
            (torch.from_numpy(ids_img)[None], dict(pixel_values=torch.from_numpy(feats).to(torch.bfloat16).cuda())),
            (torch.from_numpy(g["succ1_ids"])[None], {}),
            (torch.from_numpy(rng.integers(3, IMG_TOK, 21))[None], {})][:n_req]
    budgets = [30, 22, 41, 17][:n_req]
    want = [sm.baseline_generate(ids, max_new_tokens=b, max_steps=b + 1, **kw) for (ids, kw), b in zip(reqs, budgets)]
    st = {}
    got = baseline_generate_cohort([sm] + members, reqs, max_new_tokens=budgets, stats=st)
    assert st["steps"] >= max(budgets)
    for t, (a, w, b) in enumerate(zip(got, want, budgets)):
        np.testing.assert_array_equal(a[0].cpu().numpy(), w[0].cpu().numpy(), err_msg=f"request {t}")
        assert a.shape[1] == reqs[t][0].shape[1] + b + 1  # no EOS in these streams: budget + the token that crossed it
    spec = specgenerate_cohort([sm] + members, reqs, max_new_tokens=budgets)
    for t, ((toks, new_token, idx, acc), a) in enumerate(zip(spec, got)):
        n = min(toks.shape[1], a.shape[1])
        np.testing.assert_array_equal(toks[0, :n].cpu().numpy(), a[0, :n].cpu().numpy(), err_msg=f"request {t}: speculative != AR")
    # a second AR cohort on the same contexts replays the captured graph; the leader alone afterwards is still itself
    got2 = baseline_generate_cohort([sm] + members, reqs, max_new_tokens=budgets)
    for a, b in zip(got, got2):
        assert torch.equal(a, b)
    again = sm.baseline_generate(reqs[0][0], max_new_tokens=budgets[0], max_steps=budgets[0] + 1)
    assert torch.equal(again, want[0])


def test_destroying_a_leader_keeps_its_workspaces_until_the_last_member_is_gone():
    """Round-3 advice: a member's GEMM workspaces are rows of its leader's.  Closing the leader first must not leave the members with dangling
    pointers: the leader's allocations live on until its last member is closed, the members keep working as single requests, and a cohort
    round on the destroyed leader is refused."""
    sm, _, _ = build(50, 60, True)
    m1, m2 = sm.make_cohort_member(), sm.make_cohort_member()
    rng = np.random.default_rng(17)
    ids = torch.from_numpy(rng.integers(3, T["V"], size=14))[None]
    want = m1.specgenerate(ids, max_new_tokens=12)
    lead_engine = sm.engine
    lead_engine.close()
    got = m1.specgenerate(ids, max_new_tokens=12)  # runs on rows of the (kept) leader workspaces
    assert torch.equal(got, want)
    got2 = m2.specgenerate(ids, max_new_tokens=12)
    assert torch.equal(got2, want)
    m1.engine.close()
    assert torch.equal(m2.specgenerate(ids, max_new_tokens=12), want)
    m2.engine.close()  # the last member frees the leader's allocations too
