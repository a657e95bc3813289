"""GPU parity of each HIP kernel (through the C-ABI) against the numpy oracle in bf16-emulation mode.

Tolerances (written here, used below):
  * bf16 results whose fp32 accumulation order differs from numpy's: equal up to ONE bf16 ulp
    (|a-b| <= 2^-7 * max(|a|,|b|) + 1e-6) on every element, and >= 97 % of the elements bit-identical;
  * integer outputs (argmax / top-k indices, tree tables): bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from helpers import T, vo  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402


def dev():
    return torch.device("cuda:0")


def tb(x):
    """numpy fp32 (bf16-representable) -> bf16 device tensor"""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(dev()).contiguous()


def fn(t):
    return t.float().cpu().numpy()


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def assert_bf16_close(got, want, min_exact=0.97, ulps=1, scale=None, outlier_frac=2e-5, outlier_mult=2, atol=1e-6):
    """`scale`: magnitude the ulp is taken at (default: the values themselves).  For a chained epilogue
    (linear -> round -> + residual -> round) a 1-ulp flip of the FIRST rounding survives at the magnitude of the
    operands, not of the (possibly cancelling) sum, so the operands' magnitude is passed as scale."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    mag = np.maximum(np.abs(got), np.abs(want))
    if scale is not None:
        mag = np.maximum(mag, np.abs(scale))
    tol = ulps * 2.0 ** -7 * mag + atol
    bad = np.abs(got - want) > tol
    # statistical tail of large shapes: a handful of elements per million sit where two roundings flip together (e.g. a
    # result next to a binade boundary); allow <= 2e-5 of the elements up to twice the bound, nothing beyond that
    if bad.mean() <= outlier_frac and not (np.abs(got - want) > outlier_mult * tol).any():
        bad[:] = False
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(got - want) / tol), got.shape)
        raise AssertionError(f"{bad.sum()} / {bad.size} beyond {ulps} bf16 ulp; worst abs {np.abs(got - want).max()}; "
                             f"worst rel-to-tol {(np.abs(got - want) / tol).max():.2f} at {i}: got {got[i]} want {want[i]}; "
                             f"exact frac {(got == want).mean():.4f}")
    exact = (got == want).mean()
    assert exact >= min_exact, f"only {exact:.4f} bit-identical"


@pytest.fixture(scope="module")
def lib():
    return L.load(build_if_missing=True)


@pytest.fixture(scope="module")
def engine():
    from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights
    tcfg = TargetConfig(hidden_size=T["D"], num_heads=T["H"], num_kv_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"],
                        num_layers=T["NL"], max_position_embeddings=T["max_pos"])
    dcfg = DraftConfig(hidden_size=T["D"], num_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"],
                       max_position_embeddings=T["max_pos"])
    tw = TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=0), dev())
    dw = DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], seed=1), 2, dev())
    return Engine(tcfg, dcfg, tw, dw)


def packed(w, swiglu=False):
    from vispec_amd.engine import pack_weight, swiglu_order
    return pack_weight(swiglu_order(tb(w)) if swiglu else tb(w))


def test_pack_weight_layout(lib):
    """W32: tile (32 rows x 16 k) = lane l holds W[32t + (l&31)][16c + 8(l>>5) ...+8]; rows padded with zeros."""
    rng = np.random.default_rng(1)
    N, K = 72, 48
    w = synth.bf16_grid(rng.standard_normal((N, K), dtype=np.float32))
    P = fn(packed(w)).reshape(3, K // 16, 64, 8)
    for t in range(3):
        for c in range(K // 16):
            for l in (0, 5, 31, 32, 40, 63):
                row, k = 32 * t + (l & 31), 16 * c + 8 * (l >> 5)
                want = w[row, k : k + 8] if row < N else np.zeros(8, np.float32)
                np.testing.assert_array_equal(P[t, c, l], want)


SKINNY_SHAPES = [(1, 256, 256), (5, 64, 704), (8, 1008, 256), (16, 256, 8192), (30, 768, 256),
                 (30, 256, 704), (32, 4096, 4096), (30, 12288, 4096), (7, 96, 11008), (30, 32064, 512),
                 (33, 256, 704), (64, 4096, 4096), (60, 8192, 256), (47, 1008, 8192), (40, 96, 11008)]  # > 32 rows: two tiles


def epi_cases(shapes, epis=(0, 1, 2)):
    """(shape..., epi) products without the combinations that do not exist (SwiGLU packs 16 gate + 16 up rows per tile: N % 16 == 0)."""
    return [tuple(sh) + (e,) for sh in shapes for e in epis if not (e == 2 and sh[-2] % 16)]


@pytest.mark.parametrize("M,N,K,epi", epi_cases(SKINNY_SHAPES))
@pytest.mark.parametrize("bias", [False, True])
def test_gemm_skinny(lib, engine, M, N, K, epi, bias):
    rng = np.random.default_rng(M * 131 + N * 7 + K + epi)
    o = vo.Ops(bf16=True)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    rows = 2 * N if epi == 2 else N
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32)) if bias else None
    r = synth.bf16_grid(rng.standard_normal((M, N), dtype=np.float32))
    scale = None
    if epi == 0:
        want = o.linear(x, w, b)
    elif epi == 1:
        lin = o.linear(x, w, b)
        want = o.add(r, lin)
        scale = np.maximum(np.abs(r), np.abs(lin))
    else:
        gu = o.linear(x, w, b)
        want = o.silu_mul(gu[:, :N], gu[:, N:])
    X, W, B, R = tb(x), packed(w, swiglu=(epi == 2)), (tb(b) if bias else None), tb(r)
    Y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny(engine.h, stream(), p(X), K, p(W), p(B), p(Y), N, p(R), N, M, N, K, epi))
    torch.cuda.synchronize()
    # one extra ulp for the SwiGLU epilogue: it chains three rounded ops, a flipped gate rounding propagates (a near-zero gate
    # carries the rounding of much larger terms: the handful of tail elements may sit at up to 3x the bound)
    # ... and a SwiGLU output that is itself ~0 (|gate| ~ 1e-5 after cancelling O(1) products over K terms) has no relative
    # accuracy to speak of in ANY summation order: absolute floor of 1e-4 against outputs of O(1)
    assert_bf16_close(fn(Y), want, min_exact=0.90 if epi == 2 else 0.97, ulps=2 if epi == 2 else 1, scale=scale,
                      outlier_mult=3 if epi == 2 else 2, atol=1e-4 if epi == 2 else 1e-6, outlier_frac=5e-5 if epi == 2 else 2e-5)


def test_gemm_strided_output_and_padding_rows_untouched(lib, engine):
    rng = np.random.default_rng(3)
    M, N, K = 7, 64, 256
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    X, W = tb(x), packed(w)
    Y = torch.full((16, 2 * N), 3.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny(engine.h, stream(), p(X), K, p(W), None, C.c_void_p(Y.data_ptr() + 2 * N), 2 * N, None, 0, M, N, K, 0))
    torch.cuda.synchronize()
    y = fn(Y)
    assert_bf16_close(y[:M, N:], vo.Ops(True).linear(x, w))
    assert (y[:, :N] == 3.0).all() and (y[M:] == 3.0).all()


@pytest.mark.parametrize("M,N,K,res,bias", [(30, 4096, 4096, True, False), (8, 256, 704, True, True), (1, 4096, 11008, False, True),
                                            (60, 4096, 4096, True, True), (35, 256, 704, True, False), (64, 3584, 18944, True, False)])
def test_gemm_with_fused_rmsnorm(lib, engine, M, N, K, res, bias):
    """o_proj/down_proj + residual + the RMSNorm that follows, in one split-K GEMM + one reduce."""
    rng = np.random.default_rng(N + K + M)
    o = vo.Ops(bf16=True)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(N, dtype=np.float32)) if bias else None
    r = synth.bf16_grid(rng.standard_normal((M, N), dtype=np.float32) * 3)
    nw = synth.bf16_grid(1 + 0.1 * rng.standard_normal(N, dtype=np.float32))
    lin = o.linear(x, w, b)
    h = o.add(r, lin) if res else lin
    want_n = o.rmsnorm(h, nw, 1e-5)
    X, W, B, R, NW = tb(x), packed(w), (tb(b) if bias else None), tb(r), tb(nw)
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    Yn = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny_norm(engine.h, stream(), p(X), K, p(W), p(B), p(Y), N, p(R) if res else None, N, p(NW), p(Yn), N, 1e-5,
                                        M, N, K))
    torch.cuda.synchronize()
    assert_bf16_close(fn(Y), h, scale=np.maximum(np.abs(r), np.abs(lin)) if res else None)
    # the norm sees an h that may differ by an ulp in a few places: 2 ulp on the normed output
    # (where r + lin cancels, a 1-ulp flip of lin at operand scale is many ulps of the small h: rare elements up to 4x the bound)
    assert_bf16_close(fn(Yn), want_n, min_exact=0.9, ulps=2, outlier_mult=4)


@pytest.mark.parametrize("M,D", [(1, 256), (30, 4096), (8, 3584)])
def test_rmsnorm(lib, M, D):
    rng = np.random.default_rng(D + M)
    x = synth.bf16_grid(rng.standard_normal((M, D), dtype=np.float32) * 3)
    w = synth.bf16_grid(1 + 0.1 * rng.standard_normal(D, dtype=np.float32))
    X, W = tb(x), tb(w)
    Y = torch.empty_like(X)
    L.check(lib.vispec_rmsnorm(None, stream(), p(X), p(W), p(Y), M, D, 1e-5))
    torch.cuda.synchronize()
    assert_bf16_close(fn(Y), vo.Ops(True).rmsnorm(x, w, 1e-5))


@pytest.mark.parametrize("H,Hkv", [(2, 2), (4, 2)])
def test_rope_append(lib, H, Hkv):
    rng = np.random.default_rng(11)
    M, hd, S = 9, 128, 96
    o = vo.Ops(True)
    cos, sin = vo.rope_tables(hd, 256, 10000.0)
    cos, sin = synth.bf16_grid(cos), synth.bf16_grid(sin)
    qkv = synth.bf16_grid(rng.standard_normal((M, (H + 2 * Hkv) * hd), dtype=np.float32))
    pos_off = rng.integers(0, 20, size=M).astype(np.int32)
    base, kvb = 40, 17
    QKV = tb(qkv)
    kc = torch.zeros(Hkv, S, hd, dtype=torch.bfloat16, device=dev())
    vc = torch.zeros_like(kc)
    d_base = torch.tensor([base], dtype=torch.int32, device=dev())
    d_kvb = torch.tensor([kvb], dtype=torch.int32, device=dev())
    d_off = torch.from_numpy(pos_off).to(dev())
    COS, SIN = tb(cos), tb(sin)  # keep alive: a temporary would be freed (and its block reused) before the launch
    L.check(lib.vispec_rope_append(None, stream(), p(QKV), M, H, Hkv, hd, p(COS), p(SIN), p(d_base), p(d_off), p(kc), p(vc), S, p(d_kvb)))
    torch.cuda.synchronize()
    pos = base + pos_off
    q = qkv[:, : H * hd].reshape(M, H, hd).transpose(1, 0, 2)
    k = qkv[:, H * hd : (H + Hkv) * hd].reshape(M, Hkv, hd).transpose(1, 0, 2)
    v = qkv[:, (H + Hkv) * hd :].reshape(M, Hkv, hd).transpose(1, 0, 2)
    np.testing.assert_array_equal(fn(QKV)[:, : H * hd].reshape(M, H, hd).transpose(1, 0, 2), o.rope(q, cos, sin, pos))
    np.testing.assert_array_equal(fn(kc)[:, kvb : kvb + M], o.rope(k, cos, sin, pos))
    np.testing.assert_array_equal(fn(vc)[:, kvb : kvb + M], v)
    assert (fn(kc)[:, :kvb] == 0).all() and (fn(kc)[:, kvb + M :] == 0).all()


def _tree_mask(rng, M, tail):
    m = np.zeros((M, tail), bool)
    for i in range(M):
        m[i, rng.integers(0, tail)] = True  # at least one visible tail key per row
        m[i] |= rng.random(tail) < 0.3
    return m


@pytest.mark.parametrize("H,Hkv,M,prefix,tail,eager", [
    (2, 2, 30, 300, 30, 1), (2, 2, 30, 0, 30, 1), (2, 2, 1, 777, 1, 1), (2, 2, 8, 131, 24, 0), (2, 2, 5, 1, 5, 0),
    (2, 2, 2, 600, 0, 0), (4, 2, 30, 257, 30, 1), (14, 2, 3, 90, 3, 1), (2, 2, 64, 500, 64, 1), (2, 2, 30, 1500, 30, 1),
    (2, 1, 30, 9000, 30, 0), (2, 2, 30, 33000, 30, 1)])  # the last two: 18 key splits; a 40 960-row cache (key range per workgroup doubled to 1024)
def test_tree_attention(lib, engine, H, Hkv, M, prefix, tail, eager):
    rng = np.random.default_rng(H * 1000 + M * 10 + prefix + tail)
    hd, S = 128, 2048 if prefix + tail <= 2048 else (10240 if prefix + tail <= 10240 else 40960)
    o = vo.Ops(True)
    q = synth.bf16_grid(rng.standard_normal((M, H, hd), dtype=np.float32))
    k = synth.bf16_grid(rng.standard_normal((Hkv, S, hd), dtype=np.float32))
    v = synth.bf16_grid(rng.standard_normal((Hkv, S, hd), dtype=np.float32))
    mask = _tree_mask(rng, M, tail) if tail else np.zeros((M, 0), bool)
    bits = np.zeros(M, np.uint64)
    for i in range(M):
        for t in range(tail):
            if mask[i, t]:
                bits[i] |= np.uint64(1) << np.uint64(t)
    allow = np.concatenate([np.ones((M, prefix), bool), mask], axis=1)
    n = prefix + tail
    rep = H // Hkv
    kk, vv = np.repeat(k[:, :n], rep, axis=0), np.repeat(v[:, :n], rep, axis=0)
    qh = q.transpose(1, 0, 2)
    want = (o.attn_eager if eager else o.attn_sdpa)(qh, kk, vv, allow).transpose(1, 0, 2).reshape(M, H * hd)
    # magnitude of the summands sum_k p_k |v_k|: rounding P (and, in eager mode, the scores) to bf16 perturbs each term by
    # 2^-9 relative, so the error lives at THIS scale, not at the scale of the (cancelling) sum
    sc = np.einsum("hqd,hkd->hqk", qh, kk) / np.sqrt(hd)
    sc = np.where(allow[None], sc, -np.inf)
    pr = np.exp(sc - sc.max(-1, keepdims=True))
    pr /= pr.sum(-1, keepdims=True)
    scale = np.einsum("hqk,hkd->hqd", pr, np.abs(vv)).transpose(1, 0, 2).reshape(M, H * hd)
    Q, Kc, Vc = tb(q.reshape(M, H * hd)), tb(k), tb(v)
    d_prefix = torch.tensor([prefix], dtype=torch.int32, device=dev())
    d_bits = torch.from_numpy(bits.view(np.int64)).to(dev())
    O = torch.zeros(M, H * hd, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_tree_attention(engine.h, stream(), p(Q), H * hd, p(Kc), p(Vc), S, H, Hkv, hd, M, p(d_prefix), tail,
                                      p(d_bits) if tail else None, p(O), H * hd, eager))
    torch.cuda.synchronize()
    # P is rounded to bf16 against a different (chunk-local) max than the oracle's global one, and eager mode also
    # rounds the scores: 2 ulp at the summand scale; about half of the outputs are bit-identical
    assert_bf16_close(fn(O), want, min_exact=0.40, ulps=2, scale=scale)


@pytest.mark.parametrize("H,Hkv,Lq,eager", [(2, 2, 300, 1), (2, 2, 129, 1), (4, 2, 257, 0), (2, 2, 1, 1), (14, 2, 95, 0), (2, 1, 640, 1), (1, 1, 128, 0),
                                            (32, 32, 2100, 1), (28, 4, 2590, 0)])  # the last two: paired row blocks (long + short per workgroup)
def test_prefill_attention_against_the_oracle(lib, engine, H, Hkv, Lq, eager):
    """vispec_prefill_attention (causal attention of a prompt's rows over the K/V rows its prefill wrote — the reference's eager
    LlamaAttention / SDPA Qwen attention at prefill time, modeling_llama_kv.py:595-640) against the oracle's attention with a causal
    mask: same tolerance as the decode kernel whose tile arithmetic it shares.  Row counts around the 128-row workgroup and 32-row wave
    boundaries, one row, GQA."""
    rng = np.random.default_rng(H * 1000 + Lq * 7 + eager)
    hd, S = 128, 1024 if Lq <= 1024 else 2600
    o = vo.Ops(True)
    q = synth.bf16_grid(rng.standard_normal((Lq, H, hd), dtype=np.float32))
    k = synth.bf16_grid(rng.standard_normal((Hkv, S, hd), dtype=np.float32))
    v = synth.bf16_grid(rng.standard_normal((Hkv, S, hd), dtype=np.float32))
    allow = np.tril(np.ones((Lq, Lq), bool))
    rep = H // Hkv
    kk, vv = np.repeat(k[:, :Lq], rep, axis=0), np.repeat(v[:, :Lq], rep, axis=0)
    qh = q.transpose(1, 0, 2)
    want = (o.attn_eager if eager else o.attn_sdpa)(qh, kk, vv, allow).transpose(1, 0, 2).reshape(Lq, H * hd)
    sc = np.einsum("hqd,hkd->hqk", qh, kk) / np.sqrt(hd)
    sc = np.where(allow[None], sc, -np.inf)
    pr = np.exp(sc - sc.max(-1, keepdims=True))
    pr /= pr.sum(-1, keepdims=True)
    scale = np.einsum("hqk,hkd->hqd", pr, np.abs(vv)).transpose(1, 0, 2).reshape(Lq, H * hd)
    ldq = (H + 2 * Hkv) * hd  # the fused q|k|v row of the prefill: q heads first
    Qfull = torch.zeros(Lq, ldq, dtype=torch.bfloat16, device=dev())
    Qfull[:, : H * hd] = tb(q.reshape(Lq, H * hd))
    Kc, Vc = tb(k), tb(v)
    O = torch.full((Lq + 3, H * hd), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_prefill_attention(engine.h, stream(), p(Qfull), ldq, p(Kc), p(Vc), S, H, Hkv, Lq, p(O), H * hd, eager))
    torch.cuda.synchronize()
    assert (O[Lq:].float() == 7.0).all(), "rows past L must stay untouched"
    assert_bf16_close(fn(O[:Lq]), want, min_exact=0.40, ulps=2, scale=scale)


def test_argmax_rows_first_max_wins(lib):
    rng = np.random.default_rng(5)
    M, V = 30, 32064
    x = synth.bf16_grid(rng.standard_normal((M, V), dtype=np.float32) * 4)
    x[3, 100] = x[3, 20000] = 99.0  # tie -> lowest index
    x[4, V - 1] = 120.0
    x[5, 0] = 120.0
    X = tb(x)
    out = torch.zeros(M, dtype=torch.int32, device=dev())
    L.check(lib.vispec_argmax_rows(None, stream(), p(X), V, M, V, p(out)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), np.argmax(x, axis=1))
    assert out[3].item() == 100 and out[4].item() == V - 1 and out[5].item() == 0


@pytest.mark.parametrize("M,V,k", [(1, 32064, 8), (8, 1008, 8), (8, 152064, 8), (3, 1008, 16),
                                   (3, 1003, 8), (3, 200000, 8)])  # the last two take the multi-pass form (V % 8 != 0; V > 163 840)
def test_logsoftmax_topk(lib, engine, M, V, k):
    rng = np.random.default_rng(V + M)
    o = vo.Ops(True)
    x = synth.bf16_grid(rng.standard_normal((M, V), dtype=np.float32) * 3)
    x[0, 7] = x[0, 900] = x[0, 901] = 30.0  # ties inside the top-k
    if M >= 3:  # massive ties: a constant row, and a row whose k-th value is shared by many entries spread over all chunks
        x[1, :] = 0.5
        x[2, :] = -1.0
        x[2, ::97] = 2.0
        x[2, 500] = 3.0
    X = tb(x)
    idx = torch.zeros(M, k, dtype=torch.int32, device=dev())
    lp = torch.zeros(M, k, dtype=torch.float32, device=dev())
    L.check(lib.vispec_logsoftmax_topk(engine.h, stream(), p(X), V, M, V, k, p(idx), p(lp)))
    torch.cuda.synchronize()
    want = o.log_softmax(x)
    got_lp, got_idx = lp.cpu().numpy(), idx.cpu().numpy()
    for r in range(M):
        wv, wi = vo.topk_desc(want[r], k)
        # the kernel's fp32 log-sum-exp uses fast exp/log: values may sit one bf16 ulp away, which can permute
        # neighbours that the oracle separates by exactly that ulp; require set-equality on a 1-ulp-tolerant basis
        assert_bf16_close(got_lp[r], wv, min_exact=0.0, ulps=1)
        recomputed = want[r][got_idx[r]]
        assert_bf16_close(recomputed, got_lp[r], min_exact=0.0, ulps=1)
        assert len(set(got_idx[r].tolist())) == k
        assert (np.diff(got_lp[r]) <= 0).all()
    assert got_idx[0, :3].tolist() == [7, 900, 901]
    if M >= 3:  # ties resolve to the lowest indices, in index order
        assert got_idx[1].tolist() == list(range(k))
        mult = [i for i in range(0, V, 97)]
        assert got_idx[2].tolist() == ([500] + mult + [i for i in range(V) if i % 97 and i != 500])[:k]


@pytest.mark.parametrize("M,N,K", [(30, 5120, 5120), (30, 2 * 13824, 5120), (8, 5120, 13824), (30, 3584 + 2 * 512, 3584), (8, 18944 * 2, 3584)])
def test_gemm_other_model_shapes(lib, engine, M, N, K):
    """LLaVA-1.6-13B (D=5120, I=13824) and Qwen2.5-VL-7B (D=3584, GQA qkv, I=18944) GEMM shapes."""
    rng = np.random.default_rng(N + K)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((N, K), dtype=np.float32) * 0.03)
    X, W = tb(x), packed(w)
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny(engine.h, stream(), p(X), K, p(W), None, p(Y), N, None, 0, M, N, K, 0))
    torch.cuda.synchronize()
    assert_bf16_close(fn(Y), vo.Ops(True).linear(x, w))


@pytest.mark.parametrize("M,N,K,epi", epi_cases([(1, 64, 64), (30, 256, 704), (8, 1008, 256), (30, 4096, 3584), (30, 1024, 18944), (5, 96, 11008),
                                                 (60, 4096, 3584), (37, 1024, 704)]))
def test_gemm_fp8_weights(lib, engine, M, N, K, epi):
    """W8A16: e4m3 weights (per-output-channel scale), bf16 activations; Y = bf16(scale * (X · q^T) + b) (+ epilogue)."""
    from vispec_amd.engine import pack_weight_fp8, quantize_fp8
    rng = np.random.default_rng(N * 3 + K + M + epi)
    o = vo.Ops(bf16=True)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    rows = 2 * N if epi == 2 else N
    w = synth.bf16_grid(rng.standard_normal((rows, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(rows, dtype=np.float32))
    r = synth.bf16_grid(rng.standard_normal((M, N), dtype=np.float32))
    q_u8, sc = quantize_fp8(tb(w))
    # the product quantiser (torch float8 on the GPU) and the oracle's numpy e4m3 quantiser agree except on exact .5 ties of
    # w/scale (bf16 weights make such ties real; the last ulp of the fp32 division decides them)
    qo, so = vo.quantize_fp8(w)
    qp = q_u8.view(torch.float8_e4m3fn).float().cpu().numpy()
    assert (qp != qo).mean() < 2e-2 and np.abs(qp - qo).max() <= np.abs(qo).max() / 8
    np.testing.assert_allclose(sc.cpu().numpy(), so, rtol=2e-7)
    qo, so = qp, sc.cpu().numpy()  # the GEMM itself is checked on the product's own codes
    from vispec_amd.engine import swiglu_order
    P8 = pack_weight_fp8(swiglu_order(q_u8) if epi == 2 else q_u8)
    scale = None
    if epi == 0:
        want = o.linear(x, (qo, so), b)
    elif epi == 1:
        lin = o.linear(x, (qo, so), b)
        want = o.add(r, lin)
        scale = np.maximum(np.abs(r), np.abs(lin))
    else:
        gu = o.linear(x, (qo, so), b)
        want = o.silu_mul(gu[:, :N], gu[:, N:])
    X, B, R = tb(x), tb(b), tb(r)
    Y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_gemm_skinny_fp8(engine.h, stream(), p(X), K, p(P8), p(sc), p(B), p(Y), N, p(R), N, M, N, K, epi))
    torch.cuda.synchronize()
    # SwiGLU on fp8 codes (|q| up to 448): a gate that cancels to ~0 carries the fp32 accumulation-order error of much larger
    # terms, so a few elements per 10^4 sit further out (bounded at 8x)
    assert_bf16_close(fn(Y), want, min_exact=0.90 if epi == 2 else 0.97, ulps=2 if epi == 2 else 1, scale=scale,
                      outlier_frac=1e-3 if epi == 2 else 2e-5, outlier_mult=8 if epi == 2 else 2)


@pytest.mark.parametrize("H,Hkv,K,M,fp8,bias", [(48, 8, 256, 30, False, True), (56, 4, 128, 7, False, False), (48, 8, 256, 30, True, True),
                                                (2, 2, 256, 9, False, True), (28, 4, 3584, 30, False, True), (28, 4, 3584, 30, True, True),
                                                (48, 8, 256, 60, False, True), (28, 4, 3584, 41, True, True), (2, 2, 256, 50, False, True)])
def test_gemm_qkv_rope_fused(lib, engine, H, Hkv, K, M, fp8, bias):
    """One-launch q|k|v projection + rotary + KV append (EPI_ROPE, weight packed in rope order) == the GEMM followed by
    vispec_rope_append, bit for bit, and == the oracle's linear + rope.  (2,2) is below the fused threshold: same entry
    point, two-kernel path, natural weight order."""
    from vispec_amd.engine import pack_weight, pack_weight_fp8, quantize_fp8, qkv_rope_order
    rng = np.random.default_rng(H * 7 + K + M)
    hd, S = 128, 96
    N = (H + 2 * Hkv) * hd
    assert bool(lib.vispec_qkv_rope_fused(N)) == (N // 32 >= 128)
    o = vo.Ops(True)
    cos, sin = vo.rope_tables(hd, 128, 10000.0)
    cos, sin = synth.bf16_grid(cos), synth.bf16_grid(sin)
    x = synth.bf16_grid(rng.standard_normal((M, K), dtype=np.float32))
    w = synth.bf16_grid(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    b = synth.bf16_grid(rng.standard_normal(N, dtype=np.float32)) if bias else None
    pos_off = rng.integers(0, 20, size=M).astype(np.int32)
    base, kvb = 40, 11
    X, W, B = tb(x), tb(w), (tb(b) if bias else None)
    COS, SIN = tb(cos), tb(sin)
    d_base = torch.tensor([base], dtype=torch.int32, device=dev())
    d_kvb = torch.tensor([kvb], dtype=torch.int32, device=dev())
    d_off = torch.from_numpy(pos_off).to(dev())
    if fp8:
        q_u8, sc = quantize_fp8(W)
        P_nat, P_rope = pack_weight_fp8(q_u8), pack_weight_fp8(qkv_rope_order(q_u8, H + Hkv))
        wo = (q_u8.view(torch.float8_e4m3fn).float().cpu().numpy(), sc.cpu().numpy())
    else:
        sc = None
        P_nat, P_rope = pack_weight(W), pack_weight(qkv_rope_order(W, H + Hkv))
        wo = w
    # reference composition: GEMM (natural order) then the rotary/append kernel
    Y0 = torch.zeros((M, N), dtype=torch.bfloat16, device=dev())
    kc0 = torch.zeros(Hkv, S, hd, dtype=torch.bfloat16, device=dev())
    vc0 = torch.zeros_like(kc0)
    if fp8:
        L.check(lib.vispec_gemm_skinny_fp8(engine.h, stream(), p(X), K, p(P_nat), p(sc), p(B), p(Y0), N, None, 0, M, N, K, 0))
    else:
        L.check(lib.vispec_gemm_skinny(engine.h, stream(), p(X), K, p(P_nat), p(B), p(Y0), N, None, 0, M, N, K, 0))
    L.check(lib.vispec_rope_append(None, stream(), p(Y0), M, H, Hkv, hd, p(COS), p(SIN), p(d_base), p(d_off), p(kc0), p(vc0), S, p(d_kvb)))
    # fused entry point
    Y1 = torch.zeros((M, N), dtype=torch.bfloat16, device=dev())
    kc1 = torch.zeros_like(kc0)
    vc1 = torch.zeros_like(kc0)
    L.check(lib.vispec_gemm_qkv_rope(engine.h, stream(), p(X), K, p(P_rope), p(sc), p(B), p(Y1), M, H, Hkv, hd, K, p(COS), p(SIN),
                                     p(d_base), p(d_off), p(kc1), p(vc1), S, p(d_kvb)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(fn(Y1)[:, : H * hd], fn(Y0)[:, : H * hd])
    np.testing.assert_array_equal(fn(kc1), fn(kc0))
    np.testing.assert_array_equal(fn(vc1), fn(vc0))
    # and against the oracle
    qkv = o.linear(x, wo, b)
    pos = base + pos_off
    qo = o.rope(qkv[:, : H * hd].reshape(M, H, hd).transpose(1, 0, 2), cos, sin, pos)
    ko = o.rope(qkv[:, H * hd : (H + Hkv) * hd].reshape(M, Hkv, hd).transpose(1, 0, 2), cos, sin, pos)
    vo_ = qkv[:, (H + Hkv) * hd :].reshape(M, Hkv, hd).transpose(1, 0, 2)
    # x1*cos - x2*sin cancels now and then: the ulp that matters is that of the operands (|x_d|, |x_{d+-64}|), passed as scale
    q_in = qkv[:, : H * hd].reshape(M, H, hd).transpose(1, 0, 2)
    k_in = qkv[:, H * hd : (H + Hkv) * hd].reshape(M, Hkv, hd).transpose(1, 0, 2)
    pair = lambda x: np.maximum(np.abs(x), np.abs(np.roll(x, 64, axis=-1)))
    assert_bf16_close(fn(Y1)[:, : H * hd].reshape(M, H, hd).transpose(1, 0, 2), qo, ulps=2, min_exact=0.9, scale=pair(q_in))
    assert_bf16_close(fn(kc1)[:, kvb : kvb + M], ko, ulps=2, min_exact=0.9, scale=pair(k_in))
    assert_bf16_close(fn(vc1)[:, kvb : kvb + M], vo_)
    assert (fn(kc1)[:, :kvb] == 0).all() and (fn(kc1)[:, kvb + M :] == 0).all()


@pytest.mark.parametrize("M,N,bias", [(1, 152064, False), (37, 4608, True), (300, 1008, True), (5, 8, False)])
def test_scale_bias_cast_equals_the_torch_op_sequence_bit_for_bit(lib, M, N, bias):
    """Epilogue of the fp8-weight prefill GEMMs: bf16(acc * scale + bias) in one pass == torch's mul / add / cast sequence (the arithmetic the
    round-2 fp8 prefill used, pinned by test_fp8_prefill_computes_with_codes_and_scales_like_the_decode_gemms), exactly."""
    g = torch.Generator(device="cpu").manual_seed(M + N)
    acc = (torch.randn(M, N, generator=g) * 37.0).to(dev())
    sc = (torch.rand(N, generator=g) * 0.01 + 1e-4).to(dev())
    b = (torch.randn(N, generator=g) * 0.5).to(torch.bfloat16).to(dev()) if bias else None
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    L.check(lib.vispec_scale_bias_cast(None, stream(), p(acc), N, p(sc), p(b), p(out), N, M, N))
    want = acc * sc
    if bias:
        want = want + b.float()
    torch.cuda.synchronize()
    assert torch.equal(out, want.to(torch.bfloat16))
