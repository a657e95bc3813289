"""The cohort-8 GEMM (csrc/gemm_c8.h: eight requests per weight pass, one accumulator chain per element) shape by shape at M = 240, kernel
alone, weights rotated through > 1 GB, next to the eight-row-block kernel at M = 120 (four requests) and the single-request kernel on the
same weights; first a correctness pass of vispec_gemm_cohort(n_req = 5..8) against an fp64 product.
    python tools/c8_bench.py            (SHAPES=llava7b|qwen7b|llava13b)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights, pack_weight  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
T = synth.TINY
tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"])
dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
eng = Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev),
             DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev))
SHAPE_SETS = {"llava7b": [("qkv", 12288, 4096, 1), ("o_proj", 4096, 4096, 4), ("gate_up", 22016, 4096, 1), ("down", 4096, 11008, 4), ("lm_head", 32064, 4096, 1)],
              "llava13b": [("qkv", 15360, 5120, 1), ("o_proj", 5120, 5120, 4), ("gate_up", 27648, 5120, 1), ("down", 5120, 13824, 4), ("lm_head", 32064, 5120, 1)],
              "qwen7b": [("qkv", 4608, 3584, 1), ("o_proj", 3584, 3584, 4), ("gate_up", 37888, 3584, 1), ("down", 3584, 18944, 4), ("lm_head_65536_of_152064", 65536, 3584, 1)]}
SHAPES = SHAPE_SETS[os.environ.get("SHAPES", "llava7b")]
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

# ---- correctness: every epilogue, n_req 5..8, a split-K shape and an un-split one, against the fp64 product rounded like the epilogues do
for N, K in [(4096, 4096), (512, 11008), (1008, 256), (256, 64)]:
    for n_req in (5, 8):
        for epi in (0, 1, 2):
            if epi == 2 and N % 16:
                continue
            g = torch.Generator(device="cpu").manual_seed(N + K + n_req + epi)
            rows = 2 * N if epi == 2 else N
            X = (torch.randn(32 * n_req, K, generator=g)).to(torch.bfloat16).to(dev)
            Wn = (torch.randn(rows, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
            B = torch.randn(rows, generator=g).to(torch.bfloat16).to(dev)
            R = torch.randn(32 * n_req, N, generator=g).to(torch.bfloat16).to(dev)
            if epi == 2:
                from vispec_amd.engine import swiglu_order
                W = pack_weight(swiglu_order(Wn))
            else:
                W = pack_weight(Wn)
            Y = torch.full((32 * n_req, N), 7.0, dtype=torch.bfloat16, device=dev)
            L.check(lib.vispec_gemm_cohort(eng.h, st(), p(X), K, p(W), None, p(B), p(Y), N, p(R), N, n_req, 30, N, K, epi))
            torch.cuda.synchronize()
            acc = X.double() @ Wn.double().T + B.double()
            bf = lambda t: t.to(torch.bfloat16).double()
            if epi == 0:
                want = bf(acc)
            elif epi == 1:
                want = bf(R.double() + bf(acc))
            else:
                y, u = bf(acc[:, :N]), bf(acc[:, N:])
                want = bf(bf(y / (1 + torch.exp(-y))) * u)
            live = torch.zeros(32 * n_req, dtype=torch.bool, device=dev)
            for t in range(n_req):
                live[32 * t:32 * t + 30] = True
            err = (Y.double() - want)[live].abs().max().item()
            scale = want[live].abs().max().item()
            pad_ok = bool((Y[~live].float() == 7.0).all())
            ok = err <= scale * 2 ** -6 and pad_ok
            print(f"check N={N} K={K} n_req={n_req} epi={epi}: max err {err:.4g} of scale {scale:.4g} padding untouched {pad_ok} {'OK' if ok else 'FAIL'}", flush=True)
            assert ok


def timed(v, X, K, Ws, Y, N, M):
    for w in Ws[:2]:
        L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K, p(w), p(Y), N, M, N, K))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 3 * len(Ws)
    e0.record()
    for i in range(iters):
        L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K, p(Ws[i % len(Ws)]), p(Y), N, M, N, K))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, N, K, S in SHAPES:
    nbuf = max(2, int(1.5e9 // (N * K * 2)))
    Ws = [pack_weight((torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
    X = torch.randn(256, K, device=dev, dtype=torch.bfloat16)
    Y = torch.empty(256, N, device=dev, dtype=torch.bfloat16)
    res = []
    us = timed(10000 + S * 100, X, K, Ws, Y, N, 30)
    res.append(f"single M=30: {us:6.1f}us {N * K * 2 / us / 1e6:5.2f}TB/s")
    us4 = timed(90000 + S * 100 + 5, X, K, Ws, Y, N, 120)
    res.append(f"wide8 M=120: {us4:6.1f}us {N * K * 2 / us4 / 1e6:5.2f}TB/s")
    for S8 in sorted({S, max(1, S // 2), min(8, S * 2)}):
        us8 = timed(90000 + S8 * 100, X, K, Ws, Y, N, 240)
        res.append(f"c8 M=240 S={S8}: {us8:6.1f}us {N * K * 2 / us8 / 1e6:5.2f}TB/s = {us8 / us4:4.2f} x wide8 for 2 x the requests")
    print(f"{name:8s} [{N}x{K}] S={S} " + " | ".join(res), flush=True)
    del Ws
