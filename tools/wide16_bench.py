"""PROTOTYPE measurement: csrc/gemm_wide16.h (sixteen row blocks of one (split, K-quarter) per workgroup: one byte of X per FOUR bytes of W,
quarter sums written as fp32 partials + a reduce) next to the wide-cohort kernel of the product path, shape by shape at M = 120, kernel
alone, weights rotated through > 1 GB; and the bit-identity check of (wide16 + reduce) against vispec_gemm_cohort on the same operands.
    python tools/wide16_bench.py"""
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
SHAPES = [("qkv", 12288, 4096, 1), ("o_proj", 4096, 4096, 4), ("gate_up", 22016, 4096, 1), ("down", 4096, 11008, 4), ("lm_head", 32064, 4096, 1)]
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(v, Xs, ldx, Ws, Y, N, K):
    for w in Ws[:2]:
        L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(Xs), ldx, p(w), p(Y), N, 120, N, K))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 3 * len(Ws)
    e0.record()
    for i in range(iters):
        L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(Xs), ldx, p(Ws[i % len(Ws)]), p(Y), N, 120, N, K))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, N, K, S in SHAPES:
    nbuf = max(2, int(1.5e9 // (N * K * 2)))
    Ws = [pack_weight((torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
    X = torch.randn(128, K, device=dev, dtype=torch.bfloat16)
    Y = torch.full((128, N), 7.0, device=dev, dtype=torch.bfloat16)
    part = torch.zeros(4 * S, 128, N, device=dev, dtype=torch.float32)
    us_wide = timed(90000 + S * 100, X, K, Ws, Y, N, K)
    us_main = timed(90000 + S * 100 + 5, X, K, Ws, part, N, K)
    us_nost = timed(90000 + S * 100 + 7, X, K, Ws, part, N, K)
    us_red = timed(90000 + S * 100 + 6, part, K, Ws, Y, N, K)
    # bit identity: wide16 + reduce == the product path's cohort GEMM (epilogue NONE, no bias), live rows of the four tiles
    L.check(lib.vispec_gemm_skinny_tune(eng.h, 90000 + S * 100 + 5, st(), p(X), K, p(Ws[0]), p(part), N, 120, N, K))
    L.check(lib.vispec_gemm_skinny_tune(eng.h, 90000 + S * 100 + 6, st(), p(part), K, p(Ws[0]), p(Y), N, 120, N, K))
    Y2 = torch.full((128, N), 7.0, device=dev, dtype=torch.bfloat16)
    L.check(lib.vispec_gemm_cohort(eng.h, st(), p(X), K, p(Ws[0]), None, None, p(Y2), N, None, N, 4, 30, N, K, 0))
    torch.cuda.synchronize()
    live = torch.arange(128, device=dev) % 32 < 30
    same = bool((Y[live].view(torch.int16) == Y2[live].view(torch.int16)).all())
    nz = float(Y2[live].float().abs().mean())
    wg = (N // 32 + 15) // 16 * 4 * S
    print(f"{name:8s} [{N}x{K}] S={S} | wide (product): {us_wide:6.1f}us {N * K * 2 / us_wide / 1e6:5.2f}TB/s | wide16 main ({wg} wgs): {us_main:6.1f}us "
          f"{N * K * 2 / us_main / 1e6:5.2f}TB/s (without its partial stores {us_nost:5.1f}us) + reduce {us_red:5.1f}us ({part.numel() * 4 / 1e6:.0f} MB of partials) | bit-identical to vispec_gemm_cohort: {same} (mean |y| {nz:.3f})",
          flush=True)
    del Ws, part
