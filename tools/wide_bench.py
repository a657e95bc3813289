"""The wide-cohort GEMM (csrc/gemm_wide.h) shape by shape at M = 120 (four requests), kernel alone, weights rotated through > 1 GB:
variant 9SS00 = the kernel, 9SS01 = without its activation DMAs, 9SS02 = without its weight loads (wrong results on purpose: what each
of the two streams costs), 9SS03 / 9SS04 = two / three row blocks per workgroup instead of four, next to the single-request kernel (1SS00, M = 30) on the same weights.
    python tools/wide_bench.py [extra variant digits ...]"""
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
              # Qwen2.5-VL-7B (BASELINE configs 3 / 5): GQA 28/4 -> 4608 q|k|v rows, I = 18944, V = 152064; split factors = choose_split's
              "qwen7b": [("qkv", 4608, 3584, 1), ("o_proj", 3584, 3584, 4), ("gate_up", 37888, 3584, 1), ("down", 3584, 18944, 4), ("lm_head_131072_of_152064", 131072, 3584, 1)]}  # (the tune entry point's partial workspace holds 128 x 131072 floats)
SHAPES = SHAPE_SETS[os.environ.get("SHAPES", "llava7b")]
UN = [int(v) for v in sys.argv[1:]] or [0, 4, 3, 5]
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


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
    X = torch.randn(128, K, device=dev, dtype=torch.bfloat16)
    Y = torch.empty(128, N, device=dev, dtype=torch.bfloat16)
    res = []
    us = timed(10000 + S * 100, X, K, Ws, Y, N, 30)
    res.append(f"single M=30: {us:6.1f}us {N * K * 2 / us / 1e6:5.2f}TB/s")
    for u in UN:
        us = timed(90000 + S * 100 + u, X, K, Ws, Y, N, 120)
        res.append(f"wide/{u}: {us:6.1f}us {N * K * 2 / us / 1e6:5.2f}TB/s")
    print(f"{name:8s} [{N}x{K}] S={S} wgs={(N // 32 + 3) // 4 * S:4d} " + " | ".join(res), flush=True)
    del Ws
