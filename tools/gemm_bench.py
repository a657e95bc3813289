"""Microbenchmark of the W32 skinny GEMM decompositions at the LLaVA-7B shapes (weights rotated through > 1 GiB so neither L2 nor
the 256 MiB infinity cache can serve them).  python tools/gemm_bench.py [variants...]   variant = S*100 + NWcode*10 + UNcode"""
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
SHAPES = [("o_proj", 4096, 4096), ("qkv", 12288, 4096), ("down", 4096, 11008), ("gate_up", 22016, 4096), ("lm_head", 32064, 4096)]
variants = [int(v) for v in sys.argv[1:]] or [100, 101, 110, 111, 201, 401, 411, 402, 801]
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M in [int(m) for m in os.environ.get("MS", "30,8").split(",")]:
    for name, N, K in SHAPES:
        nbuf = max(2, int(1.5e9 // (N * K * 2)))
        Ws = [pack_weight((torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
        PAD = int(os.environ.get("XPAD", "0"))
        Xfull = torch.randn(M, K + PAD, device=dev, dtype=torch.bfloat16)
        X = Xfull
        Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = []
        for v in variants:
            if v // 10000 != 7 and ((v % 10000) // 100) * (128 if M > 64 else 64 if M > 32 else 32) * N > 8 * 64 * 16384:
                continue
            for w in Ws[:2]:
                L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K + PAD, p(w), p(Y), N, M, N, K))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 3 * nbuf
            e0.record()
            for i in range(iters):
                L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K + PAD, p(Ws[i % nbuf]), p(Y), N, M, N, K))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            res.append(f"{v}:{us:6.1f}us {N * K * 2 / us / 1e6:5.2f}TB/s")
        print(f"M={M:2d} {name:8s} [{N}x{K}] " + " | ".join(res), flush=True)
        del Ws
