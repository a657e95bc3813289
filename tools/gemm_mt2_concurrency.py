"""Two-tile (cohort) skinny GEMM under lane concurrency: NT=1 (one row block per workgroup, variant 1SSxx) against NT=2 (two row blocks
sharing each staged activation group, variant 5SSxx) at M = 60 on 1..4 streams, LLaVA-7B layer shapes, kernels alone (no reduce)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vispec_amd import lib as L, synth
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights, pack_weight
lib = L.load(); dev = torch.device("cuda:0"); T = synth.TINY
tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"]); dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
mk = lambda: Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev),
                    DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev))
NS = 4
engs = [mk() for _ in range(NS)]
p = lambda t: C.c_void_p(t.data_ptr())
M = int(os.environ.get("M", "60"))
UNC = int(os.environ.get("UNC", "0"))  # wide kernel (M > 96) only: 1 = no activation DMAs, 2 = no weight loads, 3 = two row blocks per workgroup
SHAPES = [("qkv", 12288, 4096, 1), ("o_proj", 4096, 4096, 4), ("gate_up", 22016, 4096, 1), ("down", 4096, 11008, 4)]
NB = 6
W = {n: [pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(NB)] for n, N, K, S in SHAPES}
X = {K: torch.randn(max(M, 128), K, device=dev, dtype=torch.bfloat16) for K in (4096, 11008)}  # (the wide kernel reads whole 32-row tiles)
Y = torch.empty(128, 22016, device=dev, dtype=torch.bfloat16)
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
bytes_layer = sum(N * K * 2 for n, N, K, S in SHAPES)
def layer(si, it, code):
    s = C.c_void_p(streams[si].cuda_stream)
    for n, N, K, S in SHAPES:
        L.check(lib.vispec_gemm_skinny_tune(engs[si].h, code * 10000 + S * 100 + UNC, s, p(X[K]), K, p(W[n][(it * NS + si) % NB]), p(Y), N, M, N, K))
for code, name in (((8, "MT=4 NT=1"), (9, "wide (16 waves, shared staging)")) if M > 96 else ((9, "wide NL=3"),) if M > 64 else ((1, "NT=1"), (5, "NT=2"), (6, "NT=2, half the activation loads (upper bound)"))):
    for ns in (1, 2, 3, 4):
        for it in range(3):
            for si in range(ns): layer(si, it, code)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 40
        e0.record()
        for si in range(ns): streams[si].wait_event(e0)
        for it in range(iters):
            for si in range(ns): layer(si, it, code)
        for si in range(ns): torch.cuda.current_stream().wait_stream(streams[si])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        print(f"{name} M={M} {ns} stream(s): {us / iters / ns:7.1f} us per layer-equivalent ({bytes_layer / 1e6:.0f} MB) -> aggregate {bytes_layer * iters * ns / us / 1e6:5.2f} TB/s", flush=True)
