"""Is the skinny GEMM limited by workgroup/CU imbalance?  TB/s vs number of 32-row tiles (K = 4096, M = 30, S = 1, 4 waves)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vispec_amd import lib as L, synth
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights, pack_weight
lib = L.load(); dev = torch.device("cuda:0"); T = synth.TINY
tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"]); dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
eng = Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev),
             DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev))
p = lambda t: C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
K, M = 4096, 30
for tiles in (128, 192, 256, 288, 320, 344, 384, 448, 512, 576, 640, 688, 768, 896, 1002, 1024, 1280, 2048):
    N = tiles * 32
    if N > 16384 * 1: S_list = [1]
    S_list = [1] if N > 16384 else [1, 2]
    nbuf = max(2, int(1.2e9 // (N * K * 2)))
    Ws = [pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
    X = torch.randn(M, K, device=dev, dtype=torch.bfloat16); Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = []
    for S in S_list:
        v = 10000 + S * 100  # GEMM kernel alone (no reduce launch)
        for w in Ws[:2]: L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K, p(w), p(Y), N, M, N, K))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 4 * nbuf
        e0.record()
        for i in range(iters): L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K, p(Ws[i % nbuf]), p(Y), N, M, N, K))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out.append(f"S={S}: {us:6.1f}us {N*K*2/us/1e6:5.2f}TB/s")
    print(f"tiles {tiles:5d} ({tiles/256:4.2f}/CU) N={N:6d} " + " | ".join(out), flush=True)
    del Ws
