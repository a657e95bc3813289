"""What does a second 32-row activation tile cost?  The layer's GEMMs through the product entry points at M = 30 (one tile) and
M = 60 (two tiles, weights still streamed once), weights rotated through > 1 GiB."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vispec_amd import lib as L, synth
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights, pack_weight
lib = L.load(); dev = torch.device("cuda:0"); T = synth.TINY
tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"]); dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
eng = Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev),
             DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev))
p = lambda t: None if t is None else C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [("qkv", 12288, 4096, "plain"), ("o_proj+norm", 4096, 4096, "norm"), ("gate_up", 11008, 4096, "swiglu"), ("down+norm", 4096, 11008, "norm"),
          ("lm_head", 32064, 4096, "plain")]
for name, N, K, kind in SHAPES:
    rows = 2 * N if kind == "swiglu" else N
    nbuf = max(3, int(1.3e9 // (rows * K * 2)))
    Ws = [pack_weight((torch.randn(rows, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
    nw = torch.ones(N, device=dev, dtype=torch.bfloat16)
    res = []
    for M in (30, 60):
        X = torch.randn(M, K, device=dev, dtype=torch.bfloat16); Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        R = torch.randn(M, N, device=dev, dtype=torch.bfloat16); Yn = torch.empty_like(Y)
        def run(i):
            w = Ws[i % nbuf]
            if kind == "norm":
                L.check(lib.vispec_gemm_skinny_norm(eng.h, st(), p(X), K, p(w), None, p(Y), N, p(R), N, p(nw), p(Yn), N, 1e-5, M, N, K))
            else:
                L.check(lib.vispec_gemm_skinny(eng.h, st(), p(X), K, p(w), None, p(Y), N, None, 0, M, N, K, 2 if kind == "swiglu" else 0))
        for i in range(nbuf): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 4 * nbuf
        e0.record(); [run(i) for i in range(it)]; e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / it)
    mb = rows * K * 2 / 1e6
    print(f"{name:12s} {mb:6.1f} MB  M=30 {res[0]:6.1f} us ({mb/res[0]:4.2f} TB/s) | M=60 {res[1]:6.1f} us ({mb/res[1]:4.2f} TB/s)  x{res[1]/res[0]:.2f}", flush=True)
    del Ws
