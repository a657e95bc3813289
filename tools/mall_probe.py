"""Does the 256 MiB Infinity Cache serve a weight stream that was touched just before?  Times the skinny GEMM on ONE weight
buffer re-used back to back (cache-warm) vs rotated through > 1 GiB (cold), and after a plain read-only "touch" pass."""
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
M = 30
for name, N, K in [("o_proj", 4096, 4096), ("qkv", 12288, 4096), ("down", 4096, 11008), ("gate_up/2", 11008, 4096), ("lm_head", 32064, 4096)]:
    nbuf = max(3, int(1.5e9 // (N * K * 2)))
    Ws = [pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
    X = torch.randn(M, K, device=dev, dtype=torch.bfloat16); Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    S = 4 if N <= 4096 else 1
    v = 10000 + S * 100
    def run(i): L.check(lib.vispec_gemm_skinny_tune(eng.h, v, st(), p(X), K, p(Ws[i]), p(Y), N, M, N, K))
    def timed(fn, iters):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [fn(i) for i in range(iters)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / iters
    for i in range(nbuf): run(i)
    cold = timed(lambda i: run(i % nbuf), 3 * nbuf)
    warm = timed(lambda i: run(0), 3 * nbuf)
    # touch pass (a plain elementwise read) of the next buffer, then the GEMM on it: [touch(i+1) | gemm(i)] pairs
    def pair(i):
        Ws[(i + 1) % nbuf].view(torch.int32).sum()  # reads the whole buffer once
        run(i % nbuf)  # touched one iteration earlier
    both = timed(pair, 3 * nbuf)
    touch = timed(lambda i: Ws[(i + 1) % nbuf].view(torch.int32).sum(), 3 * nbuf)
    mb = N * K * 2 / 1e6
    print(f"{name:10s} {mb:6.1f} MB  cold {cold:6.1f} us ({mb/cold:5.2f} TB/s) | same buffer {warm:6.1f} us ({mb/warm:5.2f} TB/s) | "
          f"touch alone {touch:6.1f} us, touch(next)+gemm(touched) {both:6.1f} us -> gemm after touch ~{both-touch:6.1f} us", flush=True)
    del Ws
