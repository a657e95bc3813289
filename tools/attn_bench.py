"""Tree-attention microbenchmark at the LLaVA-7B verify shape (32 heads, 30 query rows, tail 30): partial / reduce kernel time vs
context length, K/V rotated through several caches so HBM (not L2 / Infinity Cache) serves them.  env VISPEC_ATT_KPW=keys per workgroup."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vispec_amd import lib as L, synth
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights
lib = L.load(); dev = torch.device("cuda:0"); T = synth.TINY
tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"]); dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
eng = Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev),
             DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev))
p = lambda t: C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, Hkv, M, tail, S = 32, 32, 30, 30, 4096
NB = 12  # 12 x (2 x 33.5 MB) of K/V > 256 MiB
Ks = [torch.randn(Hkv, S, 128, device=dev).to(torch.bfloat16) for _ in range(NB)]
Vs = [torch.randn(Hkv, S, 128, device=dev).to(torch.bfloat16) for _ in range(NB)]
q = torch.randn(M, H * 128, device=dev).to(torch.bfloat16); out = torch.empty_like(q)
mask = torch.tensor([(1 << (i + 1)) - 1 for i in range(M)], dtype=torch.int64, device=dev)
for n in (64, 256, 512, 1024, 2048, 3000, 4000):
    pre = torch.tensor([n], dtype=torch.int32, device=dev)
    def run(i):
        L.check(lib.vispec_tree_attention(eng.h, st(), p(q), H * 128, p(Ks[i % NB]), p(Vs[i % NB]), S, H, Hkv, 128, M, p(pre), tail, p(mask), p(out), H * 128, 1))
    for i in range(NB): run(i)
    torch.cuda.synchronize()
    eng.prof_enable(True)
    for i in range(4 * NB): run(i)
    rep = eng.prof_report(); eng.prof_enable(False)
    a, r = rep["attn_partial"], rep["attn_reduce"]
    mb = 2 * Hkv * (n + tail) * 128 * 2 / 1e6
    us = 1e3 * a["ms"] / a["launches"]
    print(f"n={n:5d} KV {mb:6.1f} MB  partial {us:6.1f} us ({mb / us:5.2f} TB/s)  reduce {1e3 * r['ms'] / r['launches']:5.1f} us", flush=True)
