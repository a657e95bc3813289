"""log-softmax + top-k at the BASELINE vocabularies: one-launch row kernel against the three-pass form (VISPEC_LSTK_ROW_MAX_V=0)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vispec_amd import lib as L, synth
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights
lib = L.load(); dev = torch.device("cuda:0"); T = synth.TINY
tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"]); dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
eng = Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev),
             DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev))
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for V in (32064, 152064):
    for M in (1, 8):
        x = (torch.randn(M, V, device=dev) * 3).to(torch.bfloat16)
        idx = torch.zeros(M, 8, dtype=torch.int32, device=dev); lp = torch.zeros(M, 8, device=dev)
        for _ in range(5): L.check(lib.vispec_logsoftmax_topk(eng.h, st(), p(x), V, M, V, 8, p(idx), p(lp)))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): L.check(lib.vispec_logsoftmax_topk(eng.h, st(), p(x), V, M, V, 8, p(idx), p(lp)))
        e1.record(); torch.cuda.synchronize()
        print(f"V={V} M={M}: {e0.elapsed_time(e1) * 1e3 / 200:.1f} us per call", flush=True)
