"""vispec_prefill_attention against torch's flash SDPA on the target prefill's shapes (causal flops = 2 L^2 128 H).   GPU only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vispec_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
for Ln, H, HK, eager in ((2704, 32, 32, 1), (3488, 32, 32, 1), (2704, 40, 40, 1), (1584, 28, 4, 0), (2124, 28, 4, 0), (643, 32, 32, 1)):
    S = 8192
    qkv = torch.randn(Ln, (H + 2 * HK) * 128, device=dev, dtype=torch.bfloat16)
    kc = torch.randn(HK, S, 128, device=dev, dtype=torch.bfloat16)
    vc = torch.randn(HK, S, 128, device=dev, dtype=torch.bfloat16)
    out = torch.empty(Ln, H * 128, device=dev, dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.check(lib.vispec_prefill_attention(None, st, p(qkv), qkv.shape[1], p(kc), p(vc), S, H, HK, Ln, p(out), H * 128, eager))
    q4 = qkv[:, : H * 128].view(Ln, H, 128).transpose(0, 1)[None]
    g = lambda: torch.nn.functional.scaled_dot_product_attention(q4, kc[None, :, :Ln], vc[None, :, :Ln], is_causal=True, enable_gqa=(H != HK))
    res = []
    for name, fn_ in (("hip", f), ("torch sdpa", g)):
        try:
            for _ in range(3): fn_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn_()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            res.append(f"{name} {us:7.1f} us {2 * Ln * Ln * 128 * H / us / 1e6:5.0f} TF")
        except Exception as e:
            res.append(f"{name} failed: {str(e)[:60]}")
    ref = g() if H == HK else None
    err = "" if ref is None else f" | max |hip - sdpa| {float((out.view(Ln, H, 128).transpose(0, 1).float() - ref[0].float()).abs().max()):.4f}"
    print(f"L={Ln} H={H}/{HK} eager={eager}: " + " | ".join(res) + err, flush=True)
