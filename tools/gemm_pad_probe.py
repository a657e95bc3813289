"""hipBLASLt on the prefill's GEMM shapes with zero rows appended to the weight (N + pad): where does the heuristic pick a poor kernel?
python tools/gemm_pad_probe.py [L]"""
import sys, torch
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2704
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008),
                   ("qkv13", 15360, 5120), ("o13", 5120, 5120), ("gu13", 27648, 5120), ("down13", 5120, 13824),
                   ("qkvQ", 4608, 3584), ("oQ", 3584, 3584), ("guQ", 37888, 3584), ("downQ", 3584, 18944)):
    x = torch.randn(L, K, device=dev, dtype=torch.bfloat16)
    res = []
    for pad in (0, 128, 256, 512, 768):
        w = torch.randn(N + pad, K, device=dev, dtype=torch.bfloat16) * 0.02
        us = t(lambda: torch.nn.functional.linear(x, w))
        res.append(f"+{pad}: {us:6.1f}us {2*L*N*K/us/1e6:5.0f}TF")
    print(f"{name:7s} N={N:6d} K={K:6d} | " + " | ".join(res), flush=True)
