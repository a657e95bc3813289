"""torch SDPA back ends on the target prefill's attention shape (causal, L x L, H heads of 128).  python tools/sdpa_backend_probe.py [L] [H] [H_kv]"""
import sys, torch
from torch.nn.attention import SDPBackend, sdpa_kernel
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2704
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
HK = int(sys.argv[3]) if len(sys.argv) > 3 else H
dev = torch.device("cuda:0")
q = torch.randn(1, H, L, 128, device=dev, dtype=torch.bfloat16)
k = torch.randn(1, HK, L, 128, device=dev, dtype=torch.bfloat16)
v = torch.randn(1, HK, L, 128, device=dev, dtype=torch.bfloat16)
flops = 4 * L * L * 128 * H / 2
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel(be):
            f = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=(HK != H))
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"{name:10s} {us:8.1f} us  {flops / us / 1e6:6.0f} TFLOP/s (causal flops)", flush=True)
    except Exception as e:
        print(name, "failed:", str(e)[:120], flush=True)
