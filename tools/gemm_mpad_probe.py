import torch
dev = torch.device("cuda:0")
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up+512", 22528, 4096), ("down", 4096, 11008)):
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    res = []
    for M in (2704, 2720, 2752, 2816, 3072):
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        us = t(lambda: torch.nn.functional.linear(x, w))
        res.append(f"M={M}: {us:6.1f}us ({2*2704*N*K/us/1e6:5.0f} useful TF)")
    print(f"{name:12s} | " + " | ".join(res), flush=True)
