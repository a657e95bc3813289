import torch, sys
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2704
K = 4096
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = torch.randn(L, K, device=dev, dtype=torch.bfloat16)
for N in (22016, 11008, 22016 - 512, 22528, 16384, 5504, 13824 * 2, 13824, 18944 * 2, 18944):
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    us = t(lambda: torch.nn.functional.linear(x, w))
    out = torch.empty(L, N, device=dev, dtype=torch.bfloat16)
    print(f"N={N:6d}: {us:7.1f} us {2*L*N*K/us/1e6:6.0f} TF", flush=True)
w = torch.randn(22016, K, device=dev, dtype=torch.bfloat16) * 0.02
out = torch.empty(L, 22016, device=dev, dtype=torch.bfloat16)
def two():
    torch.mm(x, w[:11008].t(), out=out[:, :11008]) if False else None
us2 = t(lambda: (torch.nn.functional.linear(x, w[:11008]), torch.nn.functional.linear(x, w[11008:])))
print(f"two halves of 11008: {us2:7.1f} us {2*L*22016*K/us2/1e6:6.0f} TF")
us4 = t(lambda: [torch.nn.functional.linear(x, w[i*5504:(i+1)*5504]) for i in range(4)])
print(f"four quarters of 5504: {us4:7.1f} us {2*L*22016*K/us4/1e6:6.0f} TF")
