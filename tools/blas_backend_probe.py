import torch, time, sys
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2704
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    tot = 0
    out = []
    for name, N, K in shapes:
        x = torch.randn(L, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        for _ in range(3): y = torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = torch.nn.functional.linear(x, w)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tot += us
        out.append(f"{name} {us:.0f}us {2*L*N*K/us/1e6:.0f}TF")
    print(lib, " | ".join(out), f"| layer {tot:.0f} us", flush=True)
