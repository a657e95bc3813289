import time, torch, numpy as np, os
print("cpus", os.cpu_count(), torch.__config__.parallel_info().split("\n")[0:6])
W = torch.randn(11008, 4096); x30 = torch.randn(30, 4096); x1 = torch.randn(1, 4096)
Wn = W.numpy(); xn = x30.numpy()
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    for x, nm in ((x30, "M30"), (x1, "M1")):
        torch.nn.functional.linear(x, W)
        t = time.perf_counter()
        for _ in range(5): torch.nn.functional.linear(x, W)
        dt = (time.perf_counter() - t) / 5
        print(f"threads {th:3d} {nm}: {dt*1e3:8.2f} ms  ({W.numel()*4/dt/1e9:6.1f} GB/s)", flush=True)
t = time.perf_counter()
for _ in range(3): xn @ Wn.T
print("numpy M30", (time.perf_counter() - t) / 3 * 1e3, "ms")
