"""Can the fp8-weight prefill run on the fp8 MFMA through the library?  torch._scaled_mm (hipBLASLt) with e4m3 x e4m3 operands and
row-wise scales (per activation row, per output channel) -> bf16, against the bf16 x bf16 GEMM and the round-2 form (bf16 activations x the
codes up-converted to bf16, fp32 out) at the Qwen2.5-VL-7B prefill shapes (L = 2124).      python tools/scaled_mm_probe.py"""
import torch

dev = torch.device("cuda:0")
L = 2124
SHAPES = [("qkv", 4608, 3584), ("o_proj", 3584, 3584), ("gate_up", 37888, 3584), ("down", 3584, 18944)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print(torch.__version__, torch.cuda.get_device_name(0))
for name, N, K in SHAPES:
    x = (torch.randn(L, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    sw = (w.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0)
    qw = (w.float() / sw).to(torch.float8_e4m3fn)
    sx = (x.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0)
    qx = (x.float() / sx).to(torch.float8_e4m3fn)
    t_bf16 = timed(lambda: torch.nn.functional.linear(x, w))
    res = {}
    for mode in ("rowwise", "tensor"):
        try:
            if mode == "rowwise":
                f = lambda: torch._scaled_mm(qx, qw.t(), scale_a=sx, scale_b=sw.t().contiguous(), out_dtype=torch.bfloat16)
            else:
                one = torch.ones((), device=dev)
                f = lambda: torch._scaled_mm(qx, qw.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
            y = f()
            ref = ((qx.float() @ qw.float().t()) * sx * sw.t()) if mode == "rowwise" else (qx.float() @ qw.float().t())
            err = float((y.float() - ref).abs().max() / ref.abs().max())
            res[mode] = f"{timed(f):7.1f} us (max err {err:.1e} of scale)"
        except Exception as e:  # noqa: BLE001
            res[mode] = f"FAILED: {str(e)[:120]}"
    fl = 2.0 * L * N * K
    print(f"{name:8s} [{L} x {K}] x [{N} x {K}]^T: bf16 {t_bf16:7.1f} us ({fl / t_bf16 / 1e6:5.0f} TFLOP/s) | fp8 row-wise scales {res['rowwise']} | fp8 unit scales {res['tensor']}", flush=True)
    tq = timed(lambda: (x.float() / (x.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0)).to(torch.float8_e4m3fn))
    print(f"         torch-side row quantisation of x: {tq:6.1f} us")
