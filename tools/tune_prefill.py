"""Offline tuning of the target prefill's library GEMMs (torch's TunableOp: every rocBLAS / hipBLASLt solution of a shape is timed, the
fastest is recorded) -> vispec_amd/tunable/prefill_gemms_gfx950.csv, the table vispec_amd/model/target.py looks shapes up in at run time
(lookup only: nothing is timed inside a serving process, every rank and lane runs the same recorded solution).

    python tools/tune_prefill.py [out.csv]        (GPU box, idle GPU, one process; ~1-2 min per model)

Shapes: the four GEMMs per layer (q|k|v, o_proj, gate|up, down) + the last-row LM head of every BASELINE.json model at the prompt lengths
bench.py builds (LLaVA-7B / 13B: L = 2704 and 3488 = --n-img 2928; Qwen2.5-VL-7B: L = 1584 multi-turn, L = 2124 high-res), bf16.  A prompt of
another length runs the libraries' default kernels.  Prints the per-shape times of default vs recorded solutions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import torch.cuda.tunable as tn  # noqa: E402

from vispec_amd.model.target import PREFILL_GEMM_FILE  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else PREFILL_GEMM_FILE
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.remove(out)
dev = torch.device("cuda:0")
MODELS = {  # D, q|k|v rows, I, (V,), bias, prompt lengths
    "llava7b": (4096, 3 * 4096, 11008, False, (2704, 3488)),
    "llava13b": (5120, 3 * 5120, 13824, False, (2704, 3488)),
    "qwen7b": (3584, (28 + 2 * 4) * 128, 18944, True, (1584, 2124)),
}
shapes = []
for name, (D, QKV, I, bias, Ls) in MODELS.items():
    for L in Ls:
        shapes += [(name, "qkv", L, QKV, D, bias), (name, "o_proj", L, D, D, False), (name, "gate_up", L, 2 * I, D, False), (name, "down", L, D, I, False)]


_calls = {}


def run(x, w, b):
    if not FP8:
        return F.linear(x, w, b)
    key = (x.data_ptr(), w.data_ptr())
    if key not in _calls:
        _calls[key] = fp8_call(x, w, b)
    return _calls[key]()


def timed(x, w, b, iters=20):
    for _ in range(3):
        run(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


FP8 = os.environ.get("FP8", "0") == "1"  # FP8=1 python tools/tune_prefill.py /tmp/fp8.csv: ONLY the W8A8 prefill's fp8 x fp8 GEMMs (torch._scaled_mm,
# row-wise scales: vispec_amd/model/target.py scaled_linear(q8=...)) of the Qwen2.5-VL-7B shapes, into a file of their own (its entry lines are
# appended to the committed table by hand: the bf16 entries stay as recorded)
if FP8:
    shapes = [(n, k, L, N, K, b) for (n, k, L, N, K, b) in shapes if n == "qwen7b" and k != "o_proj"]


def fp8_call(x, w, b):
    sx = (x.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0)
    sw = (w.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0)
    qx, qw = (x.float() / sx).to(torch.float8_e4m3fn), (w.float() / sw).to(torch.float8_e4m3fn)
    swt = sw.t().contiguous()
    return lambda: torch._scaled_mm(qx, qw.t(), scale_a=sx, scale_b=swt, bias=b, out_dtype=torch.bfloat16)


tensors = {}
for name, kind, L, N, K, bias in shapes:
    g = torch.Generator(device="cpu").manual_seed(L + N + K)
    tensors[(name, kind, L)] = ((torch.randn(L, K, generator=g) * 0.5).to(dev, torch.bfloat16), (torch.randn(N, K, generator=g) * 0.02).to(dev, torch.bfloat16),
                                (torch.randn(N, generator=g) * 0.02).to(dev, torch.bfloat16) if bias else None)
base = {k: timed(*v) for k, v in tensors.items()}
tn.enable(True)
tn.tuning_enable(True)
tn.set_filename(out)
tn.set_max_tuning_duration(30)      # ms per candidate solution
tn.set_max_tuning_iterations(20)
t0 = time.time()
for k, v in tensors.items():
    run(*v)                         # first call of a shape tunes it
torch.cuda.synchronize()
t_tune = time.time() - t0
tn.tuning_enable(False)
tuned = {k: timed(*v) for k, v in tensors.items()}
print(f"tuned {len(tensors)} shapes in {t_tune:.1f} s -> {out}")
tot_b = tot_t = 0.0
for (name, kind, L, N, K, bias) in shapes:
    b_, t_ = base[(name, kind, L)], tuned[(name, kind, L)]
    fl = 2.0 * L * N * K
    print(f"{name:9s} {kind:8s} [{L} x {K}] x [{N} x {K}]^T{' + bias' if bias else '':7s}: default {b_:7.1f} us ({fl / b_ / 1e6:6.0f} TFLOP/s)  recorded {t_:7.1f} us "
          f"({fl / t_ / 1e6:6.0f} TFLOP/s)  {b_ / t_:5.2f}x", flush=True)
for r in tn.get_results():
    print("result:", r)
print("validators:", tn.get_validators())
# TunableOp writes its file when the process ends (or as it goes, depending on the PyTorch version); make sure it exists before we leave
del tensors
