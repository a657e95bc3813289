"""Why are the fp8 cohort GEMMs no faster than the bf16 ones?  The four-row-block wide kernel (csrc/gemm_wide.h) alone, cohort of four
(M = 4 x 30 rows), no epilogue, S = 1, over K and over the number of workgroups, for the three weight / activation formats:
bf16 x bf16, e4m3 weights x bf16 activations (W8A16), e4m3 x e4m3 on the f8f6f4 MFMA (W8A8, its quantisation pass left out).
A straight-line fit T = a + b K per format and grid separates the per-launch constant from the per-k cost.
    python tools/fp8_k_sweep.py            (GPU box; RB=8: the eight-row-block forms; COHORT=8: the cohort-8 kernel of csrc/gemm_c8.h, M = 8 x 30 rows)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vispec_amd import lib as L, synth  # noqa: E402
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights, pack_weight, pack_weight_fp8  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
D, H, I, V, NL = 512, 4, 16384, 1024, 1
tcfg = TargetConfig(hidden_size=D, num_heads=H, num_kv_heads=2, intermediate_size=I, vocab_size=V, num_layers=NL, max_position_embeddings=512)
dcfg = DraftConfig(hidden_size=D, num_heads=H, intermediate_size=704, vocab_size=V, max_position_embeddings=512)
eng = Engine(tcfg, dcfg, TargetWeights.from_state_dict(tcfg, synth.make_target_weights(D, H, I, V, NL, seed=0, H_kv=2), dev),
             DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(D, H, 704, V, seed=1), 2, dev))
RB = int(os.environ.get("RB", "4"))
COHORT = int(os.environ.get("COHORT", "4"))
if COHORT == 8:
    RB = 8
eng.set_wide_row_blocks(RB)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
os.environ["VISPEC_A8_SKIP_QUANT"] = "1"


def timed(call, Ws):
    for w in Ws[:2]:
        call(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = max(30, 3 * len(Ws))
    e0.record()
    for i in range(iters):
        call(Ws[i % len(Ws)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


KS = [1024, 2048, 3584, 7168, 14336]
for n_req, m_tile in ((COHORT, 30),) if COHORT != 4 else (((4, 30), (1, 30)) if RB == 4 else ((4, 30),)):  # (RB=8 python tools/fp8_k_sweep.py: the eight-row-block forms)
    for N in (8192, 32768, 65536):
        rows = {}
        for K in KS:
            nb16 = int(min(24, max(2, 6e8 // (N * K * 2)))) if n_req >= 4 else 2
            nb8 = int(min(24, max(2, 6e8 // (N * K)))) if n_req >= 4 else 2
            W16 = [pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nb16)]
            W8 = [pack_weight_fp8(torch.randint(0, 120, (N, K), device=dev, dtype=torch.uint8)) for _ in range(nb8)]
            sc = torch.full((N,), 0.01, device=dev, dtype=torch.float32)
            M = 32 * (n_req - 1) + m_tile
            X = torch.randn(256, K, device=dev, dtype=torch.bfloat16)
            Y = torch.empty(256, N, device=dev, dtype=torch.bfloat16)
            if n_req >= 4:
                bf = lambda w: L.check(lib.vispec_gemm_cohort(eng.h, st(), p(X), K, p(w), None, None, p(Y), N, None, 0, n_req, m_tile, N, K, 0))
                a16 = lambda w: L.check(lib.vispec_gemm_cohort(eng.h, st(), p(X), K, p(w), p(sc), None, p(Y), N, None, 0, n_req, m_tile, N, K, 0))
            else:
                bf = lambda w: L.check(lib.vispec_gemm_skinny(eng.h, st(), p(X), K, p(w), None, p(Y), N, None, 0, M, N, K, 0))
                a16 = lambda w: L.check(lib.vispec_gemm_skinny_fp8(eng.h, st(), p(X), K, p(w), p(sc), None, p(Y), N, None, 0, M, N, K, 0))
            a8 = lambda w: L.check(lib.vispec_gemm_fp8a8(eng.h, st(), p(X), K, p(w), p(sc), None, p(Y), N, None, 0, n_req, m_tile, M, N, K, 0, None, None,
                                                          C.c_float(1e-6)))
            a8(W8[0])  # (fills the quantised scratch once... with the skip flag it holds whatever it held: timing only)
            rows[K] = (timed(bf, W16), timed(a16, W8), timed(a8, W8))
            del W16, W8
        wgs = (N // 32 + RB - 1) // RB if n_req >= 4 else N // 32
        print(f"n_req {n_req}  N = {N:6d} ({wgs} workgroups)  us at K = {KS}:")
        for i, name in enumerate(("bf16", "W8A16", "W8A8")):
            t = np.array([rows[K][i] for K in KS])
            b, a = np.polyfit(np.array(KS, float), t, 1)
            print(f"   {name:6s} " + " ".join(f"{v:7.1f}" for v in t) + f"   fit: {a:5.1f} us + {b * 1024:5.2f} us per 1024 k", flush=True)
