"""Does partitioning the chip between the lanes help the cohort GEMMs?  (round 6)

Finding behind it (tools/wg_clock.py): with four lanes every cohort-8 workgroup runs 1.5-1.75 x slower than alone although HBM carries ~57 % of
what the chip streams.  Hypothesis: every XCD's 4 MB L2 serves workgroups of ALL lanes (dispatch is round-robin over the XCDs), so it has to
hold the 2 MB activation block of each of them (4 x 2 MB, 5.6 MB for down_proj) next to their weight and K/V streams — the activation
re-reads (one block per workgroup: as many bytes as the weight stream) then miss L2 and travel over the fabric.

This probe launches the timed region's dominant GEMMs (vispec_gemm_cohort, eight requests, LLaVA-7B shapes, every launch on another weight
buffer) on R streams at once,
  (a) plain streams (any workgroup on any CU), and
  (b) streams created with hipExtStreamCreateWithCUMask so that lane l only uses the XCDs of its share (8 / R XCDs each),
in both candidate layouts of the mask bits ("striped": bit i = CU i // 8 of XCD i % 8; "blocked": bit i = CU i % 32 of XCD i // 32), and prints
us per launch per stream.  With VISPEC_LIB_VARIANT=wgclk it also prints which XCC ids each lane's workgroups ran on — the layout under
which every lane reports only its own XCDs is the driver's.

    python tools/cu_mask_probe.py [R=4]"""
import ctypes as C
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vispec_amd import lib as L, synth  # noqa: E402
from vispec_amd.engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights, pack_weight, swiglu_order  # noqa: E402
from vispec_amd.evaluation.bench_launch import masked_stream  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = L.load()
    T = synth.TINY
    tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"])
    dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
    tw = TargetWeights.from_state_dict(tcfg, synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"]), dev)
    dw = DraftWeightsDev.from_state_dict(dcfg, synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"]), 2, dev)
    engs = [Engine(tcfg, dcfg, tw, dw) for _ in range(R)]  # (one ctx per lane: the split-K GEMMs use the ctx's partial workspace)
    for e in engs:
        e.lib.vispec_set_wide_row_blocks(e.h, 84)
    p = lambda t: C.c_void_p(t.data_ptr())
    shapes = [("gate|up (SwiGLU)", 11008, 4096, 2), ("q|k|v as plain GEMM", 12288, 4096, 0), ("o_proj (+residual, split-K)", 4096, 4096, 1),
              ("down (+residual, split-K)", 4096, 11008, 1)]
    wg = hasattr(lib, "vispec_debug_wgclock_set")
    out = {}
    for name, N, K, epi in shapes:
        rows = 2 * N if epi == 2 else N
        nbuf = max(2, int(1.2e9 // (rows * K * 2)))
        Ws = []
        for _ in range(nbuf):
            w = (torch.randn(rows, K, device=dev) * 0.02).to(torch.bfloat16)
            Ws.append(pack_weight(swiglu_order(w) if epi == 2 else w))
            del w
        X = [torch.randn(256, K, device=dev, dtype=torch.bfloat16) for _ in range(R)]
        Y = [torch.empty(256, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
        Rr = [torch.randn(256, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
        res = {}
        # "plain-shared-weights": every lane's launch `it` streams the SAME weight buffer (the lanes run the same layer in lockstep): the second
        # to fourth reader can hit the Infinity Cache — what merging the lanes' launches would buy if HBM is what binds them
        # "plain-shared-x": every lane multiplies the SAME activation block (an XCD's L2 then holds one 2 MB block instead of one per lane): what the
        # lanes' activation blocks cost each other in L2
        modes = ("plain", "plain-shared-weights", "plain-shared-x") if os.environ.get("PROBE_MASKS", "0") == "0" else ("plain", "masked-striped", "masked-blocked")
        for mode in modes:
            os.environ["VISPEC_CU_MASK_LAYOUT"] = mode.split("-")[-1]
            streams = [torch.cuda.Stream(dev) if mode.startswith("plain") else masked_stream(dev, lane, R) for lane in range(R)]
            shared, shared_x = mode == "plain-shared-weights", mode == "plain-shared-x"

            def launch(lane, it):
                L.check(lib.vispec_gemm_cohort(engs[lane].h, C.c_void_p(streams[lane].cuda_stream), p(X[0 if shared_x else lane]), K,
                                               p(Ws[(it if shared else it * R + lane) % nbuf]), None, None, p(Y[lane]), N, p(Rr[lane]), N, 8, 30, N, K, epi))
            for it in range(3):
                for lane in range(R):
                    launch(lane, it)
            torch.cuda.synchronize()
            if wg:
                REC = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("kid", "<u4"), ("tag", "<u4"), ("blk", "<u4"), ("nblk", "<u4"), ("hw", "<u4"), ("xcc", "<u4"),
                                ("p0", "<u4"), ("p1", "<u4")])
                buf = torch.zeros((1 << 20) * REC.itemsize, dtype=torch.uint8, device=dev)
                L.check(lib.vispec_debug_wgclock_set(p(buf), C.c_uint(1 << 20)))
            iters = 24
            e0 = torch.cuda.Event(enable_timing=True)
            ends = [torch.cuda.Event(enable_timing=True) for _ in range(R)]
            e0.record()
            for st in streams:
                st.wait_event(e0)
            for it in range(iters):
                for lane in range(R):
                    launch(lane, it)
            for lane in range(R):
                ends[lane].record(streams[lane])
            torch.cuda.synchronize()
            wall_ms = max(e0.elapsed_time(e) for e in ends)
            nbytes = rows * K * 2
            res[mode] = dict(us_per_launch_per_stream=round(1e3 * wall_ms / iters, 1), weight_GBps_delivered=round(iters * R * nbytes / (wall_ms * 1e-3) / 1e9, 1))
            if wg:
                lib.vispec_debug_wgclock_count.restype = C.c_longlong
                n = int(lib.vispec_debug_wgclock_count())
                L.check(lib.vispec_debug_wgclock_set(None, C.c_uint(0)))
                rec = buf[:n * REC.itemsize].cpu().numpy().view(REC)
                by_tag = {}
                for tag in np.unique(rec["tag"]):
                    r = rec[rec["tag"] == tag]
                    by_tag[int(tag)] = dict(xccs=sorted(set(r["xcc"].tolist())), wg_us_mean=round(float(((r["t1"] - r["t0"]) * 1e-2).mean()), 1))
                res[mode]["per_lane"] = list(by_tag.values())
            del streams
        out[name] = res
        print(name, json.dumps(res), flush=True)
        del Ws, X, Y, Rr
        torch.cuda.empty_cache()
    json.dump(out, open(os.path.join("gpurun_out", f"r06_cu_mask_probe_{R}lanes.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
