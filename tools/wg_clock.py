"""Per-workgroup clocks of the timed configuration (round 6; csrc/wgclock.h).  Needs the DIAGNOSTIC build of the library:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVISPEC_WG_CLOCK vispec_amd/csrc/vispec_hip.hip -o vispec_amd/libvispec_hip_wgclk.so
    VISPEC_LIB_VARIANT=wgclk python tools/wg_clock.py --lanes 4 --cohort 8 [--max-new-tokens 160] [--no-vision] out.json

Runs one warm-up and one recorded step of bench.py's lane loop (continuous batching over the lanes' cohorts, the vision front-end inside the
requests unless --no-vision), then reduces the records:

  per kernel (c8 GEMM by epilogue, tree attention partial / reduce, prefill attention):
    wg_us          run time of ONE workgroup (its first to its last instruction), mean / p50 / p90
    span_us        a launch from its first workgroup's start to its last workgroup's end
    stagger_us     last start - first start within a launch (0 = every workgroup found a free CU at once)
    cu_us          sum of wg_us per launch x the share of a CU a workgroup holds (1 for the c8 GEMM: 128 KiB of LDS; 1/2 for attention)
  and for the recorded interval as a whole: CU-time consumed per kernel class / (256 CUs x wall time).

`--lanes 1` is the "alone" reference of the same table."""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from vispec_amd import lib as L  # noqa: E402
from vispec_amd.evaluation.bench_launch import run_lanes  # noqa: E402
from vispec_amd.evaluation.bench_vision import InLoopFrontEnd, build_front_end  # noqa: E402
from vispec_amd.model.spec_model_ours import specgenerate_stream  # noqa: E402

KID = {**{40 + e: f"skinny<{e}>" for e in range(5)}, **{48 + e: f"slab<{e}>" for e in range(5)},
       10: "c8<0> none", 11: "c8<1> residual", 12: "c8<2> gate|up", 13: "c8<3> split-K partial", 14: "c8<4> q|k|v", 20: "attn partial (eager)",
       21: "attn partial", 22: "attn reduce", 30: "prefill attention"}
CU_SHARE = {**{40 + e: 0.25 for e in range(5)}, **{48 + e: 0.25 for e in range(5)},  # (assumed: 256-thread workgroups, <= 4 per CU)
            10: 1.0, 11: 1.0, 12: 1.0, 13: 1.0, 14: 1.0, 20: 0.5, 21: 0.5, 22: 0.125, 30: 0.5}
REC = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("kid", "<u4"), ("tag", "<u4"), ("blk", "<u4"), ("nblk", "<u4"), ("hw", "<u4"), ("xcc", "<u4"),
                ("p0", "<u4"), ("p1", "<u4")])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--cohort", type=int, default=8)
    ap.add_argument("--max-new-tokens", type=int, default=96)
    ap.add_argument("--no-vision", action="store_true")
    ap.add_argument("--model", default="llava7b")
    ap.add_argument("--cap", type=int, default=48 << 20)
    a = ap.parse_args()
    lib = L.load()
    if not hasattr(lib, "vispec_debug_wgclock_set"):
        raise SystemExit("not the diagnostic build: compile with -DVISPEC_WG_CLOCK into vispec_amd/libvispec_hip_wgclk.so and set VISPEC_LIB_VARIANT=wgclk")
    lib.vispec_debug_wgclock_count.restype = C.c_longlong
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    bench.MODEL, bench.MAX_NEW = a.model, a.max_new_tokens
    R, CO = a.lanes, a.cohort
    sms, tcfg, _ = bench.build_models(dev, 0, 0, 1, R, CO)
    if not a.no_vision:
        n_img = {"qwen7b": 1024, "qwen7b-hires": 1564, "qwen7b-fp8": 1564, "qwen7b-fp8a8": 1564}.get(a.model, bench.N_IMG)
        bench.FRONT_END = InLoopFrontEnd(*build_front_end(a.model, tcfg, dev, n_img))
        for grp in sms:
            for m in grp:
                m.base_model.vision = bench.FRONT_END
    streams = [torch.cuda.Stream(dev) for _ in range(R)]
    per_lane = 2 * CO  # requests per lane per step: the slots refill once
    reqs = {i: bench.make_request(tcfg, i, dev) for i in range(2 * R * per_lane)}
    torch.cuda.synchronize()

    def lane_fn(lane, step):
        def f():
            torch.cuda.set_device(dev)
            mine = [step * R * per_lane + lane * per_lane + i for i in range(per_lane)]
            with torch.cuda.stream(streams[lane]):
                outs = specgenerate_stream(sms[lane], [reqs[i] for i in mine], max_new_tokens=a.max_new_tokens, seeds=mine)
                streams[lane].synchronize()
            return sum(int(o[1]) for o in outs), sum(o[2] + 1 for o in outs)
        return f

    run_lanes([lane_fn(l, 0) for l in range(R)])  # warm-up: graphs captured, TunableOp table loaded
    torch.cuda.synchronize()
    buf = torch.zeros(a.cap * REC.itemsize, dtype=torch.uint8, device=dev)
    L.check(lib.vispec_debug_wgclock_set(C.c_void_p(buf.data_ptr()), C.c_uint(a.cap)))
    t0 = time.time()
    res = run_lanes([lane_fn(l, 1) for l in range(R)])
    torch.cuda.synchronize()
    wall = time.time() - t0
    n = int(lib.vispec_debug_wgclock_count())
    L.check(lib.vispec_debug_wgclock_set(None, C.c_uint(0)))
    rec = buf[:min(n, a.cap) * REC.itemsize].cpu().numpy().view(REC)
    tokens, rounds = sum(r[0] for r in res), sum(r[1] for r in res)
    out = dict(lanes=R, cohort=CO, model=a.model, max_new_tokens=a.max_new_tokens, vision_in_loop=not a.no_vision, wall_s=round(wall, 3), tokens=tokens,
               request_rounds=rounds, tokens_per_s=round(tokens / wall, 1), records=n, dropped=max(0, n - a.cap), kernels={})
    tick = 1e-2  # s_memrealtime: 100 MHz -> us per tick
    span_all = (rec["t1"].max() - rec["t0"].min()) * tick
    cu_total = 0.0
    for kid in sorted(set(rec["kid"].tolist())):
        r = rec[rec["kid"] == kid]
        wg = (r["t1"] - r["t0"]).astype(np.float64) * tick
        spans, stag, cus, nb_list = [], [], [], []
        for tag in np.unique(r["tag"]):  # one stream per tag: its launches of this kernel do not overlap in time
            rt = r[r["tag"] == tag]
            rt = rt[np.argsort(rt["t0"], kind="stable")]
            t0_, t1_ = rt["t0"].astype(np.int64), rt["t1"].astype(np.int64)
            reach = np.maximum.accumulate(t1_)
            first = np.concatenate([[True], t0_[1:] > reach[:-1]])  # a workgroup that starts after everything before it has ended opens a launch
            starts = np.flatnonzero(first)
            dur = (t1_ - t0_).astype(np.float64)
            g_t0min, g_t0max = t0_[starts], t0_[np.concatenate([starts[1:], [len(t0_)]]) - 1]  # (sorted by start)
            g_t1max = np.maximum.reduceat(t1_, starts)
            g_sum = np.add.reduceat(dur, starts)
            g_n = np.diff(np.concatenate([starts, [len(t0_)]]))
            spans += ((g_t1max - g_t0min) * tick).tolist()
            stag += ((g_t0max - g_t0min) * tick).tolist()
            cus += (g_sum * tick * CU_SHARE.get(kid, 1.0)).tolist()
            nb_list += g_n.tolist()
        cu_sum = float(wg.sum() * CU_SHARE.get(kid, 1.0))
        cu_total += cu_sum
        pct = lambda v, p: round(float(np.percentile(v, p)), 2)
        out["kernels"][KID.get(kid, str(kid))] = dict(
            workgroups=int(len(r)), launches=len(spans), wgs_per_launch=round(float(np.mean(nb_list)), 1),
            wg_us=dict(mean=round(float(wg.mean()), 2), p50=pct(wg, 50), p90=pct(wg, 90), p99=pct(wg, 99)),
            span_us=dict(mean=round(float(np.mean(spans)), 2), p50=pct(spans, 50), p90=pct(spans, 90)),
            stagger_us=dict(mean=round(float(np.mean(stag)), 2), p50=pct(stag, 50), p90=pct(stag, 90)),
            core_clock_MHz=(lambda m: dict(mean=round(float(m.mean()), 0), p10=pct(m, 10), p90=pct(m, 90)))(
                r["p0"][r["p0"] > 0].astype(np.float64) / np.maximum(1.0, (r["t1"] - r["t0"])[r["p0"] > 0].astype(np.float64)) * 100.0) if (r["p0"] > 0).any() else None,
            cu_us_per_launch=round(float(np.mean(cus)), 1), cu_time_share_of_chip=round(cu_sum / (256.0 * span_all), 4),
            distinct_cus=int(len(np.unique((r["xcc"].astype(np.int64) << 8) | ((r["hw"] >> 8) & 0xff)))))
    out["recorded_interval_us"] = round(float(span_all), 1)
    out["instrumented_cu_time_share_of_chip"] = round(cu_total / (256.0 * span_all), 4)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
