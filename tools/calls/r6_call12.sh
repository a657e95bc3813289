# round 6, call 12: the core clock the workgroups run at (s_memtime cycles / s_memrealtime) — one cohort-8 lane alone vs four lanes
mkdir -p gpurun_out
export VISPEC_LIB_VARIANT=wgclk
timeout 900 python tools/wg_clock.py --lanes 1 --cohort 8 --no-vision --max-new-tokens 128 gpurun_out/r06_wgclock_clock_1lane.json > gpurun_out/r06_wgclock_clock_1lane.log 2>&1; tail -2 gpurun_out/r06_wgclock_clock_1lane.log
timeout 900 python tools/wg_clock.py --lanes 4 --cohort 8 --no-vision --max-new-tokens 128 gpurun_out/r06_wgclock_clock_4lanes.json > gpurun_out/r06_wgclock_clock_4lanes.log 2>&1; tail -2 gpurun_out/r06_wgclock_clock_4lanes.log
python - <<'PY'
import json
for f in ("1lane","4lanes"):
    d=json.load(open(f"gpurun_out/r06_wgclock_clock_{f}.json"))
    print(f, d["tokens_per_s"])
    for k,v in d["kernels"].items():
        print(f"   {k:26s} wg_us {v['wg_us']['mean']:7.1f}  clock {v.get('core_clock_MHz')}")
PY
