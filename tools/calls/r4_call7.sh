# round 4, call 7: the round's evidence — fixture tests, barrier probe, rocprofv3 kernel stats (timed shapes / one-lane shapes / four lanes /
# one request / Qwen bf16 + fp8), FETCH_SIZE pass, the driver's command, the sibling lines, the world-size-2 dry run
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fixtures_gpu.py -q -s > gpurun_out/r04g_pytest_fixtures.txt 2>&1; tail -12 gpurun_out/r04g_pytest_fixtures.txt | cut -c1-250
hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_barrier_probe.hip -o /tmp/xb && timeout 300 /tmp/xb > gpurun_out/r04_xcd_barrier_probe.txt 2>&1; cat gpurun_out/r04_xcd_barrier_probe.txt
bash tools/profile_bench.sh r04_1lane_cohort4_rb84 --lanes 1 --cohort 4 --wide-row-blocks 84
bash tools/profile_bench.sh r04_1lane_cohort4_rb0 --lanes 1 --cohort 4
bash tools/profile_bench.sh r04_4lanes_cohort4 --lanes 4 --cohort 4
bash tools/profile_bench.sh r04_1lane_cohort1 --lanes 1 --cohort 1
bash tools/profile_bench.sh r04_qwen7b_1lane_cohort4_rb84 --model qwen7b --lanes 1 --cohort 4 --wide-row-blocks 84
bash tools/profile_bench.sh r04_qwen7b_1lane_cohort4_rb0 --model qwen7b --lanes 1 --cohort 4
bash tools/profile_bench.sh r04_qwen7bfp8_1lane_cohort4_rb84 --model qwen7b-fp8 --lanes 1 --cohort 4 --wide-row-blocks 84
bash tools/pmc_traffic.sh r04_fetch --lanes 1 --cohort 4 --wide-row-blocks 84 > gpurun_out/r04_pmc_fetch.log 2>&1; tail -3 gpurun_out/r04_pmc_fetch.log
bash tools/pmc_traffic.sh r04_fetch_qwen7b --model qwen7b --lanes 1 --cohort 4 --wide-row-blocks 84 > gpurun_out/r04_pmc_fetch_qwen.log 2>&1
cp gpurun_out/pmc_r04_fetch.json profiles/r04_pmc_fetch_size.json 2>/dev/null
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line_default.json 2> gpurun_out/r04_bench_line_default.err; tail -c 1500 gpurun_out/r04_bench_line_default.json
for m in llava13b qwen7b qwen7b-fp8; do timeout 900 python bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_line_extra_$m.json 2> gpurun_out/r04_bench_line_extra_$m.err; done
timeout 900 python bench.py --model llava13b --requests 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_line_extra_llava13b_requests64.json 2>/dev/null
timeout 900 python bench.py --temperature 1.0 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_line_extra_llava7b_T1.json 2>/dev/null
timeout 900 python bench.py --n-img 2928 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_line_extra_llava7b_img2928.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_bench_line_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], "agg", d["aggregate"]["frac_of_8TBps"], "roof", d["roofline"]["frac"], d["roofline"].get("deployed"), "spd", d.get("speedup_vs_ar"), "1req", d["speedpy_comparable"]["ms_per_round"], d["speedpy_comparable"].get("speedup_vs_ar"))
    except Exception as e:
        print(f, "FAILED", e)
PY
bash tools/dryrun_world2.sh
for n in 4; do for rb in 84 4 0; do python - <<PY
import subprocess, sys
PY
done; done
