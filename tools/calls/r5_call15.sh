# round 5, call 15: records of the head — the driver's command, the whole GPU suite, kernel stats, world-2 host-side check, RCCL at world 1
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line_driver_cmd.json 2> gpurun_out/r05_bench_line_driver_cmd.err ) 2>&1 | tail -3
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_line_driver_cmd.json')); print(d['value'], d['ms_per_step'], d['aggregate'], d.get('speedup_vs_ar'), d['host'], d.get('extra_legs_error'))"
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r05_gputests_head.txt; cat gpurun_out/r05_gputests_head.txt
bash tools/profile_bench.sh r05_4lanes_cohort8 --lanes 4 --cohort 8
python tools/stats_summary.py gpurun_out/kernel_stats_r05_4lanes_cohort8.csv 14
# two ranks x 2 lanes on ONE GPU (gloo) next to one rank x 4 lanes: the same 4 lanes of GPU work, twice the host processes
VISPEC_FORCE_DEVICE=0 VISPEC_DIST_BACKEND=gloo VISPEC_BENCH_RANKLOG=gpurun_out/world2_l2c8 timeout 1500 python bench.py --gpus 2 --steps 2 --warmup 1 --lanes 2 --cohort 8 --no-cpu-baseline --no-ar > gpurun_out/r05_world2_l2c8.json 2> gpurun_out/r05_world2_l2c8.err; echo rc $?
timeout 900 python bench.py --steps 2 --warmup 1 --lanes 4 --cohort 8 --no-cpu-baseline --no-ar > gpurun_out/r05_world1_l4c8.json 2> gpurun_out/r05_world1_l4c8.err
python - <<'PY'
import json
a = json.load(open("gpurun_out/r05_world2_l2c8.json")); b = json.load(open("gpurun_out/r05_world1_l4c8.json"))
print("world 2 x 2 lanes on one GPU:", a["value"], a["host"]); print("world 1 x 4 lanes:", b["value"], b["host"])
for r in (0, 1):
    print(json.load(open(f"gpurun_out/world2_l2c8/rank{r}.json")))
PY
timeout 600 python tools/rccl_check.py > gpurun_out/r05_rccl_check.txt 2>&1; tail -12 gpurun_out/r05_rccl_check.txt
