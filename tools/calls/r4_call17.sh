# round 4, call 17: where does the fifth lane's time go?  per-kernel totals of the default line at 4 and at 5 lanes on one box
mkdir -p gpurun_out
bash tools/profile_bench.sh r04p_l4 --lanes 4
bash tools/profile_bench.sh r04p_l5 --lanes 5
python tools/stats_summary.py gpurun_out/kernel_stats_r04p_l4.csv 14
python tools/stats_summary.py gpurun_out/kernel_stats_r04p_l5.csv 14
python - <<'PY'
import json
for t in ("l4", "l5"):
    d = json.load(open(f"gpurun_out/prof_r04p_{t}.json"))
    print(t, d["value"], d["ms_per_step"], d["config"].get("parallelism"))
PY
