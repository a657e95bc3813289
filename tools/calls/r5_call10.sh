# round 5, call 10: stage fixtures g1 / g2 / g3 / g5 / g10 through the HIP path; the default line with the vision front-end inside the timed region
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stage_fixtures_gpu.py -q -m gpu 2>&1 | tail -30
( time python bench.py --no-cpu-baseline > gpurun_out/r05_bench_vision_in_loop.json 2> gpurun_out/r05_bench_vision_in_loop.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_vision_in_loop.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "mean_accept_length_tau") if k in d}); print(d["config"]["vision_front_end"]); print(d["speedpy_comparable"].get("with_vision_tower")); print(d.get("extra_legs_error"))
PY
tail -5 gpurun_out/r05_bench_vision_in_loop.err
bash tools/sweep.sh > gpurun_out/r05g_sweep.txt 2>&1 <<'S'
g_vis||
g_novis||--no-vision-in-loop
g_qwen_vis||--model qwen7b
g_13b_vis||--model llava13b
S
cat gpurun_out/r05g_sweep.txt
