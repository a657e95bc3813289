# round 5, call 18: the Qwen lines from image features (the HF vision tower outside the region, as in rounds 1-4) next to call 17's; sampling tests after
# the scratch fix of verify_accept_sample
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fixtures_gpu.py tests/test_loop_gpu.py tests/test_cohort_gpu.py -q -m gpu -k "sampl or g7 or T1 or temperature" 2>&1 | tail -3
run() { tag=$1; shift; timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/r05_bench_line_extra_$tag.json 2> gpurun_out/r05_bench_line_extra_$tag.err; }
run qwen7b_novision --model qwen7b --no-vision-in-loop
run qwen7b-fp8_novision --model qwen7b-fp8 --no-vision-in-loop
run qwen7b-fp8a8_novision --model qwen7b-fp8a8 --no-vision-in-loop
run llava13b_novision --model llava13b --no-vision-in-loop
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05_bench_line_extra_*novision.json")):
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(f.split("extra_")[1][:-5], d["value"], "tau", d["mean_accept_length_tau"], "agg", d["aggregate"]["frac_of_8TBps"], "vs AR", d.get("speedup_vs_ar"), "1req", d["speedpy_comparable"]["ms_per_round"], d.get("extra_legs_error"))
PY
