# round 5, call 17: the lines of the head — default (every leg) and the siblings for the other BASELINE configs
mkdir -p gpurun_out
python bench.py > gpurun_out/r05_bench_line_noflags.json 2> gpurun_out/r05_bench_line_noflags.err
run() { tag=$1; shift; timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/r05_bench_line_extra_$tag.json 2> gpurun_out/r05_bench_line_extra_$tag.err; }
run llava13b --model llava13b
run qwen7b --model qwen7b
run qwen7b-fp8 --model qwen7b-fp8
run qwen7b-fp8a8 --model qwen7b-fp8a8
run llava7b_T1 --temperature 1.0
run llava7b_img2928 --n-img 2928
run llava13b_requests64 --model llava13b --requests 64 --steps 1
run llava7b_novision --no-vision-in-loop
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05_bench_line_*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("line_")[1][:-5], d["value"], "tau", d["mean_accept_length_tau"], "agg", d["aggregate"]["frac_of_8TBps"], "vs AR", d.get("speedup_vs_ar"),
              "1req", d["speedpy_comparable"]["ms_per_round"], d["speedpy_comparable"].get("speedup_vs_ar"), "spec==AR", d.get("spec_equals_ar_prefix"), d.get("extra_legs_error"))
    except Exception as e:
        print(f, "FAILED", e)
PY
