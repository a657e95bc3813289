# round 4, call 27: the sibling lines at the round's head (one box)
mkdir -p gpurun_out
for m in llava13b qwen7b qwen7b-fp8 qwen7b-fp8a8; do timeout 900 python bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04z_bench_line_extra_$m.json 2> gpurun_out/r04z_bench_line_extra_$m.err; done
timeout 900 python bench.py --temperature 1.0 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04z_bench_line_extra_llava7b_T1.json 2>/dev/null
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04z_bench_line_extra_llava7b.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04z_bench_line_extra_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("extra_")[1][:-5], d["value"], "agg", d["aggregate"]["frac_of_8TBps"], "speedup_vs_ar", d.get("speedup_vs_ar"), "1req", d["speedpy_comparable"]["ms_per_round"], "tau", d["mean_accept_length_tau"])
    except Exception as e:
        print(f, "FAILED", e)
PY
