# round 6, call 15: the other BASELINE configs at the round's head (one line each, vision front-end in the region)
mkdir -p gpurun_out
for m in llava13b qwen7b qwen7b-fp8 qwen7b-fp8-w8a16; do
  timeout 1200 python bench.py --model $m --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_line_extra_$m.json 2> gpurun_out/r06_bench_line_extra_$m.err
  python - "$m" <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r06_bench_line_extra_{m}.json"))
    print(m, d["value"], d["dtype"], "tau", d["mean_accept_length_tau"], "agg", d["aggregate"]["frac_of_8TBps"], "speedup_vs_ar", d.get("speedup_vs_ar"), "1req ms/round", d["speedpy_comparable"]["ms_per_round"], d["config"]["workload"][:60], flush=True)
except Exception as e:
    print(m, "FAILED", e, flush=True)
PY
done
