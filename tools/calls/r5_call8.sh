# round 5, call 8: the default line (no flags = 4 lanes x cohort 8) with every leg, the 13B-width cohort-8 float test, FETCH_SIZE pass and
# kernel stats of one cohort-8 lane
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -s -k "cohort_of_eight_verify and llava13b" 2>&1 | grep "COHORT-8\|passed\|failed" 
( time python bench.py > gpurun_out/r05_bench_line_noflags.json 2> gpurun_out/r05_bench_line_noflags.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line_noflags.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "mean_accept_length_tau") if k in d})
print("host", d.get("host")); print("aggregate", d.get("aggregate")); print("speedup_vs_ar", d.get("speedup_vs_ar"), d.get("ar_baseline"))
r = d.get("roofline", {}); print("roofline", {k: r.get(k) for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_us", "achieved_region", "frac_region", "largest_total_time_gemm", "deployed")})
print("spc", d.get("speedpy_comparable")); print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"}); print(d.get("extra_legs_error")); print(d["config"].get("prefill_gemms"))
PY
bash tools/pmc_traffic.sh r05_cohort8 --lanes 1 --cohort 8 2>&1 | tail -40
