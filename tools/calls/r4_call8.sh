# round 4, call 8: the driver's three commands on a fresh box at the round's HEAD
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r04_gputests_head.txt 2>&1; tail -6 gpurun_out/r04_gputests_head.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r04_bench_line_noflags.json 2> gpurun_out/r04_bench_line_noflags.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_line_noflags.json"))
print("no flags:", d["value"], d["n_gpus"], d["steps"], d["warmup"], d["ms_per_step"], d["aggregate"]["frac_of_8TBps"], d["roofline"]["frac"], d["roofline"]["deployed"]["frac"], d["speedup_vs_ar"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("config0_end_to_end", {}).get("tokens_per_s_end_to_end"))
PY
