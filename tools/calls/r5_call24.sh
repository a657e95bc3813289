# round 5, call 24: the Qwen vision tower with its grid tables cached (no host synchronisation per image set) inside the timed region
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/r05_bench_line_extra_$tag.json 2> gpurun_out/r05_bench_line_extra_$tag.err; }
run qwen7b --model qwen7b
run qwen7b-fp8a8 --model qwen7b-fp8a8
run qwen7b-fp8 --model qwen7b-fp8
python - <<'PY'
import json
for t in ("qwen7b", "qwen7b-fp8", "qwen7b-fp8a8"):
    d = json.loads([l for l in open(f"gpurun_out/r05_bench_line_extra_{t}.json").read().splitlines() if l.startswith("{")][-1])
    v = d["speedpy_comparable"].get("with_vision_tower")
    print(t, d["value"], "tau", d["mean_accept_length_tau"], "vs AR", d.get("speedup_vs_ar"), "1req", d["speedpy_comparable"]["ms_per_round"], v.get("vision_s") if isinstance(v, dict) else v, d.get("extra_legs_error"))
PY
