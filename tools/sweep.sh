#!/bin/bash
# bench.py over a list of "tag|env assignments|bench args" lines (stdin), one short line per run.  GPU box only.
while IFS='|' read -r tag envs args; do
  [ -z "$tag" ] && continue
  timeout 900 env $envs python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ar $args > gpurun_out/sw_$tag.json 2> gpurun_out/sw_$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/sw_{tag}.json"))
    print(tag, d["value"], "tau", d["mean_accept_length_tau"], "agg_frac", d["aggregate"]["frac_of_8TBps"], "1req ms/round", d["speedpy_comparable"]["ms_per_round"], flush=True)
except Exception as e:
    print(tag, "FAILED", e, flush=True)
PY
done
