#!/bin/bash
# HBM read traffic of the dominant kernel from the PMC counters (own pass: --pmc with --kernel-trace only, no --stats/--sys-trace).
tag=${1:-r02}
shift
args=${@:---lanes 1 --cohort 1}
export TMPDIR=/tmp
out=/tmp/pmc_$tag
rm -rf $out
( cd "$GRAFT_REPO_ROOT" && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ar $args > /dev/null 2> gpurun_out/pmc_$tag.err )
find $out -name "*.csv" | head
f=$(find $out -name "*counter_collection.csv" | head -1)
head -3 "$f" | cut -c1-400
python - "$f" "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.json" <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != "FETCH_SIZE":
        continue
    k = r["Kernel_Name"].split("(")[0][:80]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
out = {k: {"launches": n, "fetch_size_kb_per_launch": v / n} for k, (n, v) in acc.items() if "gemm_w32" in k or "tree_attn" in k}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
