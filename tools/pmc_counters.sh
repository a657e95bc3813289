#!/bin/bash
# SQ / GRBM counters of the round's kernels, averaged per kernel name (own --pmc pass with --kernel-trace only).
#   ./tools/pmc_counters.sh <tag> "<counters>" [bench args...]      -> gpurun_out/pmc_<tag>.json
tag=$1; ctrs=$2; shift 2
args=${@:---lanes 1 --cohort 1}
export TMPDIR=/tmp
out=/tmp/pmcc_$tag; rm -rf $out
( cd "$GRAFT_REPO_ROOT" && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o q -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ar $args > /dev/null 2> gpurun_out/pmc_$tag.err )
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.json" <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not any(s in k for s in ("gemm_w32", "tree_attn", "splitk_reduce", "lstk_row")):
        continue
    a = acc[k.split("(")[0][:80]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
out = {k: dict(launches=max(n for n, _ in d.values()), **{c: v / n for c, (n, v) in sorted(d.items())}) for k, d in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
