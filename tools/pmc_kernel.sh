#!/bin/bash
# SQ counters of one command, averaged per kernel name (own --pmc pass with --kernel-trace only).
#   ./tools/pmc_kernel.sh <tag> "<counters>" <kernel-substring> -- <command...>
tag=$1; ctrs=$2; filt=$3; shift 4
export TMPDIR=/tmp
out=/tmp/pmcq_$tag; rm -rf $out
( cd "$GRAFT_REPO_ROOT" && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o q -- "$@" > /dev/null 2> gpurun_out/pmcq_$tag.err )
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" "$filt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    a = acc[k[:70]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.1f}  (x{n})")
PY
