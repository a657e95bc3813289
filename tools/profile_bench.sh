#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; the stats summary lands in gpurun_out/kernel_stats_<tag>.csv
# (copy into profiles/ to keep it).  Usage on the GPU box:  ./tools/profile_bench.sh r01
tag=${1:-r01}
export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
( cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ar --lanes 1 > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err )
find $out -name "*stats*" | head
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$tag.csv"
head -45 "$f" | cut -c1-180
cat "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.json" | cut -c1-600
