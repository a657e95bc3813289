#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; the stats summary lands in gpurun_out/kernel_stats_<tag>.csv
# (copy into profiles/ to keep it).  Usage on the GPU box:  ./tools/profile_bench.sh r02 [extra bench args, default: --lanes 1]
tag=${1:-r02}
shift
args=${@:---lanes 1 --cohort 1}
export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
( cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ar $args > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err )
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$tag.csv"
