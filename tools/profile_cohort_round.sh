#!/bin/bash
# rocprofv3 kernel stats of ONE cohort's decode rounds (tools/cohort_round_bench.py): gpurun_out/kernel_stats_<tag>.csv
tag=${1:-r03_cohort4}
n=${2:-4}
export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
( cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cr -- python tools/cohort_round_bench.py $n 256 > gpurun_out/prof_$tag.txt 2> gpurun_out/prof_$tag.err )
cp "$(find $out -name "*kernel_stats.csv" | head -1)" "$GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$tag.csv"
cat "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.txt"
