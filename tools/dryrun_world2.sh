#!/bin/bash
# The world-size-2 control flow of bench.py END TO END on ONE GPU (no 8-GPU node was available to any round so far): two ranks, both on
# device 0 (VISPEC_FORCE_DEVICE=0; two 7B replicas + their KV caches fit in 288 GB), rendezvous over gloo (RCCL refuses two ranks on one
# GPU: "Duplicate GPU detected") — self_launch, init_process_group, replicate_weights (rank 1 starts from DIFFERENT weights) in both
# VISPEC_REPLICATE modes, the barrier-bracketed timed region, the all_reduce of the statistics, the rank-0 JSON line with n_gpus = 2.
# Every rank writes what it did to gpurun_out/world2_<mode>/rank<r>.json.  Usage (GPU box):  bash tools/dryrun_world2.sh [extra bench args]
# This is a CONTROL-FLOW check, not a measurement: both ranks share one GPU, the tokens/s of these lines mean nothing.
mkdir -p gpurun_out
for mode in broadcast scatter; do
  VISPEC_FORCE_DEVICE=0 VISPEC_DIST_BACKEND=gloo VISPEC_REPLICATE=$mode VISPEC_BENCH_RANKLOG=gpurun_out/world2_$mode \
    timeout 1500 python bench.py --gpus 2 --steps 1 --warmup 1 --lanes 1 --cohort 2 --no-cpu-baseline --no-ar --max-new-tokens 64 "$@" \
    > gpurun_out/world2_$mode.json 2> gpurun_out/world2_$mode.err
  echo "mode $mode: rc $?"; tail -c 600 gpurun_out/world2_$mode.json; grep -h "replicated" gpurun_out/world2_$mode.err
done
