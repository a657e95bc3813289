# round 4, call 24: the W8A8 line with all legs at the head (prefill on the fp8 MFMA), its kernel stats on four lanes
mkdir -p gpurun_out
timeout 900 python bench.py --model qwen7b-fp8a8 > gpurun_out/r04w_bench_fp8a8.json 2> gpurun_out/r04w_bench_fp8a8.err; cut -c1-300 gpurun_out/r04w_bench_fp8a8.json
bash tools/profile_bench.sh r04w_fp8a8_4lanes --model qwen7b-fp8a8
python tools/stats_summary.py gpurun_out/kernel_stats_r04w_fp8a8_4lanes.csv 14
