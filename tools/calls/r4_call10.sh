# round 4, call 10: W8A8 tests (all) + per-kernel comparison fp8 vs fp8a8 on one cohort lane
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fp8a8_gpu.py -q > gpurun_out/r04i_pytest_fp8a8.txt 2>&1; tail -15 gpurun_out/r04i_pytest_fp8a8.txt | cut -c1-250
bash tools/profile_bench.sh r04i_fp8_1lane --model qwen7b-fp8 --lanes 1 --cohort 4 --wide-row-blocks 84
bash tools/profile_bench.sh r04i_fp8a8_1lane --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 84
python tools/stats_summary.py gpurun_out/kernel_stats_r04i_fp8_1lane.csv 10; python tools/stats_summary.py gpurun_out/kernel_stats_r04i_fp8a8_1lane.csv 12
