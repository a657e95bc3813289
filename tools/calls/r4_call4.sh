# round 4, call 4: Wide-8 with its LDS fragment reads issued ahead of the MFMAs — bit identity, kernel alone, bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wide_gpu.py -q -x -k "wide_gemm" > gpurun_out/r04d_pytest_wide.txt 2>&1; tail -3 gpurun_out/r04d_pytest_wide.txt
python tools/wide_bench.py 0 5 > gpurun_out/r04d_wide_bench_llava.txt 2>&1; cat gpurun_out/r04d_wide_bench_llava.txt
SHAPES=qwen7b python tools/wide_bench.py 0 5 > gpurun_out/r04d_wide_bench_qwen.txt 2>&1; cat gpurun_out/r04d_wide_bench_qwen.txt
bash tools/profile_bench.sh r04d_qwen7bfp8_1lane_rb8 --model qwen7b-fp8 --lanes 1 --cohort 4 --wide-row-blocks 8
python tools/stats_summary.py gpurun_out/kernel_stats_r04d_qwen7bfp8_1lane_rb8.csv 6
bash tools/sweep.sh > gpurun_out/r04d_sweep.txt 2>&1 <<'S'
llava_rb8||--wide-row-blocks 8
llava_rb4||
qwenfp8_rb8||--model qwen7b-fp8 --wide-row-blocks 8
qwenfp8_rb4||--model qwen7b-fp8
qwen_rb8||--model qwen7b --wide-row-blocks 8
qwen_rb4||--model qwen7b
llava13b_rb8||--model llava13b --wide-row-blocks 8
llava13b_rb4||--model llava13b
S
cat gpurun_out/r04d_sweep.txt
