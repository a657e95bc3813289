# round 4, call 15: the W8A8 full-size tests with their bars (chaotic element-wise differences: triangulation + mean), lanes sweep of the W8A8
# line, HBM traffic (FETCH_SIZE) of its kernels, the line with the CPU / AR legs for profiles/
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_full_size_gpu.py -q -k "fp8a8" -s > gpurun_out/r04n_pytest_fullsize_fp8a8.txt 2>&1; grep -a "full-width\|passed\|failed\|Error" gpurun_out/r04n_pytest_fullsize_fp8a8.txt | cut -c1-500 | tail -12
bash tools/sweep.sh > gpurun_out/r04n_sweep.txt 2>&1 <<'S'
n_a8_l3||--model qwen7b-fp8a8 --lanes 3
n_a8_l4||--model qwen7b-fp8a8 --lanes 4
n_a8_l5||--model qwen7b-fp8a8 --lanes 5
n_a8_l6||--model qwen7b-fp8a8 --lanes 6
n_fp8_l5||--model qwen7b-fp8 --lanes 5
S
cat gpurun_out/r04n_sweep.txt
bash tools/pmc_traffic.sh r04_fetch_qwen7bfp8a8 --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 84 > gpurun_out/r04_pmc_fetch_qwenfp8a8.log 2>&1; tail -3 gpurun_out/r04_pmc_fetch_qwenfp8a8.log | cut -c1-300
timeout 900 python bench.py --model qwen7b-fp8a8 > gpurun_out/r04n_bench_fp8a8.json 2> gpurun_out/r04n_bench_fp8a8.err; cut -c1-300 gpurun_out/r04n_bench_fp8a8.json
timeout 900 python bench.py --model qwen7b-fp8 > gpurun_out/r04n_bench_fp8.json 2> gpurun_out/r04n_bench_fp8.err; cut -c1-300 gpurun_out/r04n_bench_fp8.json
