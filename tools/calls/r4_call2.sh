# round 4, call 2: the eight-row-block wide kernel — bit identity, kernel alone, under stream concurrency, and the bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wide_gpu.py -q -x -k "wide_gemm" > gpurun_out/r04b_pytest_wide.txt 2>&1; tail -5 gpurun_out/r04b_pytest_wide.txt
timeout 1200 python -m pytest tests/test_full_size_gpu.py -q -x -k "wide_cohort" > gpurun_out/r04b_pytest_full.txt 2>&1; tail -5 gpurun_out/r04b_pytest_full.txt
python tools/wide_bench.py 0 5 > gpurun_out/r04b_wide_bench_llava.txt 2>&1; cat gpurun_out/r04b_wide_bench_llava.txt
SHAPES=qwen7b python tools/wide_bench.py 0 5 > gpurun_out/r04b_wide_bench_qwen.txt 2>&1; cat gpurun_out/r04b_wide_bench_qwen.txt
M=120 UNC=0 python tools/gemm_mt2_concurrency.py 2>&1 | grep wide > gpurun_out/r04b_conc_rb4.txt; cat gpurun_out/r04b_conc_rb4.txt
M=120 UNC=5 python tools/gemm_mt2_concurrency.py 2>&1 | grep wide > gpurun_out/r04b_conc_rb8.txt; cat gpurun_out/r04b_conc_rb8.txt
bash tools/sweep.sh > gpurun_out/r04b_sweep.txt 2>&1 <<'S'
llava_rb4||
llava_rb8|| --wide-row-blocks 8
qwen_rb8|| --model qwen7b --wide-row-blocks 8
qwenfp8_rb8|| --model qwen7b-fp8 --wide-row-blocks 8
llava_rb4b||
llava_rb8b|| --wide-row-blocks 8
qwen_rb4|| --model qwen7b
qwenfp8_rb4|| --model qwen7b-fp8
llava_rb8_l3|| --wide-row-blocks 8 --lanes 3
llava_rb8_l5|| --wide-row-blocks 8 --lanes 5
llava_rb8_l6|| --wide-row-blocks 8 --lanes 6
S
cat gpurun_out/r04b_sweep.txt
