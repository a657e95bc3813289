# round 5, call 1: the cohort-8 GEMM prototype (csrc/gemm_c8.h): correctness against fp64, alone-times next to wide8; regression run of the
# kernel / wide suites after the 4 -> 8 generalisation of the workspaces and argument packs
mkdir -p gpurun_out
timeout 900 python tools/c8_bench.py > gpurun_out/r05_c8_bench_llava7b.txt 2>&1; tail -32 gpurun_out/r05_c8_bench_llava7b.txt
SHAPES=qwen7b timeout 600 python tools/c8_bench.py 2>&1 | grep -v "^check" > gpurun_out/r05_c8_bench_qwen7b.txt; cat gpurun_out/r05_c8_bench_qwen7b.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_wide_gpu.py tests/test_cohort_gpu.py -x -q -m gpu 2>&1 | tail -5
