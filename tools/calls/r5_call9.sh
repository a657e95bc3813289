# round 5, call 9: split-K workgroups of a split on the same XCDs (csrc/gemm_c8.h): tests, FETCH_SIZE, same-box A/B against the plain mapping
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "fp64 or depend" 2>&1 | tail -2
bash tools/pmc_traffic.sh r05_cohort8_xcd --lanes 1 --cohort 8 2>&1 | grep -A2 "c8_kernel<3\|c8_kernel<2"
bash tools/sweep.sh > gpurun_out/r05f_sweep.txt 2>&1 <<'S'
f_swz||
f_noswz|VISPEC_LIB_VARIANT=noswz|
f_swz_b||
f_noswz_b|VISPEC_LIB_VARIANT=noswz|
S
cat gpurun_out/r05f_sweep.txt
SHAPES=llava7b timeout 600 python tools/c8_bench.py 2>&1 | grep -v "^check" | tail -5
