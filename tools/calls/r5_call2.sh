# round 5, call 2: cohort-8 tests (GEMM + whole loops), then the line at cohort 8 x {2,3,4} lanes next to the default 4 x 4 on the same box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_c8_gpu.py -x -q -m gpu 2>&1 | tail -15
bash tools/sweep.sh > gpurun_out/r05b_sweep.txt 2>&1 <<'S'
b_l4c4||--lanes 4 --cohort 4
b_l2c8||--lanes 2 --cohort 8
b_l3c8||--lanes 3 --cohort 8
b_l4c8||--lanes 4 --cohort 8
b_l4c4_b||--lanes 4 --cohort 4
S
cat gpurun_out/r05b_sweep.txt; tail -3 gpurun_out/sw_b_l4c8.err
