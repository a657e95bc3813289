# round 6, call 10: lanes per GPU at cohort 8 at the round's head (round 5 measured 5 lanes on the first cohort-8 form only), same box;
# + the full-size wide-tree cohort test
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r06c_sweep.txt 2>&1 <<'S'
c_l4||--no-vision-in-loop
c_l5||--no-vision-in-loop --lanes 5
c_l6||--no-vision-in-loop --lanes 6
c_l3||--no-vision-in-loop --lanes 3
c_l4_b||--no-vision-in-loop
c_l5_q16|GPU_MAX_HW_QUEUES=16|--no-vision-in-loop --lanes 5
S
cat gpurun_out/r06c_sweep.txt
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -s -k "wide_tree" --tb=short 2>&1 | tail -8
