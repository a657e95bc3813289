# round 5, call 13: attention key-split merge by the last-arriving workgroup of the partial launch: tests, then the line and the one-request round A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_cohort_gpu.py tests/test_c8_gpu.py tests/test_draft_round_gpu.py -x -q -m gpu 2>&1 | tail -4
bash tools/sweep.sh > gpurun_out/r05j_sweep.txt 2>&1 <<'S'
j_fused||
j_sep|VISPEC_ATT_FUSED_MERGE=0|
j_fused_b||
j_sep_b|VISPEC_ATT_FUSED_MERGE=0|
S
cat gpurun_out/r05j_sweep.txt
COHORT=8 timeout 900 python tools/fp8_k_sweep.py > gpurun_out/r05_c8_k_sweep.txt 2>&1; tail -14 gpurun_out/r05_c8_k_sweep.txt
