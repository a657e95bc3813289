# round 5, call 14: fused attention merge with write-through (sc1) partial stores instead of a release fence per workgroup
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_cohort_gpu.py -x -q -m gpu -k "attention or cohort" 2>&1 | tail -3
bash tools/sweep.sh > gpurun_out/r05k_sweep.txt 2>&1 <<'S'
k_fused_sc1||
k_sep|VISPEC_ATT_FUSED_MERGE=0|
k_fused_sc1_b||
k_sep_b|VISPEC_ATT_FUSED_MERGE=0|
S
cat gpurun_out/r05k_sweep.txt
