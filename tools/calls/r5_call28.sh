# round 5, call 28: 768 attention keys per workgroup for cohorts of 5..8 (512 stays for single requests and cohorts of <= 4): tests, A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_c8_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "whole_loops or eight_slots or tree_shapes or cohort_of_eight_at_full_size or default_bench or cohort_of_eight_verify" 2>&1 | tail -3
bash tools/sweep.sh > gpurun_out/r05p_sweep.txt 2>&1 <<'S'
p_768||--no-vision-in-loop
p_512|VISPEC_ATT_KPW_C8=512|--no-vision-in-loop
p_768_b||--no-vision-in-loop
p_512_b|VISPEC_ATT_KPW_C8=512|--no-vision-in-loop
p_1024|VISPEC_ATT_KPW_C8=1024|--no-vision-in-loop
S
cat gpurun_out/r05p_sweep.txt
