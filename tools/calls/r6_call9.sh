# round 6, call 9: split factor of the cohort-8 split-K GEMMs (o_proj, down) on the line: 4 (the single-request policy's) vs 1 vs 2, same box
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r06b_sweep.txt 2>&1 <<'S'
b_s4||--no-vision-in-loop
b_s1|VISPEC_C8_SPLIT=1|--no-vision-in-loop
b_s2|VISPEC_C8_SPLIT=2|--no-vision-in-loop
b_s4_b||--no-vision-in-loop
b_s1_b|VISPEC_C8_SPLIT=1|--no-vision-in-loop
S
cat gpurun_out/r06b_sweep.txt
