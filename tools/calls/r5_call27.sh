# round 5, call 27: env-only knobs at 4 lanes x cohort 8, same box: attention keys per workgroup, reduce threads, hardware queues
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r05o_sweep.txt 2>&1 <<'S'
o_base||--no-vision-in-loop
o_kpw768|VISPEC_ATT_KPW=768|--no-vision-in-loop
o_kpw640|VISPEC_ATT_KPW=640|--no-vision-in-loop
o_red1024|VISPEC_REDUCE_THREADS=1024|--no-vision-in-loop
o_red256|VISPEC_REDUCE_THREADS=256|--no-vision-in-loop
o_q16|GPU_MAX_HW_QUEUES=16|--no-vision-in-loop
o_base_b||--no-vision-in-loop
S
cat gpurun_out/r05o_sweep.txt
