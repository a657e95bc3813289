# round 5, call 11: vision prefetch on/off, W8A16 two-tile launch bounds (no scratch) vs round 4's, the fp8 divergence table
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r05h_sweep.txt 2>&1 <<'S'
h_vis_prefetch||
h_vis_inline|VISPEC_BENCH_VISION_PREFETCH=0|
h_novis||--no-vision-in-loop
h_fp8_new||--model qwen7b-fp8
h_fp8_oldw8|VISPEC_LIB_VARIANT=oldw8|--model qwen7b-fp8
h_fp8_new_c2||--model qwen7b-fp8 --cohort 2
h_fp8_oldw8_c2|VISPEC_LIB_VARIANT=oldw8|--model qwen7b-fp8 --cohort 2
S
cat gpurun_out/r05h_sweep.txt
timeout 1200 python tools/fp8_divergence.py 64 128 > gpurun_out/r05_fp8_divergence.json 2> gpurun_out/r05_fp8_divergence.err; cat gpurun_out/r05_fp8_divergence.json; tail -3 gpurun_out/r05_fp8_divergence.err
