# round 4, call 26: one round of lookahead in the continuous-batching loop (the stream never waits for the host between rounds): tests, A/B
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_wide_gpu.py tests/test_loop_gpu.py tests/test_fp8a8_gpu.py -q -k "stream or continuous or cohort" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_full_size_gpu.py -q -k "default_bench" 2>&1 | tail -2
bash tools/sweep.sh > gpurun_out/r04y_sweep.txt 2>&1 <<'S'
y_la1_a||
y_la0_a|VISPEC_STREAM_LOOKAHEAD=0|
y_la1_b||
y_la0_b|VISPEC_STREAM_LOOKAHEAD=0|
y_qwen_la1||--model qwen7b
y_qwen_la0|VISPEC_STREAM_LOOKAHEAD=0|--model qwen7b
y_a8_la1||--model qwen7b-fp8a8
y_a8_la0|VISPEC_STREAM_LOOKAHEAD=0|--model qwen7b-fp8a8
S
cat gpurun_out/r04y_sweep.txt
