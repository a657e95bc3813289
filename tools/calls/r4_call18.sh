# round 4, call 18: Qwen shapes — eight row blocks for every GEMM vs only for the large ones (q|k|v 144 tiles, o_proj / down 112 tiles stay on four)
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r04q_sweep.txt 2>&1 <<'S'
q_qwen_all8||--model qwen7b
q_qwen_big8|VISPEC_WIDE8_TILES_MIN=200|--model qwen7b
q_a8_all8||--model qwen7b-fp8a8
q_a8_big8|VISPEC_WIDE8_TILES_MIN=200|--model qwen7b-fp8a8
q_qwen_all8_b||--model qwen7b
q_qwen_big8_b|VISPEC_WIDE8_TILES_MIN=200|--model qwen7b
q_a8_all8_b||--model qwen7b-fp8a8
q_a8_big8_b|VISPEC_WIDE8_TILES_MIN=200|--model qwen7b-fp8a8
S
cat gpurun_out/r04q_sweep.txt
