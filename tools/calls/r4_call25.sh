# round 4, call 25: the W8A8 prefill with recorded solutions for its fp8 x fp8 GEMMs: breakdown, tests of the fp8a8 model, the line
mkdir -p gpurun_out
MODEL=qwen7b-fp8a8 timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04x_prefill_breakdown_fp8a8.txt 2>&1; grep -a "^iter\|Cijk\|quant_rows" gpurun_out/r04x_prefill_breakdown_fp8a8.txt | cut -c1-200 | head -10
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -k fp8a8 2>&1 | tail -2
bash tools/sweep.sh > gpurun_out/r04x_sweep.txt 2>&1 <<'S'
x_a8_a||--model qwen7b-fp8a8
x_a8_default_gemms|VISPEC_PREFILL_GEMMS=default|--model qwen7b-fp8a8
x_a8_b||--model qwen7b-fp8a8
S
cat gpurun_out/r04x_sweep.txt
