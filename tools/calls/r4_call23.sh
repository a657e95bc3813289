# round 4, call 23: W8A8 in the prefill too (row-quantised activations x e4m3 codes on the library's fp8 x fp8 GEMM): tests, prefill breakdown, line A/B
mkdir -p gpurun_out
cp gpurun_out/../gpurun_out/.keep /dev/null 2>/dev/null
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -q > gpurun_out/r04v_pytest_fp8a8.txt 2>&1; tail -15 gpurun_out/r04v_pytest_fp8a8.txt | cut -c1-250
timeout 1500 python -m pytest tests/test_full_size_gpu.py -q -k "fp8a8" -s > gpurun_out/r04v_pytest_fullsize_fp8a8.txt 2>&1; grep -a "full-width\|passed\|failed\|Error" gpurun_out/r04v_pytest_fullsize_fp8a8.txt | cut -c1-500 | tail -8
MODEL=qwen7b-fp8a8 timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04v_prefill_breakdown_fp8a8.txt 2>&1; grep -a "^iter\|Cijk\|scaled\|quant_rows\|scale_bias" gpurun_out/r04v_prefill_breakdown_fp8a8.txt | cut -c1-200 | head -14
bash tools/sweep.sh > gpurun_out/r04v_sweep.txt 2>&1 <<'S'
v_a8_a||--model qwen7b-fp8a8
v_a8_b||--model qwen7b-fp8a8
v_fp8||--model qwen7b-fp8
S
cat gpurun_out/r04v_sweep.txt
