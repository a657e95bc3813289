# round 4, call 19: W8A8 with o_proj left on bf16 activations (one quantisation launch per layer less): tests, line A/B against the previous library
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -q > gpurun_out/r04r_pytest_fp8a8.txt 2>&1; tail -4 gpurun_out/r04r_pytest_fp8a8.txt | cut -c1-250
timeout 1500 python -m pytest tests/test_full_size_gpu.py -q -k "fp8a8" -s > gpurun_out/r04r_pytest_fullsize_fp8a8.txt 2>&1; grep -a "full-width\|passed\|failed\|Error" gpurun_out/r04r_pytest_fullsize_fp8a8.txt | cut -c1-500 | tail -6
bash tools/sweep.sh > gpurun_out/r04r_sweep.txt 2>&1 <<'S'
r_a8_a||--model qwen7b-fp8a8
r_a8_b||--model qwen7b-fp8a8
r_fp8||--model qwen7b-fp8
S
cat gpurun_out/r04r_sweep.txt
