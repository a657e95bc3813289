# round 5, call 5: W8A8 on the cohort-8 kernel (unit tests); every BASELINE model at cohort 8 next to cohort 4, same box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_c8_gpu.py -x -q -m gpu -k "fp8_activations" 2>&1 | tail -5
bash tools/sweep.sh > gpurun_out/r05e_sweep.txt 2>&1 <<'S'
e_13b_l4c4||--model llava13b --lanes 4 --cohort 4
e_13b_l3c8||--model llava13b --lanes 3 --cohort 8
e_qwen_l4c4||--model qwen7b --lanes 4 --cohort 4
e_qwen_l4c8||--model qwen7b --lanes 4 --cohort 8
e_fp8_l4c4||--model qwen7b-fp8 --lanes 4 --cohort 4
e_fp8_l4c8||--model qwen7b-fp8 --lanes 4 --cohort 8
e_a8_l4c4||--model qwen7b-fp8a8 --lanes 4 --cohort 4
e_a8_l4c8||--model qwen7b-fp8a8 --lanes 4 --cohort 8
S
cat gpurun_out/r05e_sweep.txt
tail -3 gpurun_out/sw_e_13b_l3c8.err
