# round 4, call 16: is the throughput cliff at five lanes a hardware-queue limit?  lanes 4 / 5 / 6 with GPU_MAX_HW_QUEUES = 8 (default of bench.py) / 16 / 24
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r04o_sweep.txt 2>&1 <<'S'
o_l4_q8||
o_l5_q8||--lanes 5
o_l5_q16|GPU_MAX_HW_QUEUES=16|--lanes 5
o_l6_q16|GPU_MAX_HW_QUEUES=16|--lanes 6
o_l6_q24|GPU_MAX_HW_QUEUES=24|--lanes 6
o_l4_q16|GPU_MAX_HW_QUEUES=16|
o_a8_l5_q16|GPU_MAX_HW_QUEUES=16|--model qwen7b-fp8a8 --lanes 5
o_a8_l6_q16|GPU_MAX_HW_QUEUES=16|--model qwen7b-fp8a8 --lanes 6
S
cat gpurun_out/r04o_sweep.txt
