# round 4, call 1: same-box baselines of the three bench configurations the verdict names + rocprofv3 kernel stats of the Qwen configurations
# (one cohort lane, and four lanes for the fp8 one) + the wide kernel shape by shape on the Qwen2.5-VL-7B shapes
mkdir -p gpurun_out
bash tools/profile_bench.sh r04a_qwen7b_1lane_cohort4 --model qwen7b --lanes 1 --cohort 4 --wide-row-blocks 4
bash tools/profile_bench.sh r04a_qwen7bfp8_1lane_cohort4 --model qwen7b-fp8 --lanes 1 --cohort 4 --wide-row-blocks 4
bash tools/profile_bench.sh r04a_qwen7bfp8_4lanes_cohort4 --model qwen7b-fp8 --lanes 4 --cohort 4
SHAPES=qwen7b python tools/wide_bench.py 0 3 4 1 2 > gpurun_out/r04a_wide_bench_qwen.txt 2>&1
bash tools/sweep.sh > gpurun_out/r04a_sweep.txt 2>&1 <<'S'
llava||
qwen|| --model qwen7b
qwenfp8|| --model qwen7b-fp8
S
cat gpurun_out/r04a_sweep.txt; cat gpurun_out/r04a_wide_bench_qwen.txt
