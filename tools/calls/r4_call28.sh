# round 4, call 28: quant_rows with the row kept in registers between the maximum and the conversion: tests, kernel time, line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -q 2>&1 | tail -2
bash tools/profile_bench.sh r04aa_fp8a8_1lane --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 84
python tools/stats_summary.py gpurun_out/kernel_stats_r04aa_fp8a8_1lane.csv 30 | grep -a "quant_rows\|splitk\|total"
bash tools/sweep.sh > gpurun_out/r04aa_sweep.txt 2>&1 <<'S'
aa_a8_a||--model qwen7b-fp8a8
aa_a8_b||--model qwen7b-fp8a8
S
cat gpurun_out/r04aa_sweep.txt
