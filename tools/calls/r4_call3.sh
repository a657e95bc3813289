# round 4, call 3: new entry points (cohort AR step, leader destroy order), the bench line with its new legs, the fp8 configuration under the
# eight-row-block form (per-kernel), where the fp8 prefill's 145 ms go, and which GEMMs should take the eight-row-block form
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wide_gpu.py -q -x -k "cohort_ar or destroying or step_api or member_slots or closing" > gpurun_out/r04c_pytest.txt 2>&1; tail -5 gpurun_out/r04c_pytest.txt
timeout 900 python bench.py --steps 2 --warmup 1 --wide-row-blocks 8 > gpurun_out/r04c_line_rb8.json 2> gpurun_out/r04c_line_rb8.err; tail -c 3000 gpurun_out/r04c_line_rb8.json
bash tools/profile_bench.sh r04c_qwen7bfp8_1lane_rb8 --model qwen7b-fp8 --lanes 1 --cohort 4 --wide-row-blocks 8
bash tools/profile_bench.sh r04c_qwen7bfp8_4lanes_rb8 --model qwen7b-fp8 --lanes 4 --cohort 4 --wide-row-blocks 8
bash tools/profile_bench.sh r04c_llava_4lanes_rb8 --lanes 4 --cohort 4 --wide-row-blocks 8
MODEL=qwen7b-fp8 timeout 600 python tools/prefill_breakdown.py > gpurun_out/r04c_prefill_fp8.txt 2>&1; head -40 gpurun_out/r04c_prefill_fp8.txt
bash tools/sweep.sh > gpurun_out/r04c_sweep.txt 2>&1 <<'S'
llava_rb8||--wide-row-blocks 8
llava_rb8_big|VISPEC_WIDE8_TILES_MIN=300|--wide-row-blocks 8
llava_rb8_small|VISPEC_WIDE8_TILES_MAX=300|--wide-row-blocks 8
llava_rb8_notlm|VISPEC_WIDE8_TILES_MAX=800|--wide-row-blocks 8
llava_rb4||
qwenfp8_rb8_big|VISPEC_WIDE8_TILES_MIN=300|--model qwen7b-fp8 --wide-row-blocks 8
qwenfp8_rb8_small|VISPEC_WIDE8_TILES_MAX=300|--model qwen7b-fp8 --wide-row-blocks 8
S
cat gpurun_out/r04c_sweep.txt
