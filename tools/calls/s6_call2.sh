# the default line's siblings at HEAD (continuous batching + graph cache), each the profiles/r03_bench_line_extra_* command
mkdir -p gpurun_out/s6
run() { tag=$1; shift; timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/s6/extra_$tag.json 2> gpurun_out/s6/extra_$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('gpurun_out/s6/extra_$tag.json').read().strip().splitlines()[-1]); print(d['value'], 'tau', d['mean_accept_length_tau'], '1req', d['speedpy_comparable'].get('tokens_per_s'), d['speedpy_comparable'].get('ms_per_round'), 'specEQar', d.get('spec_equals_ar_prefix'))" 2>&1 | tail -1)"; }
run llava13b --model llava13b
run llava13b_requests64 --model llava13b --requests 64
run qwen7b --model qwen7b
run qwen7b-fp8 --model qwen7b-fp8
run llava7b_T1 --temperature 1.0
run llava7b_img2928 --n-img 2928
