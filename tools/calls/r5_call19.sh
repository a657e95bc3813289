# round 5, call 19: after the bench.py split — the world-2 dry run (self_launch moved), the new cohort tree-shape tests, one short default line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_world2_gpu.py tests/test_c8_gpu.py -q -m gpu -k "world or tree_shapes or refuses" 2>&1 | tail -3
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_after_split.json 2> gpurun_out/r05_after_split.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05_after_split.json').read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['roofline'].get('requests_per_launch'), d['roofline'].get('weight_bytes_delivered_to_requests_GBps'), d['host']['affinity'], d.get('extra_legs_error'))"
