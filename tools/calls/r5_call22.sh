# round 5, call 22: final records of the head — the whole GPU suite, smoke, the driver's command
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r05_gputests_head.txt; cat gpurun_out/r05_gputests_head.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line_driver_cmd.json 2> gpurun_out/r05_bench_line_driver_cmd.err ) 2>&1 | tail -3
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05_bench_line_driver_cmd.json').read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['aggregate'], d.get('speedup_vs_ar'), d['speedpy_comparable']['ms_per_round'], d['host'], d.get('extra_legs_error'))"
