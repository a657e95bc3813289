# round 5, call 26: the whole GPU suite and the driver's command at the final head
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r05_gputests_head.txt; cat gpurun_out/r05_gputests_head.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line_driver_cmd.json 2> gpurun_out/r05_bench_line_driver_cmd.err ) 2>&1 | tail -3
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05_bench_line_driver_cmd.json').read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['aggregate']['frac_of_8TBps'], d.get('speedup_vs_ar'), d['speedpy_comparable']['ms_per_round'], d['roofline']['frac'], d['roofline'].get('requests_per_launch'), d.get('extra_legs_error'))"
