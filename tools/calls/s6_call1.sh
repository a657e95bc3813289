set -x
mkdir -p gpurun_out/s6
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/s6/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s6/gputests.log
tail -25 gpurun_out/s6/gputests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s6/bench_default.json 2> gpurun_out/s6/bench_default.err; echo "bench rc=$?"
head -c 1500 gpurun_out/s6/bench_default.json
