# round 4, call 22: concurrency of the default line from a kernel trace (who is in flight with whom, workgroups asked for); the W8A8 line's JSON at the head
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ar > gpurun_out/r04u_trace_bench.json 2> gpurun_out/r04u_trace_bench.err
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); ls -la $f; head -1 $f
python tools/trace_concurrency.py $f gpurun_out/r04_trace_concurrency_4lanes.json
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ar --wide-row-blocks 4 > gpurun_out/r04u_trace_bench_rb4.json 2> gpurun_out/r04u_trace_bench_rb4.err
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_concurrency.py $f gpurun_out/r04_trace_concurrency_4lanes_rb4.json
timeout 900 python bench.py --model qwen7b-fp8a8 > gpurun_out/r04u_bench_fp8a8.json 2> gpurun_out/r04u_bench_fp8a8.err; cut -c1-200 gpurun_out/r04u_bench_fp8a8.json
