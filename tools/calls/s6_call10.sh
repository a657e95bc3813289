mkdir -p gpurun_out/s6
for mn in 512 1300; do
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ar --max-new-tokens $mn > gpurun_out/s6/maxnew_$mn.json 2> gpurun_out/s6/maxnew_$mn.err
  python - $mn <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/s6/maxnew_{sys.argv[1]}.json").read().strip().splitlines()[-1])
a = d["aggregate"]
print("max_new", sys.argv[1], "tok/s", d["value"], "tau", d["mean_accept_length_tau"], "request-rounds/s", a["request_rounds_per_s_per_gpu"], "GB/rr", a["algorithmic_GB_per_request_round"],
      "streamed", a["streamed_GBps_per_gpu"], "slot util", a["slot_utilisation"], "1req prefill ms", d["speedpy_comparable"].get("prefill_ms"), flush=True)
PY
done
