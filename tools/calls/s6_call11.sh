# same-box A/B: wide split-K launches XCD-affine (B = tree) vs plain order (A = -DVISPEC_WIDE_XCD_SPLIT=0)
mkdir -p gpurun_out/s6
cp vispec_amd/libvispec_hip.so /tmp/B.so; cp vispec_amd/libvispec_hip_xcd0.so /tmp/A.so
for v in B A B A; do
  cp /tmp/$v.so vispec_amd/libvispec_hip.so
  echo "== $v wide_bench"; timeout 300 python tools/wide_bench.py 0 2>&1 | grep -E "o_proj|down"
  echo "== $v cohort round (1 lane, rb 4)"; timeout 300 python tools/cohort_round_bench.py 4 2>&1 | tail -1
done
for v in B A B A; do
  cp /tmp/$v.so vispec_amd/libvispec_hip.so
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ar > gpurun_out/s6/ab_xcd_$v.json 2>/dev/null
  echo "== $v default line: $(python -c "import json; d=json.loads(open('gpurun_out/s6/ab_xcd_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['aggregate']['slot_utilisation'])")"
done
cp /tmp/B.so vispec_amd/libvispec_hip.so
