# round 5, call 20: SQ counters of the cohort-8 kernels (MFMA busy, wave wait / stall / issue), and the concurrency figures of the 4 x 8 line's kernel trace
mkdir -p gpurun_out
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
bash tools/pmc_counters.sh r05_sq_cohort8 "$C" --lanes 1 --cohort 8 --no-vision-in-loop > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_r05_sq_cohort8.json"))
for k, v in d.items():
    if "c8_kernel" in k or "tree_attn2" in k or "2, true" in k:
        wc = v["SQ_WAVE_CYCLES"]
        print(f"{k[:52]:52s} n {v['launches']:6d} gui {v['GRBM_GUI_ACTIVE']:9.0f} wait {v['SQ_WAIT_ANY']/wc:5.2f} stall {v['SQ_WAIT_INST_ANY']/wc:5.2f} issue {v['SQ_ACTIVE_INST_ANY']/wc:5.2f} valu {v['SQ_ACTIVE_INST_VALU']/wc:5.2f} mfma_busy/gui/1024 {v['SQ_VALU_MFMA_BUSY_CYCLES']/v['GRBM_GUI_ACTIVE']/1024:5.2f}")
PY
export TMPDIR=/tmp; rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ar > /dev/null 2> gpurun_out/r05_trace.err
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); python tools/trace_concurrency.py "$f" gpurun_out/r05_trace_concurrency_4lanes_cohort8.json | tail -30
