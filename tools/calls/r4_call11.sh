# round 4, call 11: where do the waves of the fp8 cohort GEMMs spend their time?  SQ counters per kernel: bf16 / W8A16 / W8A8 (Qwen2.5-VL-7B)
mkdir -p gpurun_out
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
bash tools/pmc_counters.sh r04_sq_qwen7b "$C" --model qwen7b --lanes 1 --cohort 4 --wide-row-blocks 4 > /dev/null 2>&1
bash tools/pmc_counters.sh r04_sq_qwen7bfp8 "$C" --model qwen7b-fp8 --lanes 1 --cohort 4 --wide-row-blocks 4 > /dev/null 2>&1
bash tools/pmc_counters.sh r04_sq_qwen7bfp8a8 "$C" --model qwen7b-fp8a8 --lanes 1 --cohort 4 --wide-row-blocks 4 > /dev/null 2>&1
python - <<'PY'
import json
for tag in ("qwen7b", "qwen7bfp8", "qwen7bfp8a8"):
    d = json.load(open(f"gpurun_out/pmc_r04_sq_{tag}.json"))
    for k, v in d.items():
        if "wide_kernel" in k:
            wc = v["SQ_WAVE_CYCLES"]
            print(f"{tag:12s} {k[:46]:46s} n {v['launches']:6d} gui {v['GRBM_GUI_ACTIVE']:9.0f} wave_cyc {wc:11.0f} wait {v['SQ_WAIT_ANY']/wc:5.2f} stall {v['SQ_WAIT_INST_ANY']/wc:5.2f} issue {v['SQ_ACTIVE_INST_ANY']/wc:5.2f} valu {v['SQ_ACTIVE_INST_VALU']/wc:5.2f} mfma_busy/gui/1024 {v['SQ_VALU_MFMA_BUSY_CYCLES']/v['GRBM_GUI_ACTIVE']/1024:5.2f}")
PY
timeout 900 python -m pytest tests/test_fp8a8_gpu.py -q > gpurun_out/r04j_pytest_fp8a8.txt 2>&1; tail -4 gpurun_out/r04j_pytest_fp8a8.txt | cut -c1-200
