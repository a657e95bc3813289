# round 5, call 12: the vision front-end as one hipGraph replay per request vs eager launches vs not in the region (LLaVA-7B, Qwen2.5-VL-7B)
mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/r05i_sweep.txt 2>&1 <<'S'
i_vis_graph||
i_vis_eager|VISPEC_BENCH_VISION_GRAPH=0|
i_novis||--no-vision-in-loop
i_vis_graph_b||
i_qwen_vis_graph||--model qwen7b
i_qwen_vis_eager|VISPEC_BENCH_VISION_GRAPH=0|--model qwen7b
i_qwen_novis||--model qwen7b --no-vision-in-loop
S
cat gpurun_out/r05i_sweep.txt
python - <<'PY'
import json
for t in ("i_vis_graph", "i_qwen_vis_graph"):
    d = json.load(open(f"gpurun_out/sw_{t}.json")); print(t, d["config"]["vision_front_end"][-160:], d["speedpy_comparable"].get("with_vision_tower"))
PY
