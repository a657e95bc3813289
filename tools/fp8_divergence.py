"""BASELINE config 5 (Qwen2.5-VL-7B, fp8 target weights, one 1280x960 image): what the two fp8 arithmetics change in the GENERATED TOKENS.
SURVEY.md §7.1 step 8 allows "same accepted tokens ... or documented divergence rate"; this is the documentation: N seeded full-size requests
(bench.py's config-5 request: L = 2124, 1564 image tokens; synthetic successor-structured weights, the same seed for all three models) decoded
greedily by (a) the bf16 model, (b) W8A16 (e4m3 weights, bf16 activations), (c) W8A8 (e4m3 weights and activations on the fp8 MFMA) — each
through the cohort-8 path — and compared token by token with (a): requests whose whole continuation is identical, mean length of the common
prefix, tokens equal position by position, mean accept length of each model's own speculative run.
    python tools/fp8_divergence.py [N=64] [NEW=128]  ->  one JSON object on stdout"""
import gc
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from vispec_amd.model.spec_model_ours import specgenerate_stream  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NEW = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
out = {}
runs = {}
for name in ("qwen7b-hires", "qwen7b-fp8", "qwen7b-fp8a8"):
    bench.MODEL = name
    sms, tcfg, _ = bench.build_models(dev, 0, 0, 1, 1, 8)
    models = sms[0]
    reqs = [bench.make_request(tcfg, 7000 + i, dev) for i in range(N)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        got = specgenerate_stream(models, reqs, max_new_tokens=NEW, seeds=list(range(N)))
        side.synchronize()
    L = reqs[0][0].shape[1]
    runs[name] = [g[0][0, L:L + NEW].cpu().numpy() for g in got]
    taus = [float(np.mean(g[3])) for g in got]
    out[name] = dict(mean_accept_length=round(float(np.mean(taus)), 3), new_tokens_per_request=NEW, requests=N)
    del sms, models, reqs, got
    gc.collect()
    torch.cuda.empty_cache()
ref = runs["qwen7b-hires"]
for name in ("qwen7b-fp8", "qwen7b-fp8a8"):
    same_req, prefix, equal_pos = 0, [], []
    for a, b in zip(ref, runs[name]):
        n = min(len(a), len(b))
        eq = a[:n] == b[:n]
        same_req += int(eq.all())
        prefix.append(int(np.argmin(eq)) if not eq.all() else n)
        equal_pos.append(float(eq.mean()))
    out[name].update(requests_identical_to_bf16=same_req, mean_common_prefix_tokens=round(float(np.mean(prefix)), 1),
                     tokens_equal_position_by_position=round(float(np.mean(equal_pos)), 4))
a8, a16 = runs["qwen7b-fp8a8"], runs["qwen7b-fp8"]
out["qwen7b-fp8a8"]["requests_identical_to_w8a16"] = int(sum(int((x[:min(len(x), len(y))] == y[:min(len(x), len(y))]).all()) for x, y in zip(a8, a16)))
print(json.dumps(out, indent=1))
