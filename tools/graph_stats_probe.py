import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.getcwd())
import torch, bench, time
bench.MODEL = "llava7b"
dev = torch.device("cuda:0")
sms, tcfg, _ = bench.build_models(dev, 0, 0, 1, 1, 1)
sm = sms[0]
req = bench.make_request(tcfg, 300, dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for rep in range(3):
        o, new_token, idx, acc, t_dec = sm.specgenerate(req[0], max_new_tokens=256, log=True, return_acceptance_len=True, return_decode_time=True, **req[1])
        print(rep, "rounds", idx + 1, "ms/round", round(1e3 * t_dec / (idx + 1), 3), sm.engine.graph_stats(), flush=True)
