"""How much aggregate throughput do R concurrent batch-1 requests (R streams, R ctxs sharing one copy of the weights) buy on one GPU?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vispec_amd import synth_gpu
from vispec_amd.engine import LLAVA_16_7B, DraftConfig, TargetConfig
from vispec_amd.model import SpecModel
from vispec_amd.model.cnets_ours import Model
from vispec_amd.model.target import TargetLM

dev = torch.device("cuda:0")
tcfg = TargetConfig(**LLAVA_16_7B)
dcfg = DraftConfig(hidden_size=4096, num_heads=32, intermediate_size=11008, vocab_size=32064, max_position_embeddings=4096)
tw, dw = synth_gpu.make_pair(tcfg, dcfg, dev, seed=0, structured=True, num_q=2)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
models = []
for r in range(R):
    sm = SpecModel(TargetLM(tcfg, tw), Model(dcfg, dw, total_tokens=30, depth=3, top_k=8, num_q=2), total_token=30, depth=3, top_k=8, num_q=2)
    models.append(sm)
reqs = [bench.make_request(tcfg, i, dev) for i in range(4 * R)]
streams = [torch.cuda.Stream(dev) for _ in range(R)]
results = [0] * R

def worker(r, idxs):
    with torch.cuda.stream(streams[r]):
        tok = 0
        for i in idxs:
            ids, pix = reqs[i]
            out, new_token, idx, acc = models[r].specgenerate(ids, max_new_tokens=512, log=True, return_acceptance_len=True, **pix)
            tok += int(new_token)
        streams[r].synchronize()
        results[r] = tok

for phase in ("warmup", "timed"):
    torch.cuda.synchronize()
    t0 = time.time()
    ths = [threading.Thread(target=worker, args=(r, list(range(r, len(reqs) if phase == "timed" else R, R)))) for r in range(R)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(phase, "R =", R, "tokens", sum(results), "time %.3f" % dt, "aggregate tok/s %.1f" % (sum(results) / dt), flush=True)
