"""One lane, one cohort of N requests (bench.py's LLaVA-7B workload): milliseconds per LOCKSTEP round of the decode loop alone (prefills
excluded), i.e. the time of the launch sequence a cohort round is.   python tools/cohort_round_bench.py [N=4] [max_new=256] [model]
Profile it with   rocprofv3 --kernel-trace --stats -- python tools/cohort_round_bench.py 4   to split a cohort round by kernel."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from vispec_amd.model.spec_model_ours import specgenerate_cohort  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NEW = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bench.MODEL = sys.argv[3] if len(sys.argv) > 3 else "llava7b"
dev = torch.device("cuda:0")
sms, tcfg, _ = bench.build_models(dev, 0, 0, 1, 1, N)
models = sms[0] if N >= 2 else [sms[0]]
reqs = [bench.make_request(tcfg, 300 + i, dev) for i in range(N)]
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for rep in range(2):  # first pass: graph capture, flash backend init
        if N >= 2:
            st = {}
            outs = specgenerate_cohort(models, reqs, max_new_tokens=NEW, seeds=list(range(N)), stats=st)
            toks = sum(int(o[1]) for o in outs)
            t_dec, rounds = st["decode_s"], st["rounds"]
        else:
            o, new_token, idx, acc, t_dec = models[0].specgenerate(reqs[0][0], max_new_tokens=NEW, log=True, return_acceptance_len=True,
                                                                   return_decode_time=True, **reqs[0][1])
            toks, rounds = int(new_token), idx + 1
        s.synchronize()
print(f"cohort {N} ({bench.MODEL}): {rounds} lockstep rounds, {toks} tokens, decode {t_dec * 1e3:.1f} ms -> {t_dec * 1e3 / rounds:.3f} ms per cohort round, "
      f"{t_dec * 1e3 / rounds / N:.3f} ms per request-round, decode-only {toks / t_dec:.0f} tok/s")
