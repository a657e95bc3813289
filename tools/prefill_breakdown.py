"""Where does the per-request prefill time go?    Times the phases of SpecModel.specgenerate before the first
round on the headline workload and prints the top kernels of the PyTorch target prefill.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

bench.MODEL = os.environ.get("MODEL", "llava7b")  # llava7b | llava13b | qwen7b | qwen7b-hires | qwen7b-fp8
dev = torch.device("cuda:0")
sms, tcfg, _ = bench.build_models(dev, 0, 0, 1, 1)
sm = sms[0]
ids, pix = bench.make_request(tcfg, 0, dev)
eng = sm.engine


def sync():
    torch.cuda.synchronize()
    return time.time()


for it in range(3):
    t0 = sync()
    sm.spec_layer.reset_kv()
    emb_in, mask, demb, pos, rd = sm._merge_vision(ids.clone(), None, dict(pix))
    emb = emb_in.reshape(-1, emb_in.shape[-1]).to(torch.bfloat16).contiguous()
    t1 = sync()
    logits, hidden = sm.base_model.prefill(emb, position_ids=pos)
    t2 = sync()
    first = sm._first_token(logits)
    eng.begin_request(ids[0].cpu().numpy(), 512)
    mask_np = None if mask is None else mask.reshape(-1).cpu().numpy()
    t3 = sync()
    eng.draft_prefill(hidden, emb, mask_np, first)
    t4 = sync()
    print(f"iter {it}: merge_vision {1e3*(t1-t0):.1f} ms | target prefill {1e3*(t2-t1):.1f} ms | first+begin {1e3*(t3-t2):.1f} ms | "
          f"draft prefill {1e3*(t4-t3):.1f} ms | total {1e3*(t4-t0):.1f} ms  (L={emb.shape[0]})", flush=True)

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    sm.base_model.prefill(emb, position_ids=pos)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    eng.begin_request(ids[0].cpu().numpy(), 512)
    eng.draft_prefill(hidden, emb, mask_np, first)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
