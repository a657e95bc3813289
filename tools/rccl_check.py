#!/usr/bin/env python3
"""RCCL smoke of the start-up weight replication on the hardware at hand (vispec_amd/parallel.py): both modes (broadcast,
scatter + all-gather) with the world sizes this box allows — 1 rank (RCCL init + collectives on self) and, with
`--same-gpu`, 2 ranks sharing GPU 0 (RCCL normally refuses duplicate devices; recorded either way).  Prints one JSON line per
(world, mode) with GB/s and the checksum verdict.

    python tools/rccl_check.py [--mb 2048] [--world 1|2] [--same-gpu]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import torch
    import torch.distributed as dist
    from vispec_amd import parallel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0 if args.same_gpu else int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    n = args.mb * (1 << 20) // 2
    for mode in ("broadcast", "scatter"):
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        ts = [torch.randn(n // 4, generator=g, device=dev).to(torch.bfloat16) for _ in range(4)] + [torch.randn(1000, generator=g, device=dev)]
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.time()
        nbytes = parallel.replicate_weights(ts, src=0, mode=mode)
        torch.cuda.synchronize()
        dt = time.time() - t0
        same = parallel.all_equal(parallel.checksum(ts))
        if rank == 0:
            print(json.dumps(dict(world=world, same_gpu=bool(args.same_gpu), mode=mode, GB=round(nbytes / 1e9, 3), seconds=round(dt, 4),
                                  GBps=round(nbytes / dt / 1e9, 1), checksums_equal=bool(same), backend="nccl (RCCL)")), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=2048)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--same-gpu", action="store_true")
    args = ap.parse_args()
    if "RANK" in os.environ:
        return worker(args)
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 124
    if rc:
        print(json.dumps(dict(world=args.world, same_gpu=bool(args.same_gpu), error=f"exit code {rc}")), flush=True)
    sys.exit(0)


if __name__ == "__main__":
    main()
