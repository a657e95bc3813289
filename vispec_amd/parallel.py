"""Multi-GPU: one process per GPU, independent (image, prompt) requests per replica, NO data-path collective
(SURVEY.md §8e — "replicas only").  The only communication is the one-time replication of the weights from rank 0
at start-up: by default one RCCL broadcast per tensor (what BASELINE.json's north star names; 14 GB is a fraction of a
second of xGMI time either way).  `mode="scatter"` (env VISPEC_REPLICATE=scatter) splits every large tensor into G shards —
scatter, so each peer link of the root carries 1/G of the payload, then all-gather — which keeps every xGMI link busy
instead of the root's egress alone; it is exercised by the gloo test but has not run on RCCL hardware yet, hence opt-in."""
from __future__ import annotations

from typing import Iterable

import torch
import torch.distributed as dist

SMALL = 1 << 20


def replicate_weights(tensors: Iterable[torch.Tensor], src: int = 0, group=None, mode: str = None) -> int:
    """In-place: after the call every rank holds rank `src`'s values.  Returns the number of bytes replicated."""
    import os
    mode = mode or os.environ.get("VISPEC_REPLICATE", "broadcast")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = 0
    # gloo moves CUDA tensors for broadcast / all_reduce only: the dry run of the N > 1 control flow on one GPU (bench.py with
    # VISPEC_DIST_BACKEND=gloo) stages every tensor through the host; RCCL ("nccl") works on the device tensors themselves
    via_host = world > 1 and dist.get_backend(group) == "gloo"
    for t_dev in tensors:
        assert t_dev.is_contiguous()
        total += t_dev.numel() * t_dev.element_size()
        if world == 1:
            continue
        t = t_dev.cpu() if via_host and t_dev.is_cuda else t_dev
        _replicate_one(t, src, group, mode, world, rank)
        if t is not t_dev and rank != src:
            t_dev.copy_(t)
    return total


def _replicate_one(t, src, group, mode, world, rank):
    """One tensor: a broadcast, or (mode "scatter") scatter of 1/G shards over the root's peer links + all-gather."""
    if mode != "scatter":
        dist.broadcast(t, src, group=group)
        return
    flat = t.view(-1)
    n = flat.numel()
    shard = n // world
    if n * t.element_size() < SMALL or shard == 0:
        dist.broadcast(t, src, group=group)
        return
    body = flat[: shard * world]
    mine = torch.empty(shard, dtype=t.dtype, device=t.device)
    dist.scatter(mine, list(body.split(shard)) if rank == src else None, src=src, group=group)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    if rank != src:
        torch.cat(parts, out=body)
    if shard * world < n:
        tail = flat[shard * world :].clone()
        dist.broadcast(tail, src, group=group)
        if rank != src:
            flat[shard * world :] = tail


def checksum(tensors: Iterable[torch.Tensor]) -> torch.Tensor:
    """Order-sensitive 64-bit checksum of the raw bits (exact, integer arithmetic)."""
    acc = None
    for i, t in enumerate(tensors):
        b = t.contiguous().view(torch.uint8).view(-1)
        pad = (-b.numel()) % 8
        if pad:
            b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
        w = b.view(torch.int64)
        s = (w * (2 * i + 1)).sum() + w[:: max(1, w.numel() // 1024)].sum() * 31
        acc = s if acc is None else acc * 1000003 + s
    return acc


def all_equal(value: torch.Tensor, group=None) -> bool:
    if dist.get_world_size(group) == 1:
        return True
    lo, hi = value.clone(), value.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool((lo == hi).all())


def shard_requests(n_requests: int, rank: int, world: int):
    """request i -> replica i mod G (static round-robin; the reference chunks contiguously, gen_spec_answer_coco_caption.py:63-80)."""
    return list(range(rank, n_requests, world))
