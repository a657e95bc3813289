"""world_size-2 gloo tests (CPU) of the N>1 path: weight replication (broadcast, and the scatter + all-gather variant) and request sharding."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, q, mode):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vispec_amd import parallel
    g = torch.Generator().manual_seed(100 + rank)  # different values per rank before replication
    tensors = [torch.randn(1 << 19, generator=g).to(torch.bfloat16), torch.randn(333, 777, generator=g).to(torch.bfloat16),
               torch.randn(17, generator=g), torch.randn((1 << 19) + 3, generator=g)]
    before = parallel.checksum(tensors)
    same_before = parallel.all_equal(before)
    nbytes = parallel.replicate_weights(tensors, src=0, mode=mode)
    after = parallel.checksum(tensors)
    ok = parallel.all_equal(after)
    ref = torch.Generator().manual_seed(100)
    want0 = torch.randn(1 << 19, generator=ref).to(torch.bfloat16)
    q.put((rank, same_before, ok, bool(torch.equal(tensors[0], want0)), nbytes, parallel.shard_requests(7, rank, world)))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["broadcast", "scatter"])
def test_replicate_weights_world2(mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + (7 if mode == "scatter" else 0)
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, same_before, ok, eq0, nbytes, shard in res:
        assert not same_before and ok and eq0 and nbytes > 0
    assert sorted(res[0][5] + res[1][5]) == list(range(7)) and not set(res[0][5]) & set(res[1][5])


@pytest.mark.parametrize("world,lanes,cohort,n_requests", [(1, 4, 4, 0), (2, 4, 4, 0), (8, 4, 4, 0), (8, 4, 4, 64), (2, 3, 2, 7), (8, 4, 1, 64)])
def test_bench_request_plan_covers_every_request_exactly_once(world, lanes, cohort, n_requests):
    """bench.py's work split over ranks x lanes x cohorts (BASELINE config 4: 64 requests over 8 replicas): in every step each request id
    appears on exactly one (rank, lane); weak scaling gives every lane one full cohort per step, the fixed batch is dealt i mod world."""
    sys.path.insert(0, ROOT)
    import bench
    steps = 3
    plans = [bench.request_plan(n_requests, r, world, lanes, cohort, steps) for r in range(world)]
    assert {p[1] for p in plans} == {"strong" if n_requests else "weak"}
    for s in range(steps):
        ids = [i for p, _ in plans for lane in p for i in lane[s]]
        assert len(ids) == len(set(ids))
        if n_requests:
            assert sorted(ids) == [i + s * n_requests for i in range(n_requests)]
            for r, (p, _) in enumerate(plans):
                assert all((i - s * n_requests) % world == r for lane in p for i in lane[s])
        else:
            assert len(ids) == world * lanes * cohort and all(len(lane[s]) == cohort for p, _ in plans for lane in p)
    all_ids = [i for p, _ in plans for lane in p for st in lane for i in st]
    assert len(all_ids) == len(set(all_ids)), "no request is reused across steps"
    if n_requests and cohort > 1:  # cohorts are filled before lanes are opened (64 requests over 8 ranks: 2 lanes x 4, not 4 lanes x 2)
        for p, _ in plans:
            sizes = sorted((len(lane[0]) for lane in p), reverse=True)
            mine = sum(sizes)
            assert sum(1 for z in sizes if z) == max(1, min(lanes, -(-mine // cohort)))
