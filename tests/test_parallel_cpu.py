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


@pytest.mark.parametrize("world,lanes,cohort,n_requests", [(1, 4, 4, 0), (2, 4, 4, 0), (8, 4, 4, 0), (8, 4, 4, 64), (2, 3, 2, 7), (8, 4, 1, 64), (1, 4, 8, 0), (8, 4, 8, 0), (8, 3, 8, 64), (2, 4, 8, 37)])
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


def test_bench_resolves_real_checkpoint_directories(tmp_path, monkeypatch):
    """bench.py --weights-dir / $VISPEC_WEIGHTS / --base-model-path + --spec-model-path (real checkpoints when present): the
    published pair is looked up under its hub names (with or without the organisation directory), explicit paths win, half a pair or a
    missing directory is refused, nothing given -> synthetic weights."""
    sys.path.insert(0, ROOT)
    import bench
    from types import SimpleNamespace as NS
    monkeypatch.delenv("VISPEC_WEIGHTS", raising=False)
    bench.MODEL = "llava7b"
    assert bench.resolve_weights(NS(base_model_path=None, spec_model_path=None, weights_dir=None)) is None
    root = tmp_path / "w"
    (root / "llava-hf" / "llava-v1.6-vicuna-7b-hf").mkdir(parents=True)
    (root / "JLKang" / "ViSpec-llava-v1.6-vicuna-7b-hf").mkdir(parents=True)
    got = bench.resolve_weights(NS(base_model_path=None, spec_model_path=None, weights_dir=str(root)))
    assert got == (str(root / "llava-hf" / "llava-v1.6-vicuna-7b-hf"), str(root / "JLKang" / "ViSpec-llava-v1.6-vicuna-7b-hf"))
    monkeypatch.setenv("VISPEC_WEIGHTS", str(root))
    assert bench.resolve_weights(NS(base_model_path=None, spec_model_path=None, weights_dir=None)) == got
    flat = tmp_path / "flat"
    (flat / "Qwen2.5-VL-7B-Instruct").mkdir(parents=True)
    (flat / "ViSpec-Qwen2.5-VL-7B-Instruct").mkdir(parents=True)
    bench.MODEL = "qwen7b-fp8"  # the fp8 / high-res variants use the Qwen2.5-VL-7B pair
    assert bench.resolve_weights(NS(base_model_path=None, spec_model_path=None, weights_dir=str(flat))) == (
        str(flat / "Qwen2.5-VL-7B-Instruct"), str(flat / "ViSpec-Qwen2.5-VL-7B-Instruct"))
    for variant in ("qwen7b-fp8a8", "qwen7b-hires"):  # (round-4 advice: "-fp8a8" used to leave "qwen7ba8" and fall back to synthetic weights silently)
        bench.MODEL = variant
        assert bench.resolve_weights(NS(base_model_path=None, spec_model_path=None, weights_dir=str(flat))) == (
            str(flat / "Qwen2.5-VL-7B-Instruct"), str(flat / "ViSpec-Qwen2.5-VL-7B-Instruct"))
    bench.MODEL = "llava13b"  # not in this directory: synthetic
    assert bench.resolve_weights(NS(base_model_path=None, spec_model_path=None, weights_dir=str(flat))) is None
    bench.MODEL = "llava7b"
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(), b.mkdir()
    assert bench.resolve_weights(NS(base_model_path=str(a), spec_model_path=str(b), weights_dir=str(root))) == (str(a), str(b))
    monkeypatch.delenv("VISPEC_WEIGHTS")
    with pytest.raises(SystemExit):
        bench.resolve_weights(NS(base_model_path=str(a), spec_model_path=None, weights_dir=None))
    with pytest.raises(SystemExit):
        bench.resolve_weights(NS(base_model_path=str(a), spec_model_path=str(tmp_path / "missing"), weights_dir=None))


def test_bench_host_helpers_degrade_without_a_gpu_or_numa_information(monkeypatch):
    """bench.py pins a rank's lane threads to the NUMA node of its GPU and reports what the rank costs the host; on a box without a GPU (here),
    without /sys NUMA files, or with VISPEC_BENCH_AFFINITY=0 the pinning is a no-op that says why — it never raises and never narrows the affinity."""
    from vispec_amd.evaluation.bench_launch import host_usage, pin_to_gpu_numa_node
    before = os.sched_getaffinity(0)
    note = pin_to_gpu_numa_node(0)
    assert note.startswith("none") and os.sched_getaffinity(0) == before
    monkeypatch.setenv("VISPEC_BENCH_AFFINITY", "0")
    assert pin_to_gpu_numa_node(0).startswith("off") and os.sched_getaffinity(0) == before
    cpu_s, rss_gb = host_usage()
    assert cpu_s > 0 and 0 < rss_gb < 64
