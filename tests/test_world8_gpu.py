"""bench.py with WORLD_SIZE = 8, end to end, on ONE GPU (round 6): BASELINE.json config 3's exact request plan — a fixed batch of 64
independent (image, prompt) requests dealt round-robin over 8 replicas — through the whole N > 1 control flow (self-launch of eight ranks,
rendezvous, weight replication from rank 0, barrier-bracketed timed region, all_reduce of the statistics, every rank leaving the process group
BEFORE rank 0's annotation legs, one line with n_gpus = 8).  No 8-GPU node has existed in any round and RCCL refuses several ranks on one
GPU, so the eight ranks share device 0 and talk over gloo, on bench.py's test-only `tiny` model (LLaVA-shaped, an eighth of the width, four
layers: the same request shape, launch sequence and kernels).  A control-flow check, not a measurement."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import ROOT  # noqa: E402


def test_bench_world_size_eight_config3_plan_on_one_gpu(tmp_path):
    log = tmp_path / "ranks"
    env = dict(os.environ, VISPEC_FORCE_DEVICE="0", VISPEC_DIST_BACKEND="gloo", VISPEC_BENCH_RANKLOG=str(log))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--model", "tiny", "--requests", "64", "--steps", "1", "--warmup", "1",
           "--lanes", "1", "--cohort", "8", "--no-cpu-baseline", "--no-ar", "--no-vision-in-loop", "--max-new-tokens", "40"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1700, cwd=ROOT)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ranks = [json.load(open(log / f"rank{i}.json")) for i in range(8)]
    assert [x["rank"] for x in ranks] == list(range(8)) and all(x["world"] == 8 and x["backend"] == "gloo" for x in ranks)
    # replication: ranks 1..7 were built from other seeds — after replicate_weights all hold rank 0's bits
    assert len({x["weights_checksum"] for x in ranks}) == 1
    assert r.stderr.count("checksums equal") == 8 and "regenerat" not in r.stderr, r.stderr[-2000:]  # (every rank logs its comparison)
    # config 3's plan: 64 request ids, each on exactly one rank, request i -> rank i mod 8 (parallel.shard_requests)
    for x in ranks:
        assert x["timed_request_ids"] == sorted(i + 64 for i in range(x["rank"], 64, 8)), x["timed_request_ids"]  # (step 1 = ids + 64: step 0 warmed up)
        assert x["warmup_request_ids"] == sorted(range(x["rank"], 64, 8))
        assert x["tokens"] > 0 and x["lanes"] == 1
        assert set(x["startup"]) >= {"build_models_s", "weight_replication_s", "warmup_steps_s"}
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["steps"] == 1
    slowest = max(x["wall_s"] for x in ranks)
    assert abs(line["value"] - sum(x["tokens"] for x in ranks) / slowest) <= 0.02 * line["value"]
    assert "dp8" in line["config"]["parallelism"] and line["aggregate"]["cohort"] == 8
    assert "startup" in line and "roofline" in line, "rank 0's annotation legs ran after the job left the process group"
    print(f"world 8 on one GPU: {wall:.0f} s wall, start-up per rank {[x['startup']['build_models_s'] for x in ranks]} s build, "
          f"{[x['startup']['warmup_steps_s'] for x in ranks]} s warm-up")
