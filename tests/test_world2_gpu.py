"""bench.py with WORLD_SIZE = 2, end to end, on ONE GPU: the N > 1 control flow (self-launch of the ranks, rendezvous, weight replication from
rank 0 in both VISPEC_REPLICATE modes, barrier-bracketed timed region, all_reduce of the statistics, the rank-0 line) had never run anywhere
— no multi-GPU node was available to rounds 1-3, and RCCL refuses two ranks on one GPU.  Here both ranks sit on device 0 and talk over gloo
(tools/dryrun_world2.sh is the same recipe by hand).  What is checked is the protocol SURVEY.md §8(e) / bench.py's contract describe, not a
speed: rank 1 starts from different weights and must end with rank 0's (checksums), requests are dealt without overlap, the line's totals
are the sum over ranks and its time the maximum."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import ROOT  # noqa: E402


@pytest.mark.parametrize("mode", ["broadcast", "scatter"])
def test_bench_world_size_two_dry_run_on_one_gpu(tmp_path, mode):
    log = tmp_path / "ranks"
    env = dict(os.environ, VISPEC_FORCE_DEVICE="0", VISPEC_DIST_BACKEND="gloo", VISPEC_REPLICATE=mode, VISPEC_BENCH_RANKLOG=str(log))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--lanes", "1", "--cohort", "2",
           "--no-cpu-baseline", "--no-ar", "--max-new-tokens", "48"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ranks = [json.load(open(log / f"rank{i}.json")) for i in range(2)]
    assert [x["rank"] for x in ranks] == [0, 1] and all(x["world"] == 2 and x["backend"] == "gloo" and x["replicate_mode"] == mode for x in ranks)
    # replication: rank 1 was built from another seed (bench.build_models) — after replicate_weights both hold rank 0's bits
    assert ranks[0]["weights_checksum"] == ranks[1]["weights_checksum"]
    assert "checksums equal" in r.stderr and "regenerat" not in r.stderr, r.stderr[-2000:]
    # the line: whole-job totals over both ranks, the slowest rank's time, one weak-scaling step of lanes x cohort requests per rank
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 1
    assert all(x["tokens"] > 0 for x in ranks)
    a, b = set(ranks[0]["timed_request_ids"]), set(ranks[1]["timed_request_ids"])
    assert len(a) == len(b) == 2 and not a & b, "every request id on exactly one rank"
    assert not (a | b) & (set(ranks[0]["warmup_request_ids"]) | set(ranks[1]["warmup_request_ids"])), "timed requests are not the warm-up's"
    wall = max(x["wall_s"] for x in ranks)
    assert abs(line["value"] - sum(x["tokens"] for x in ranks) / wall) <= 0.02 * line["value"]
    assert abs(line["ms_per_step"] - 1e3 * wall) <= 0.02 * line["ms_per_step"] + 1.0
    assert line["aggregate"]["lanes"] == 1 and line["aggregate"]["cohort"] == 2 and "dp2" in line["config"]["parallelism"]
