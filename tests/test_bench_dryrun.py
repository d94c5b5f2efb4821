"""CPU coverage of bench.py's OWN multi-rank path: `python bench.py --gpus 2` must start two ranks by itself
(no torchrun environment given), shard the batch, run its double-buffered asynchronous gather
(freddy_amd.shard.PipelinedGather, gloo here, RCCL on the GPU box) and print one JSON line with n_gpus = 2.
The search itself is stood in by a (rank, step) pattern -- this is about the distributed plumbing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_gpus_2_spawns_two_ranks_and_gathers(scaling):
    out = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "7", "--warmup", "2", "--Q", "64",
               "--scaling", scaling)
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["warmup"] == 2
    assert out["scaling"] == scaling
    assert out["config"]["parallelism"] == "dp2"
    assert out["config"]["batch_per_gpu"] == (64 if scaling == "weak" else 32)
    assert out["gather_verified"] is True
    assert out["dry_run"] is True and out["value"] > 0


def test_bench_strong_scaling_preset_q8192():
    """`--scaling strong --Q 8192`: the run that can show >= 3.5x at 8 GPUs -- one batch of 8192 queries split
    over the ranks; the line records the world size torch.distributed reports and the backend."""
    out = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--Q", "8192", "--scaling", "strong")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["batch_per_gpu"] == 4096
    assert out["config"]["world_size"] == 2 and out["config"]["backend"] == "gloo"
    assert out["gather_verified"] is True


def test_bench_single_rank_dry_run():
    out = _run("--dry-run", "--steps", "3", "--warmup", "1", "--Q", "16")
    assert out["n_gpus"] == 1 and out["gather_verified"] is True


def test_pipelined_gather_detects_a_wrong_slot():
    """verify_gather must fail when a gathered slot does not hold its rank's result (world 1: trivially true)."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
    import torch
    import bench
    from freddy_amd import shard
    pg = shard.PipelinedGather(4, 2, torch.device("cpu"))
    res = pg.next_buffer()
    bench.fill_pattern(res, 0, 0)
    pg.submit()
    pg.drain()
    assert bench.verify_gather(pg, 0, 1)
