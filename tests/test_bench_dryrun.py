"""CPU coverage of bench.py's OWN multi-rank path: `python bench.py --gpus 2` must start two ranks by itself
(no torchrun environment given), shard the batch, run its double-buffered asynchronous gather
(freddy_amd.shard.PipelinedGather, gloo here, RCCL on the GPU box) and print one JSON line with n_gpus = 2.
The search itself is stood in by a (rank, step) pattern -- this is about the distributed plumbing."""
import copy
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
    last = p.stdout.rstrip("\n").splitlines()[-1]
    assert len(last.encode()) < 2000, len(last)      # the driver keeps a 2000-character tail (BENCH_r04: a 21 KB line -> parsed: null)
    return json.loads(last)


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


@pytest.mark.parametrize("every,steps,warmup", [(4, 7, 3), (2, 8, 2), (4, 5, 1)])
def test_bench_grouped_gathers(every, steps, warmup):
    """--gather-every G: one all_gather per group of G steps over a ring of two groups (what --gpus N does by default with G = the
    batches in flight); step counts that end inside a group exercise the flush of a partly written group."""
    out = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", str(steps), "--warmup", str(warmup), "--Q", "64",
               "--gather-every", str(every))
    assert out["n_gpus"] == 2 and out["gather_verified"] is True
    assert out["config"]["gather_every"] == every and out["config"]["collective"] is True


def test_bench_force_collective_single_rank():
    """--force-collective: the world > 1 branch with a ONE-rank process group (on the GPU box: nccl; here: gloo)."""
    out = _run("--dry-run", "--force-collective", "--backend", "gloo", "--steps", "9", "--warmup", "2", "--Q", "32", "--gather-every", "4")
    assert out["n_gpus"] == 1 and out["gather_verified"] is True
    assert out["config"]["collective"] is True and out["config"]["world_size"] == 1 and out["config"]["backend"] == "gloo"


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


def _record():
    """Round 4's committed full record in the shape main() builds now (the duplicates inside `roofline` are gone)."""
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_record_r04.json")))
    abi = rec["roofline"].pop("host_buffer_abi")
    rec["roofline"].pop("other_configs")
    rec["roofline"]["host_abi_q1024_qps"] = abi["Q1024"]["queries_per_s"]
    return rec


def _worst_case_record():
    """A full measurement record as run_ivfadc / main() build it: round 4's committed one (21 KB on one line) with every string
    field inflated, so the size guard is tested against more than any real run produces."""
    rec = _record()
    rec["config"]["workload"] *= 3
    rec["cpu_baseline"]["sample"] *= 4
    rec["dtype"] *= 2
    rec["other_configs"]["more"] = copy.deepcopy(rec["other_configs"])
    return rec


def test_headline_is_what_the_driver_can_keep():
    """the LAST stdout line must parse, stay under 2000 bytes and carry the metric, roofline.frac and
    cpu_baseline.value; the details go to bench_details.json and a digest line of at most 5 KB, each once."""
    sys.path[:0] = [ROOT]
    import bench
    rec = _worst_case_record()
    line = json.dumps(bench.headline(rec))
    assert len(line.encode()) <= bench.HEADLINE_MAX_BYTES < 2000
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in got, k
    assert got["value"] == rec["value"] and got["ms_per_step"] == rec["ms_per_step"]
    assert got["config"]["workload"].startswith("IVFADC batch") and got["config"]["recall_at_5"] == rec["config"]["recall_at_5"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert got["roofline"][k] == rec["roofline"][k], k
    for k in ("value", "unit", "cores", "kind", "value_1_core", "parity_with_gpu_on_sample"):
        assert got["cpu_baseline"][k] == rec["cpu_baseline"][k], k
    assert got["cpu_baseline"]["queries_checked"] == rec["cpu_baseline"]["timed_region_parity"]["queries_checked"]
    assert "other_configs" not in got and "host_buffer_abi" not in got and "other_configs" not in got["roofline"]
    dg = bench.digest(rec)
    assert dg is None or len(json.dumps(dg).encode()) <= bench.DIGEST_MAX_BYTES


def test_emit_prints_the_headline_last_and_the_details_once(tmp_path, monkeypatch, capsys):
    sys.path[:0] = [ROOT]
    import bench
    rec = _record()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    bench.emit(rec)
    lines = capsys.readouterr().out.rstrip("\n").splitlines()
    assert len(lines) == 2 and all(len(l.encode()) <= bench.DIGEST_MAX_BYTES for l in lines)
    last = json.loads(lines[-1])
    assert len(lines[-1].encode()) < 2000 and last["roofline"]["frac"] == rec["roofline"]["frac"] and last["cpu_baseline"]["value"] > 0
    assert "bench_digest" in json.loads(lines[0])
    for d in (tmp_path, tmp_path / "gpurun_out"):
        full = json.load(open(d / "bench_details.json"))
        assert full["other_configs"].keys() == rec["other_configs"].keys() and "other_configs" not in full["roofline"]
