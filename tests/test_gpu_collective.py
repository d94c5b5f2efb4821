"""-m gpu: the world > 1 branch of bench.py on the ONE GPU of the box -- a one-rank `nccl` (RCCL) process group, the asynchronous
all_gather_into_tensor of the per-shard lists behind the four searching streams, option reserve_cus (VERDICT r5 task 2).  The gathered
slot must equal the local buffer, the lists must equal the oracle's, and the line must carry the with / without comparison."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.rstrip("\n").splitlines()[-1])


@pytest.mark.parametrize("path,every", [("rccl", 1), ("rccl", 4), ("c10d", 1), ("c10d-async", 4)])
def test_one_rank_rccl_gather_equals_the_local_buffer(path, every):
    line = _bench("--force-collective", "--gather-path", path, "--gather-every", str(every), "--N", "200000", "--C", "100", "--steps", "22",
                  "--warmup", "5", "--cpu-sample", "1024", "--no-other-configs", "--no-host-abi", "--no-recall", "--ab-rounds", "1")
    assert line["n_gpus"] == 1 and line["gather_verified"] is True
    assert line["config"]["backend"].startswith("rccl") and line["config"]["world_size"] == 1
    assert line["config"]["gather_every"] == every and line["config"]["gather_path"] == path
    assert line["cpu_baseline"]["parity_with_gpu_on_sample"] is True      # the buffers the collective path left behind
    det = json.load(open(os.path.join(ROOT, "bench_details_collective.json")))
    ab = det["collective_1rank"]
    assert ab["gather_verified"] is True and ab["with_collective_qps"] > 0 and ab["without_qps"] > 0
