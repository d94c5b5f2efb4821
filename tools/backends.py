#!/usr/bin/env python3
"""1 / 2 / 4 BACKENDS (processes) on one GPU at config-3 size (3 M x 300, C = 1000, m = 12, K = 1024, nprobe 10, k = 5): every
process pins its own copy of the index and makes 1024-query host-buffer calls (freddy_gpu_ivfadc_search) for a few seconds,
all at the same time (tests/backend_worker.py; the test of the same situation is tests/test_gpu_backends.py).
Prints per-process and aggregate queries/s -> profiles/r06_backends.txt.   usage: tools/backends.py [seconds] [N] [configs]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
rng = np.random.default_rng(7)
qid = np.sort(rng.choice(np.arange(1, N + 1), size=2048, replace=False))
qs = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
idx = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
ei, ed = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)   # (this path against the oracle: tests/test_gpu_fullsize.py)
index_mb = idx.nbytes / 1e6
idx.close()
del x
torch.cuda.empty_cache()
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "tables.npz")
np.savez(path, coarse=np.asarray(tab["coarse"]), codebook=np.asarray(tab["codebook"]), list_off=np.asarray(tab["list_off"]), ids=np.asarray(tab["ids"]),
         codes=np.asarray(tab["codes"]), queries=qs, k=5, W=10, exp_ids=ei, exp_dist=ed, shapes=np.array([1024], np.int32))
print(f"# backends on one MI355X: {N} x 300, C = 1000, K = 1024, nprobe 10, k = 5; every process pins its own copy ({index_mb:.0f} MB) and makes")
print(f"# 1024-query host-buffer calls for {seconds:.0f} s; lists compared bit for bit with the single-process lists")
# (processes, scan_share, GPU_MAX_HW_QUEUES, extra environment); None = not set: the LIBRARY decides (core.hip: registry of live backends
# -> two queues when others are alive, six when alone; scan_share auto = 2 while another backend is searching)
CONFIGS = [(1, None, None, {}), (2, None, None, {}), (4, None, None, {}), (8, None, None, {}),
           (1, 1, 6, {}), (2, 2, 2, {}), (4, 2, 2, {}), (4, 1, 6, {"FREDDY_GPU_REGISTRY": "0"})]
if len(sys.argv) > 3:   # e.g. "4:2:2,8:0:0,4:0:6:FREDDY_GPU_LANE0_OWN=1" = processes : scan_share : GPU_MAX_HW_QUEUES [: NAME=value ...] (0 = not set)
    CONFIGS = []
    for c in sys.argv[3].split(","):
        f = c.split(":")
        CONFIGS.append((int(f[0]), int(f[1]) or None, int(f[2]) or None, dict(kv.split("=", 1) for kv in f[3:])))
base_env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FREDDY_GPU_SCAN_SHARE")}
for P, share, hwq, extra in CONFIGS:
    env = dict(base_env, **extra)
    if share:
        env["FREDDY_GPU_SCAN_SHARE"] = str(share)
    if hwq:
        env["GPU_MAX_HW_QUEUES"] = str(hwq)
    with tempfile.TemporaryDirectory() as sync:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "backend_worker.py"), path, str(r), str(P), str(seconds), sync, "batch"],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(P)]
        outs = []
        for p in procs:
            so, se = p.communicate(timeout=600)
            outs.append(json.loads(so.strip().splitlines()[-1]) if p.returncode == 0 else {"error": se[-300:]})
    if any("error" in o for o in outs):
        print(P, "processes: FAILED", outs)
        continue
    per = [o["queries_per_s"] for o in outs]
    what = f"scan_share {share or 'auto'}, GPU_MAX_HW_QUEUES {hwq or 'by the library'}" + (f", {extra}" if extra else "")
    print(f"{P} process(es), {what}: aggregate {sum(per) / 1e6:6.2f} M queries/s   per process {[round(v / 1e6, 2) for v in per]} M   "
          f"hw queues {[o.get('hw_queues') for o in outs]}   mismatching calls {sum(o['mismatches'] for o in outs)}   "
          f"bracket violations {sum(o['bound_violations'] for o in outs)}   pin {outs[0]['pin_seconds']} s", flush=True)
os.remove(path); os.rmdir(d)
