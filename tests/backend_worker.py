"""One "backend": a process of its own that pins the tables of an .npz file and searches for a few seconds beside its siblings
(tests/test_gpu_backends.py, tools/backends.py).  A PostgreSQL server is N forked backends, each with its own HIP context, its own
pinned copy and its own persistent scans: this is that situation without PostgreSQL.
usage: backend_worker.py <tables.npz> <rank> <n_procs> <seconds> <sync_dir> [mode]
Prints ONE JSON line: calls / queries per call shape, mismatches against the expected lists of the file, the kernels it launched."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]

import numpy as np

from freddy_amd import gpu


def main():
    path, rank, n_procs, seconds, sync_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
    mode = sys.argv[6] if len(sys.argv) > 6 else "mixed"
    z = np.load(path)
    t0 = time.time()
    ivf = gpu.IVFIndex(z["coarse"], z["codebook"], z["list_off"], z["ids"], z["codes"], device=0)
    pq = gpu.PQIndex(z["pq_codebook"], z["pq_ids"], z["pq_codes"], device=0) if "pq_codebook" in z.files else None
    pin_s = time.time() - t0
    qs = z["queries"]
    k, W = int(z["k"]), int(z["W"])
    exp_i, exp_d = z["exp_ids"], z["exp_dist"]
    shapes = [int(s) for s in z["shapes"]]                      # queries per call, taken in turn
    if mode == "batch":
        shapes = [max(shapes)]
    ivf.profile_enable(True)
    # everybody pinned -> go (so that the searches really overlap)
    open(os.path.join(sync_dir, f"ready{rank}"), "w").close()
    t_wait = time.time()
    while not all(os.path.exists(os.path.join(sync_dir, f"ready{r}")) for r in range(n_procs)):
        if time.time() - t_wait > 120:
            print(json.dumps({"rank": rank, "error": "the other backends never became ready"}), flush=True)
            return 2
        time.sleep(0.005)
    out = {"rank": rank, "pin_seconds": round(pin_s, 2), "calls": {}, "queries": 0, "mismatches": 0, "pq_calls": 0, "pq_mismatches": 0}
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        Q = shapes[n % len(shapes)]
        off = (7 * n + 13 * rank) % max(1, qs.shape[0] - Q + 1)   # a different window of the queries every call and every backend
        gi, gd = ivf.search(qs[off:off + Q], k, W, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
        ok = np.array_equal(gi, exp_i[off:off + Q]) and np.array_equal(gd.view(np.uint32), exp_d[off:off + Q].view(np.uint32))
        out["mismatches"] += 0 if ok else 1
        out["calls"][str(Q)] = out["calls"].get(str(Q), 0) + 1
        out["queries"] += Q
        if pq is not None and mode == "mixed" and n % 4 == 0:
            j = (n // 4 + rank) % 64            # (the first 64 queries have pq_search expectations)
            pi, pd = pq.search(qs[j:j + 1], k, sentinel=100.0)
            okp = np.array_equal(pi, z["pq_exp_ids"][j:j + 1]) and np.array_equal(pd.view(np.uint32), z["pq_exp_dist"][j:j + 1].view(np.uint32))
            out["pq_mismatches"] += 0 if okp else 1
            out["pq_calls"] += 1
        n += 1
    out["seconds"] = round(time.time() - t0, 3)
    out["queries_per_s"] = round(out["queries"] / out["seconds"], 1)
    prof = ivf.profile_read()
    out["kernels"] = {name: int(l) for name, (l, ms) in prof.items()}
    out["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")   # (what the library chose for this process unless the environment had it)
    out["bound_violations"] = int(ivf.bound_violations())
    ivf.close()
    if pq is not None:
        pq.close()
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
