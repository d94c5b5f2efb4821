"""GPU-box helper: exact brute-force kNN at 3M x 300 through the host-buffer ABI,
per-kernel HIP-event times from the library's own profiler, CPU oracle beside it."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
from oracle.oracle import Oracle

N = int(os.environ.get("EXACT_N", 3_000_000))
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, seed=11, device=dev).cpu().numpy()
ids = np.arange(1, N + 1, dtype=np.int32)
t0 = time.perf_counter()
idx = gpu.VectorIndex(ids, x)
out = {"N": N, "pin_s": round(time.perf_counter() - t0, 2)}
rng = np.random.default_rng(0)
qs = x[rng.choice(N, 1024, replace=False)]
for Q in (1, 8, 64, 1024):
    idx.search(qs[:Q], 5)
    idx.profile_enable(True)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        gi, gs = idx.search(qs[:Q], 5)
    dt = (time.perf_counter() - t0) / n
    prof = idx.profile_read()
    out[f"Q{Q}"] = {"ms_per_call": round(dt * 1e3, 3), "qps": round(Q / dt, 1),
                    "kernels_us": {k: round(v[1] / v[0] * 1e3, 1) for k, v in prof.items()},
                    "algorithmic_GBps": round(Q * N * 1200 / dt / 1e9, 1),
                    "valu_Tlaneops": round(Q * N * 600 / dt / 1e12, 2)}
    idx.profile_enable(False)
if os.environ.get("FAKE") == "1":   # timing experiment: coalesced (lane-linear) row loads, results are garbage
    idx.set_option("exact_fake_coalesced", 1)
    for Q in (1, 64):
        idx.search(qs[:Q], 5)
        idx.profile_enable(True)
        for _ in range(3):
            idx.search(qs[:Q], 5)
        prof = idx.profile_read()
        out[f"FAKE_Q{Q}"] = {k: round(v[1] / v[0] * 1e3, 1) for k, v in prof.items()}
        idx.profile_enable(False)
    idx.set_option("exact_fake_coalesced", 0)
sub = rng.choice(ids, 100000, replace=False).astype(np.int32)
t0 = time.perf_counter(); idx.search(qs[:64], 5, subset_ids=sub); out["subset_100k_Q64_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
o = Oracle()
t0 = time.perf_counter()
exp = o.exact_knn(x, ids, qs[0], 5)
out["cpu_oracle_s_per_query_1core"] = round(time.perf_counter() - t0, 3)
out["parity"] = bool(np.array_equal(exp["id"], gi[0]) and np.array_equal(exp["dist"].view(np.uint32), gs[0].view(np.uint32)))
print(json.dumps(out, indent=1))
