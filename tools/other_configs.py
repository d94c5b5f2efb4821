"""GPU-box helper: wall-clock of BASELINE configs[1] (PQ search 1M) and configs[3] (kNN-join
5000 x 100000) through the synchronous host-buffer ABI (PCIe + host work included), next to the
CPU oracle on the same inputs.  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
from oracle.oracle import Oracle

dev = torch.device("cuda", 0)
o = Oracle()
out = {}
cores = os.cpu_count() or 1


def timeit(f, n=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r


# ---- config 2: PQ search 1M x 300, m=12, K=1024, k=5 ---------------------------------------
N = 1_000_000
x = ib.make_corpus(N, seed=11, device=dev)
t = ib.build_pq_index(x, m=12, K=1024, train_size=100000, iters=6, seed=1)
idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
ot = o.pq_table(t["codebook"], t["ids"], t["codes"])
qs = x[torch.arange(0, N, N // 64, device=dev)[:64]].cpu().numpy()
exact = ib.exact_topk(x, torch.from_numpy(qs).to(dev), 5)
for Q in (1, 64):
    dt, (gi, gd) = timeit(lambda: idx.search(qs[:Q], 5, sentinel=100.0))
    out[f"cfg2_pq_search_Q{Q}"] = {"gpu_ms_per_call": round(dt * 1e3, 3), "gpu_qps": round(Q / dt, 1),
                                   "algorithmic_GBps": round(Q * N * 28 / dt / 1e9, 1)}
out["cfg2_recall_at_5"] = round(ib.recall_at_k(gi, exact), 4)
t0 = time.perf_counter()
exp = np.stack([o.pq_search(ot, q, 5) for q in qs[:16]])
out["cfg2_cpu_oracle_qps_1core"] = round(16 / (time.perf_counter() - t0), 2)
targets = np.random.default_rng(1).choice(np.arange(1, N + 1), 100000, replace=False).astype(np.int32)
qs5k = x[torch.from_numpy(np.random.default_rng(2).choice(N, 5000, replace=False)).to(dev)].cpu().numpy()
dt, _ = timeit(lambda: idx.search(qs5k, 5, sentinel=1000.0, subset_ids=targets), n=3)
out["cfg4_baseline_pq_search_in_batch_5000x100000"] = {"gpu_s_per_call": round(dt, 4)}
idx.close(); del x

# ---- config 4: knn_join 5000 x 100000, k=5, alpha=100, pvf=20, method 2 --------------------
x = ib.make_corpus(N, seed=5, device=dev)
t = ib.build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=6, seed=3)
idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
ot = o.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
rng = np.random.default_rng(4)
qid = rng.choice(np.arange(1, N + 1), 5000, replace=False)
targets = rng.choice(np.arange(1, N + 1), 100000, replace=False).astype(np.int32)
qs = t["vectors"][qid - 1]
tx = torch.from_numpy(t["vectors"][targets - 1]).to(dev)
ex_local = ib.exact_topk(tx, torch.from_numpy(qs).to(dev), 5)
ex = targets[ex_local - 1]
for method in (0, 1, 2):
    dt, (gi, gd, it) = timeit(lambda: idx.knn_join(qs, 5, targets, 100, 20, method), n=3)
    t0 = time.perf_counter()
    exp, _ = o.ivpq_search_in(ot, qs, 5, targets, 100, 20, method)
    cdt = time.perf_counter() - t0
    out[f"cfg4_knn_join_method{method}"] = {"gpu_s_per_call": round(dt, 4), "gpu_queries_per_s": round(5000 / dt, 1),
                                            "cpu_oracle_s_1core": round(cdt, 3), "iterations": it,
                                            "precision_vs_exact": round(ib.recall_at_k(gi, ex), 4),
                                            "parity": bool(np.array_equal(gi, exp["id"]))}
out["host_cores"] = cores
print(json.dumps(out, indent=1))
