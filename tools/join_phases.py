"""GPU-box helper: where join_query_kernel's time goes -- the config-4 call with method 0 / 2 (post verification on / off),
different alpha (rows scanned) and pvf (rows verified).   python tools/join_phases.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
N, Q, T = 1_000_000, 5000, 100_000
x = ib.make_corpus(N, d=300, seed=5, device=dev)
t = ib.build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=6, seed=3)
index = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
rng = np.random.default_rng(4)
qid = rng.choice(np.arange(1, N + 1), Q, replace=False)
targets = rng.choice(np.arange(1, N + 1), T, replace=False).astype(np.int32)
qs = t["vectors"][qid - 1]
for method, alpha, pvf in [(2, 100, 20), (0, 100, 20), (2, 100, 1), (2, 25, 20), (0, 25, 20), (0, 400, 20), (1, 100, 20)]:
    for _ in range(2):
        index.knn_join(qs, 5, targets, alpha, pvf, method)
    tot = None
    for _ in range(5):
        t0 = time.perf_counter()
        index.knn_join(qs, 5, targets, alpha, pvf, method)
        dt = time.perf_counter() - t0
        tr = index.last_track()
        tr["wall"] = dt
        tot = tr if tot is None else {k: tot[k] + tr[k] for k in tr}
    tot = {k: v / 5 for k, v in tot.items()}
    print(f"method={method} alpha={alpha} pvf={pvf}: call {tot['wall']*1e3:.3f} ms, join kernels {tot['join_kernel_time']*1e3:.3f} ms, rows {tot['candidate_rows']:.0f}, rounds {tot['iterations']:.0f}, "
          f"precomp {tot['precomputation_time']*1e3:.3f} traverse {tot['determine_coarse_quantization_time']*1e3:.3f}", flush=True)
