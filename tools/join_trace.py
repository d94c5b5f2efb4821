import os, sys, time
import numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/postgres-word2vec_amd"]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
N = 1_000_000
x = ib.make_corpus(N, seed=5, device=dev)
t = ib.build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=6, seed=3)
idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
rng = np.random.default_rng(4)
qid = rng.choice(np.arange(1, N + 1), 5000, replace=False)
targets = rng.choice(np.arange(1, N + 1), 100000, replace=False).astype(np.int32)
qs = t["vectors"][qid - 1]
idx.knn_join(qs, 5, targets, 100, 20, 2)
t0 = time.perf_counter(); idx.knn_join(qs, 5, targets, 100, 20, 2); print("total", time.perf_counter() - t0)
for name, v in idx.last_track().items():   # the reference's TRACK stage names (freddy_gpu_last_track)
    print("TRACK", name, v)
