"""GPU-box helper: host timeline of the config-4 kNN-join call (lab build: FREDDY_GPU_SO=.../libfreddy_gpu_lab.so FREDDY_GPU_JOIN_TRACE=1)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
N, Q, T = 1_000_000, 5000, 100_000
x = ib.make_corpus(N, d=300, seed=5, device=dev)
t = ib.build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=6, seed=3)
index = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
rng = np.random.default_rng(4)
qid = rng.choice(np.arange(1, N + 1), Q, replace=False)
tg = [rng.choice(np.arange(1, N + 1), T, replace=False).astype(np.int32) for _ in range(3)]
qs = np.ascontiguousarray(t["vectors"][qid - 1])
for i in range(8):
    sys.stderr.write(f"--- call {i}\n"); sys.stderr.flush()
    index.knn_join(qs, 5, tg[i % 3], 100, 20, 2)
# (the variable is read once, at the first call: set before the process starts)
