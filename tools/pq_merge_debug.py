"""GPU-box helper: FREDDY_GPU_MERGE_ABLATE=16 for a batch over the flat PQ table (1 M rows, as bench.py --config pq):
per query, how many of the kept lower bounds qualify for the exact stage, whether all of them do (revisit path), E, T."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
N, Q = int(os.environ.get("N", 1000000)), 64
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, d=300, seed=11, device=dev)
tab = ib.build_pq_index(x, m=12, K=1024, train_size=100000, iters=6, seed=1)
idx = gpu.PQIndex(tab["codebook"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
qids = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False))
qs = x[torch.from_numpy(qids - 1).to(dev)].cpu().numpy()
idx.set_option("merge_ablate", 16)
gi, gd = idx.search(qs, 5, sentinel=100.0)
print("qualifying of the kept keys: mean %.1f max %d; all-qualify (revisit) queries: %d of %d; E mean %.2e; T mean %.4f; Lth d_lo mean %.4f" % (
    gd[:, 0].mean(), gd[:, 0].max(), int(gd[:, 1].sum()), Q, gd[:, 2].mean(), gd[:, 3].mean(), gd[:, 4].mean()))
