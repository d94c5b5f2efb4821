"""Debugging aid: per-query numbers of the filter + refine merge (FREDDY_GPU_MERGE_ABLATE=16) on the bench corpus."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "postgres-word2vec_amd"))
os.environ["FREDDY_GPU_MERGE_ABLATE"] = "16"
import numpy as np, torch
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
x = ib.make_corpus(3000000, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
qids = np.sort(rng.choice(np.arange(1, 3000001), size=1024, replace=False)).astype(np.int64)
q = x[torch.from_numpy(qids - 1).to(dev)].contiguous().cpu().numpy()
ids, dist = index.search(q, 5, 10)
n_in, all_in, E, T, kth = dist[:, 0], dist[:, 1], dist[:, 2], dist[:, 3], dist[:, 4]
print("n_in mean %.2f max %d; all_in %d of %d; E mean %.3g; kth mean %.4f; T-kth mean %.3g" %
      (n_in.mean(), n_in.max(), int(all_in.sum()), len(all_in), E.mean(), kth.mean(), (T - kth).mean()))
print(dist[:5])
