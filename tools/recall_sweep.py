"""GPU-box helper: recall@5 of the HIP IVFADC path on variants of the synthetic corpus."""
import os, sys, time, itertools
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 3000000))
for latent, spread, noise in [(50, 0.35, 0.02), (50, 0.35, 0.01), (50, 0.25, 0.01), (30, 0.35, 0.01), (50, 0.5, 0.01), (50, 0.35, 0.005), (24, 0.35, 0.01)]:
    t0 = time.time()
    x = ib.make_corpus(N, seed=20260101, latent=latent, spread=spread, noise=noise, device=dev)
    tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
    idx = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"])
    rng = np.random.default_rng(7)
    qids = np.sort(rng.choice(np.arange(1, N + 1), size=1024, replace=False)).astype(np.int64)
    q = x[torch.from_numpy(qids - 1).to(dev)].contiguous()
    exact = ib.exact_topk(x, q, 5)
    qs = q.cpu().numpy()
    out = []
    for W in (1, 3, 10):
        gi, gd = idx.search(qs, 5, W)
        out.append((W, round(ib.recall_at_k(gi, exact), 4)))
    lens = np.diff(tab["list_off"])
    print(f"latent={latent} spread={spread} noise={noise}: recall {out} lists min/mean/max {lens.min()}/{lens.mean():.0f}/{lens.max()} ({time.time()-t0:.1f}s)", flush=True)
    idx.close(); del x, tab
