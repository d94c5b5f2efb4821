"""GPU-box helper: wall-clock latency of ONE call through the host-buffer ABI (H2D, kernels, D2H, sync) for small batches of
the bench index: ivfadc_search (the reference's single-query SRF, freddy.c:174-393) and batches of 4 .. 256 queries."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 3000000))
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
idx = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
for Q in (1, 4, 16, 26, 64, 256):
    qids = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False))
    qs = x[torch.from_numpy(qids - 1).to(dev)].cpu().numpy()
    for _ in range(5):
        idx.search(qs, 5, 10, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        idx.search(qs, 5, 10, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
    dt = (time.perf_counter() - t0) / n
    print(f"ivfadc_search Q={Q:4d}: {dt * 1e3:.3f} ms per call, {Q / dt:,.0f} queries/s")
