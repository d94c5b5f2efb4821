"""GPU-box helper: ONE query per call through the host-buffer ABI -- ivfadc_search (freddy.c:174-393) on the 3 M-row bench
index and pq_search (freddy.c:28-152) on the 1 M-row table: wall-clock per call.   python tools/single_query.py [n_calls]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = ib.make_corpus(3_000_000, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
idx = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
qs = x[12345:12346].cpu().numpy()
for _ in range(10):
    idx.search(qs, 5, 10, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
t0 = time.perf_counter()
for _ in range(n):
    idx.search(qs, 5, 10, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
print(f"ivfadc_search, one query: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per call", flush=True)
idx.close()
xp = ib.make_corpus(1_000_000, d=300, seed=11, device=dev)
pt = ib.build_pq_index(xp, m=12, K=1024, train_size=100000, iters=6, seed=1)
pidx = gpu.PQIndex(pt["codebook"], pt["ids"], pt["codes"], device=0)
qp = xp[777:778].cpu().numpy()
for _ in range(10):
    pidx.search(qp, 5, sentinel=100.0)
t0 = time.perf_counter()
for _ in range(n):
    pidx.search(qp, 5, sentinel=100.0)
print(f"pq_search, one query: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per call", flush=True)
