"""GPU-box helper: the reference's OTHER shipped index shape -- index_creation/config/ivfadc_complete_config.json: m = 5
sub-vectors of 5 dimensions (25-d GloVe twitter vectors, 1.19 M rows), K = 256, 32 coarse cells -- which the filter + refine scan
does not cover (it is built for m = 12, S = 25, K <= 1024).  Batches take the cell-grouped exact scan of multi.h (round 5), single
queries and small batches the generic LUT + streaming-scan kernels.  Prints one JSON object: parity with the oracle on a sample,
queries/s for batches of 1 / 100 / 1024 queries (the reference's ivfadc_batch_search call shapes), the kernels' durations.
usage: python tools/other_shape.py [N]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
import torch
from freddy_amd import gpu, index_build as ib
from oracle.oracle import Oracle
import util

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_193_514
d, m, K, C, k, W = 25, 5, 256, 32, 5, 2
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, d=d, seed=31, n_clusters=200, latent=10, device=dev)
t = ib.build_ivf_index(x, C=C, m=m, K=K, train_size=100000, iters=10, seed=2)
idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
rng = np.random.default_rng(5)
qid = rng.choice(N, size=1024, replace=False)
qs = x[torch.from_numpy(qid).to(dev)].cpu().numpy().astype(np.float32)
out = {"shape": {"N": N, "d": d, "m": m, "K": K, "C": C, "nprobe": W, "k": k}, "batches": {}}
oracle = Oracle()
ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
exp = oracle.ivfadc_search_many(ot, qs[:64], k, W, sentinel=1000.0, found_rule=0, n_threads=8)
gi, gd = idx.search(qs[:64], k, W, sentinel=1000.0)
util.assert_same_lists(gi, gd, exp, "other shape")
out["parity_with_oracle_on_64_queries"] = True
for Q in (1, 100, 1024):
    q = qs[:Q]
    for _ in range(3): idx.search(q, k, W, sentinel=1000.0)
    n = 30 if Q > 1 else 200
    t0 = time.perf_counter()
    for _ in range(n): idx.search(q, k, W, sentinel=1000.0)
    dt = (time.perf_counter() - t0) / n
    out["batches"][str(Q)] = {"ms_per_call": round(dt * 1e3, 4), "queries_per_s": round(Q / dt, 1)}
idx.profile_enable(True)
for _ in range(5): idx.search(qs, k, W, sentinel=1000.0)
prof = idx.profile_read()
idx.profile_enable(False)
out["kernels_batch_1024_us"] = {n: round(1e3 * ms / max(l, 1), 2) for n, (l, ms) in prof.items()}
rows_per_query = W * N / C
out["algorithmic_bytes_per_query"] = int(rows_per_query * (m * 2 + 4))
print(json.dumps(out))
