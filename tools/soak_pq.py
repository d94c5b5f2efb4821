"""GPU-box helper: randomised soak of the batches over the flat PQ table (cell-grouped scan over pseudo-lists)
against the oracle: random table sizes around the pseudo-list boundaries, codebook sizes, duplicate-heavy code pools (many
equal distances), batch sizes, k, subsets; both paths (pq_fused on / off).  usage: python tools/soak_pq.py [seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: F401
from freddy_amd import gpu
from oracle.oracle import Oracle
import util

oracle = Oracle()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
t0 = time.time()
n = 0
for seed in range(seeds):
    rng = np.random.default_rng(5000 + seed)
    d, m = 300, 12
    K = int(rng.choice([16, 64, 256, 1024]))
    N = int(rng.choice([60, 4095, 4096, 4097, 9000, 40000, 70000]))
    codebook = (rng.standard_normal((m, K, 25)) * 0.3).astype(np.float32)
    ids = (np.arange(N) * 2 + 5).astype(np.int32)
    n_distinct = int(rng.choice([1, 7, 300, 100000]))
    pool = rng.integers(0, K, size=(n_distinct, m)).astype(np.int16)
    codes = pool[rng.integers(0, n_distinct, size=N)]
    ot = oracle.pq_table(codebook, ids, codes)
    idx = gpu.PQIndex(codebook, ids, codes)
    Q = int(rng.choice([16, 17, 48, 130]))
    qs = (0.5 * rng.standard_normal((Q, d))).astype(np.float32)
    if seed % 4 == 0:
        qs[0] *= np.float32(40.0)      # beyond the sentinel 100.0
    for k in (1, 5, 32):
        exp = np.stack([oracle.pq_search(ot, q, k) for q in qs])
        for mode in (1, 0):
            idx.set_option("pq_fused", mode)
            gi, gd = idx.search(qs, k, sentinel=100.0)
            util.assert_same_lists(gi, gd, exp, f"seed={seed} K={K} N={N} Q={Q} k={k} pq_fused={mode}")
            n += 1
    if N >= 9000:
        sub = rng.choice(ids, size=int(rng.choice([4200, 6000])), replace=False).astype(np.int32)
        sub = np.concatenate([sub, sub[:30], np.array([2, 4], np.int32)])
        exp = oracle.pq_search_in_batch(ot, qs, 5, sub, use_target_lists=True)
        for mode in (-1, 0):
            idx.set_option("pq_fused", mode)
            gi, gd = idx.search(qs, 5, sentinel=1000.0, subset_ids=sub)
            util.assert_same_lists(gi, gd, exp, f"seed={seed} subset K={K} N={N} Q={Q} pq_fused={mode}")
            n += 1
    assert idx.bound_violations() == 0
    idx.close()
print(f"pq soak ok: {n} searches over {seeds} random tables in {time.time() - t0:.1f}s")
