#!/usr/bin/env python3
"""GPU-box helper: randomised comparison of the IVFADC search on index shapes OTHER than m = 12 / S = 25 (multi.h's cell-grouped
exact scan, the generic kernels behind it) and on the filter + refine scan with the running bound, against the CPU oracle.
usage: python tools/soak_shapes.py [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib
from oracle.oracle import Oracle
import util

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
oracle = Oracle()
SHAPES = [(25, 5, 256), (300, 6, 256), (300, 10, 64), (300, 15, 128), (300, 30, 32), (300, 4, 512), (300, 20, 16), (300, 12, 256), (300, 12, 1024), (50, 10, 256), (64, 8, 128)]
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    d, m, K = SHAPES[int(rng.integers(0, len(SHAPES)))]
    N = int(rng.choice([900, 5000, 30000, 70000]))
    C = int(rng.choice([1, 3, 16, 40, 200]))
    C = min(C, max(1, N // 20))
    torch.manual_seed(seed)
    x = ib.make_corpus(N, d=d, seed=seed, n_clusters=60, latent=min(10, d), dup_frac=0.02, device="cpu")
    t = ib.build_ivf_index(x, C=C, m=m, K=K, train_size=min(N, 4000), iters=3, seed=seed)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    Q = int(rng.choice([1, 7, 40, 300, 900]))
    qs = x.numpy()[rng.integers(0, N, size=Q)].astype(np.float32) * np.float32(rng.choice([1.0, 1.0, 1.02]))
    for (W, k, rule, sent) in [(int(rng.choice([1, 2, 5])), int(rng.choice([1, 5, 20, 32])), 0, 1000.0), (1, 5, 2, 100.0), (3, 10, 1, 100.0)]:
        W = min(W, C)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule, n_threads=8)
        for fused, rb in ((-1, 1), (1, 1), (1, 0), (0, 1)):
            idx.set_option("fused", fused); idx.set_option("running_bound", rb)
            gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"seed={seed} d={d} m={m} K={K} N={N} C={C} Q={Q} W={W} k={k} rule={rule} fused={fused} rb={rb}")
    assert idx.bound_violations() == 0
    idx.close()
    print(f"seed {seed}: d={d} m={m} K={K} N={N} C={C} Q={Q} ok ({time.time() - t0:.0f} s)", flush=True)
print("soak_shapes ok")
