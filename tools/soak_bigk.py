#!/usr/bin/env python3
"""GPU-box helper: randomised comparison against the CPU oracle of the paths for long lists -- ivfadc_search* / pq_search* with
512 < k <= 4096 (bigk.h: selection in passes of 1024 keys, the replay in closed form, carried lists over several probing rounds)
and the kNN-join's post verification with 1024 < k * pvf <= 8192 (join_query_kernel<16, true>) -- on small tables with many
equal distances.  usage: python tools/soak_bigk.py [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib
from oracle.oracle import Oracle
import util

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
oracle = Oracle()
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.choice([3000, 12000, 40000]))
    d, m, K = [(300, 12, 256), (300, 12, 1024), (50, 10, 64), (300, 6, 256)][int(rng.integers(0, 4))]
    C = int(rng.choice([1, 8, 40]))
    torch.manual_seed(seed)
    x = ib.make_corpus(N, d=d, seed=seed, n_clusters=40, latent=min(10, d), dup_frac=float(rng.choice([0.02, 0.3])), device="cpu")
    # ---- IVFADC
    t = dict(ib.build_ivf_index(x, C=C, m=m, K=K, train_size=min(N, 4000), iters=3, seed=seed))
    codes = t["codes"].copy()
    lo = t["list_off"]
    for c in range(len(lo) - 1):   # runs of equal code rows: equal distances around the k-th place
        n = min(int(rng.integers(0, 700)), int(lo[c + 1] - lo[c]))
        if n: codes[lo[c]:lo[c] + n] = codes[lo[c]]
    t["codes"] = codes
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    Q = int(rng.choice([1, 3, 9]))
    qs = x.numpy()[rng.integers(0, N, size=Q)].astype(np.float32)
    for _ in range(3):
        k = int(rng.choice([513, 700, 1024, 1025, 1500, 2048, 3000, 4096]))
        W = min(int(rng.choice([1, 2, 5])), C)
        rule, sent = [(0, 1000.0), (1, 1000.0), (0, float(rng.choice([0.5, 2.0, 8.0]))), (2, 100.0)][int(rng.integers(0, 4))]
        if rule == 2: W = 1
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule, n_threads=8)
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"ivfadc seed={seed} d={d} m={m} K={K} N={N} C={C} Q={Q} W={W} k={k} rule={rule} sent={sent}")
    idx.close()
    # ---- PQ
    tp = ib.build_pq_index(x, m=m, K=K, train_size=min(N, 4000), iters=3, seed=seed)
    pcodes = tp["codes"].copy()
    a = int(rng.integers(0, N // 2)); pcodes[a:a + int(rng.integers(1, 1500))] = pcodes[a]
    op = oracle.pq_table(tp["codebook"], tp["ids"], pcodes)
    pidx = gpu.PQIndex(tp["codebook"], tp["ids"], pcodes)
    k = int(rng.choice([513, 900, 2048, 4096]))
    gi, gd = pidx.search(qs, k, sentinel=100.0)
    exp = np.stack([oracle.pq_search(op, q, k) for q in qs])
    util.assert_same_lists(gi, gd, exp, f"pq seed={seed} d={d} m={m} K={K} N={N} Q={Q} k={k}")
    targets = rng.choice(np.arange(1, N + 1), size=min(N, int(rng.choice([k - 7, 2 * k, N // 2]))), replace=False).astype(np.int32)
    gi, gd = pidx.search(qs, k, sentinel=1000.0, subset_ids=targets)
    exp = oracle.pq_search_in_batch(op, qs, k, targets, use_target_lists=True)
    util.assert_same_lists(gi, gd, exp, f"pq_search_in seed={seed} N={N} k={k} targets={len(targets)}")
    pidx.close()
    # ---- kNN-join, post verification of more than 1024 candidates
    if d == 300 and seed % 2 == 0:
        tj = ib.build_ivpq_index(x, m=30, K=32, k_coarse=8, train_size=min(N, 4000), iters=3, seed=seed)
        oj = oracle.ivpq_table(tj["codebook"], tj["coarse"], tj["ids"], tj["coarse_id"], tj["codes"], tj["vectors"], tj["stats"])
        jidx = gpu.IVPQIndex(tj["codebook"], tj["coarse"], tj["ids"], tj["coarse_id"], tj["codes"], tj["vectors"], tj["stats"])
        tg = rng.choice(np.arange(1, N + 1), size=int(N * float(rng.choice([0.2, 0.8]))), replace=False).astype(np.int32)
        jq = x.numpy()[rng.integers(0, N, size=12)].astype(np.float32)
        for _ in range(2):
            kj = int(rng.choice([20, 60, 100, 400]))
            pvf = int(rng.choice([p for p in (20, 50, 100, 400) if 1024 < kj * p <= 8192] or [8192 // kj]))
            alpha = int(rng.choice([3, 30, 200]))
            use_tl = bool(rng.integers(0, 2))
            gi, gd, git = jidx.knn_join(jq, kj, tg, alpha, pvf, 2, use_target_lists=use_tl, confidence=0.8)
            exp, eit = oracle.ivpq_search_in(oj, jq, kj, tg, alpha, pvf, 2, use_target_lists=use_tl, confidence=0.8)
            assert git == eit, (seed, git, eit)
            util.assert_same_lists(gi, gd, exp, f"join seed={seed} N={N} k={kj} pvf={pvf} alpha={alpha} tl={use_tl} targets={len(tg)}")
        jidx.close()
    print(f"seed {seed}: d={d} m={m} K={K} N={N} C={C} Q={Q} ok ({time.time() - t0:.0f} s)", flush=True)
print("soak_bigk ok")
