"""GPU-box helper: randomised soak of round 3's new paths against the oracle:
  * the host-buffer pipeline of freddy_gpu_ivfadc_search (random batch sizes, sub-batch sizes, lanes, W, k, found rules,
    tiny cells that force extra probing rounds, pinned and pageable query buffers, single queries, replicated handles);
  * the kNN-join with the traversal on the device (random multi-index sizes incl. duplicate centroids, targets, k, alpha,
    pvf, methods, confidences, target lists on / off; every call also with the host heap);
  * single-query pq_search through the pinned direct I/O;
  * the item-wise scan of thin cells forced on / off, the combined coarse + table launch on / off.
usage: python tools/soak_round3.py [seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: F401
from freddy_amd import gpu, index_build as ib
from oracle.oracle import Oracle
import util

oracle = Oracle()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.time()
n_ivf = n_join = n_pq = 0
for seed in range(seeds):
    rng = np.random.default_rng(9000 + seed)
    # ---- IVFADC host pipeline
    N = int(rng.choice([700, 5000, 30000]))
    C = int(rng.choice([8, 40, 150])) if N < 1000 else int(rng.choice([16, 64]))
    K = int(rng.choice([64, 256, 1024]))
    x = ib.make_corpus(N, seed=100 + seed, n_clusters=50, dup_frac=0.02, device="cpu")
    t = ib.build_ivf_index(x, C=C, m=12, K=K, train_size=min(N, 3000), iters=3, seed=seed)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    devices = [0, 0] if seed % 5 == 0 else None
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"], devices=devices)
    Q = int(rng.choice([1, 3, 9, 65, 700, 2100]))
    qs = x.numpy()[rng.integers(0, N, size=Q)].astype(np.float32) * np.float32(rng.choice([1.0, 1.0, 1.03]))
    idx.set_option("pipeline_batch", int(rng.choice([16, 100, 256, 1024])))
    idx.set_option("pipeline_lanes", int(rng.choice([1, 2, 4])))
    idx.set_option("sparse_items", int(rng.choice([2, 0, -1, -3, -16])))   # (negative: the item-wise scan forced for cells of up to that many items)
    for (W, k, rule, sent) in [(int(rng.choice([1, 2, 5])), int(rng.choice([1, 5, 20])), 0, 1000.0), (1, 5, 2, 100.0), (3, 10, 1, 100.0)]:
        W = min(W, C)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule, n_threads=8)
        buf = None
        if seed % 3 == 0:
            buf = gpu.PinnedBuffer(qs.shape); buf.array[:] = qs
        gi, gd = idx.search(buf.array if buf else qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"seed={seed} ivf N={N} C={C} K={K} Q={Q} W={W} k={k} rule={rule}")
        if buf: buf.close()
        n_ivf += 1
    assert idx.bound_violations() == 0
    idx.close()
    # ---- kNN-join
    Nj = int(rng.choice([3000, 20000]))
    kc = int(rng.choice([4, 8, 32]))
    xj = ib.make_corpus(Nj, seed=200 + seed, n_clusters=40, dup_frac=0.02, device="cpu")
    tj = dict(ib.build_ivpq_index(xj, m=30, K=32, k_coarse=kc, train_size=min(Nj, 3000), iters=3, seed=seed))
    if seed % 4 == 1:   # duplicate multi-index centroids: equal keys among the nearest cells
        co = tj["coarse"].copy(); co[0, 1] = co[0, 0]; co[1, kc - 1] = co[1, 0]; tj["coarse"] = co
    otj = oracle.ivpq_table(tj["codebook"], tj["coarse"], tj["ids"], tj["coarse_id"], tj["codes"], tj["vectors"], tj["stats"])
    jdx = gpu.IVPQIndex(tj["codebook"], tj["coarse"], tj["ids"], tj["coarse_id"], tj["codes"], tj["vectors"], tj["stats"])
    Qj = int(rng.choice([1, 40, 300]))
    qj = xj.numpy()[rng.integers(0, Nj, size=Qj)].astype(np.float32)
    T = int(rng.choice([5, 200, Nj // 4]))
    targets = rng.choice(np.arange(1, Nj + 1), size=T, replace=False).astype(np.int32)
    for _ in range(3):
        k = int(rng.choice([1, 5, 12])); alpha = int(rng.choice([1, 3, 50, 1000])); pvf = int(rng.choice([1, 4, 20]))
        method = int(rng.choice([0, 1, 2])); conf = float(rng.choice([0.05, 0.5, 0.8, 0.99])); tl = bool(rng.integers(0, 2))
        exp, eit = oracle.ivpq_search_in(otj, qj, k, targets, alpha, pvf, method, use_target_lists=tl, confidence=conf)
        for host in (0, 1):
            jdx.set_option("join_host_traversal", host)
            gi, gd, git = jdx.knn_join(qj, k, targets, alpha, pvf, method, use_target_lists=tl, confidence=conf)
            assert git == eit, (seed, git, eit)
            util.assert_same_lists(gi, gd, exp, f"seed={seed} join kc={kc} Q={Qj} T={T} k={k} alpha={alpha} pvf={pvf} method={method} conf={conf} tl={tl} host={host}")
            n_join += 1
    jdx.close()
    # ---- single-query / small-batch pq_search through the pinned direct I/O
    Np = int(rng.choice([60, 4097, 30000]))
    Kp = int(rng.choice([16, 256, 1024]))
    codebook = (rng.standard_normal((12, Kp, 25)) * 0.3).astype(np.float32)
    ids = (np.arange(Np) * 3 + 2).astype(np.int32)
    codes = rng.integers(0, Kp, size=(Np, 12)).astype(np.int16)
    otp = oracle.pq_table(codebook, ids, codes)
    pidx = gpu.PQIndex(codebook, ids, codes)
    for Qp in (1, 2, 8, 9):
        qp = (0.5 * rng.standard_normal((Qp, 300))).astype(np.float32)
        for k in (1, 5):
            exp = np.stack([oracle.pq_search(otp, q, k) for q in qp])
            gi, gd = pidx.search(qp, k, sentinel=100.0)
            util.assert_same_lists(gi, gd, exp, f"seed={seed} pq N={Np} K={Kp} Q={Qp} k={k}")
            n_pq += 1
    pidx.close()
print(f"round-3 soak ok: {n_ivf} pipeline searches, {n_join} joins, {n_pq} small pq searches over {seeds} seeds in {time.time() - t0:.1f}s")
