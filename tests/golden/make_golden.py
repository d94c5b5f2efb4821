"""Regenerates tests/golden/freddy_small.npz.

The reference ships no golden vectors and cannot be run here, so these vectors
do NOT pin the oracle to the reference.  They are inputs plus the ORACLE's outputs at the time of
writing: a regression pin for both implementations (a compiler flag that lets an FMA in, a changed tie
rule or summation order shows up as a mismatch against committed data, on the CPU and on the GPU box).
Inputs are stored, not re-derived, so no RNG or library version is part of the contract.

    python tests/golden/make_golden.py        # needs oracle/libfreddy_oracle.so (make -C oracle)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE))]
from oracle.oracle import Oracle  # noqa: E402


def main():
    o = Oracle()
    rng = np.random.default_rng(20260101)
    d, m, K, s, C, N = 300, 12, 16, 25, 8, 360
    base = rng.standard_normal((40, d)).astype(np.float32)
    x = base[rng.integers(0, 40, N)] + 0.05 * rng.standard_normal((N, d)).astype(np.float32)
    x[100:130] = x[70:100]                                            # exact duplicates: distance ties
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x = x.astype(np.float32)
    ids = np.arange(1, N + 1, dtype=np.int32)
    coarse = x[rng.choice(N, C, replace=False)].copy()
    codebook = (0.15 * rng.standard_normal((m, K, s))).astype(np.float32)
    pq_codebook = x[rng.choice(N, K, replace=False)].reshape(K, m, s).transpose(1, 0, 2).copy()
    out = dict(x=x, ids=ids, coarse=coarse, codebook=codebook, pq_codebook=pq_codebook)
    # --- flat PQ
    pq_codes = o.encode_pq(pq_codebook, x)
    pq_t = o.pq_table(pq_codebook, ids, pq_codes)
    qs = x[[3, 77, 101, 250, 300, 359]].copy()
    sub = np.array([5, 17, 17, 72, 102, 300, 361, -4], np.int32)
    out.update(pq_codes=pq_codes, queries=qs, subset=sub,
               pq_search=np.stack([o.pq_search(pq_t, q, 5) for q in qs]),
               pq_search_in=np.stack([o.pq_search_in(pq_t, q, 4, sub) for q in qs]))
    # --- IVFADC
    cell = o.assign_coarse(coarse, x)
    order = np.lexsort((ids, cell))
    res = np.stack([o.vec_minus(x[i], coarse[cell[i]]) for i in range(N)])
    codes = o.encode_pq(codebook, res)
    list_off = np.zeros(C + 1, np.int32)
    list_off[1:] = np.cumsum(np.bincount(cell, minlength=C))
    ivf_ids, ivf_codes = ids[order], codes[order]
    ivf_t = o.ivf_table(coarse, codebook, list_off, ivf_ids, ivf_codes)
    out.update(list_off=list_off, ivf_ids=ivf_ids, ivf_codes=ivf_codes,
               ivfadc_w3_k5=o.ivfadc_search_many(ivf_t, qs, 5, 3, sentinel=1000.0, found_rule=0),
               ivfadc_w1_k7_accepted=o.ivfadc_search_many(ivf_t, qs, 7, 1, sentinel=100.0, found_rule=1),
               ivfadc_w8_k20=o.ivfadc_search_many(ivf_t, qs, 20, 8, sentinel=1000.0, found_rule=0))
    # --- ivpq / kNN-join
    Kc, m2, K2 = 4, 30, 8
    s2 = d // m2
    cb2 = (0.1 * rng.standard_normal((m2, K2, s2))).astype(np.float32)
    cq = x[rng.choice(N, 2 * Kc, replace=False)].reshape(2, Kc, d)[:, :, :d // 2].copy()
    cq[1] = x[rng.choice(N, Kc, replace=False)][:, d // 2:]
    codes2 = o.encode_pq(cb2, x)
    c0 = o.encode_pq(cq[0][None], x[:, :d // 2])[:, 0].astype(np.int32)
    c1 = o.encode_pq(cq[1][None], x[:, d // 2:])[:, 0].astype(np.int32)
    coarse_id = (c0 + Kc * c1).astype(np.int32)
    stats = np.append(np.bincount(coarse_id, minlength=Kc * Kc) / N, N).astype(np.float32)
    ivpq_t = o.ivpq_table(cb2, cq, ids, coarse_id, codes2, x, stats)
    targets = np.concatenate([np.arange(2, N, 3), [9, 9, 700]]).astype(np.int32)
    out.update(ivpq_codebook=cb2, ivpq_coarse=cq, ivpq_codes=codes2, ivpq_coarse_id=coarse_id, ivpq_stats=stats, targets=targets)
    for method in (0, 1, 2):
        exp, it = o.ivpq_search_in(ivpq_t, qs, 5, targets, 3, 4, method)
        out[f"knn_join_m{method}"] = exp
        out[f"knn_join_m{method}_iterations"] = np.int32(it)
    # --- exact kNN and grouping
    out["exact_knn"] = np.stack([o.exact_knn(x, ids, q, 6) for q in qs])
    gi, gg = o.grouping_pq(pq_t, x[[9, 199, 349]], ids[::5])
    out.update(grouping_ids=gi, grouping_group=gg, grouping_input=ids[::5].copy())
    np.savez_compressed(os.path.join(HERE, "freddy_small.npz"), **out)
    print("wrote", os.path.join(HERE, "freddy_small.npz"), os.path.getsize(os.path.join(HERE, "freddy_small.npz")), "bytes")


if __name__ == "__main__":
    main()
