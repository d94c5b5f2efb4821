"""-m gpu: randomised soak of the IVFADC scans against the oracle with a FIXED seed budget (the long version
is tools/soak_fused.py): random shapes -- K from 4 to 1024, 1 to 40 cells incl. empty ones, 50 to 30 000
rows, 1 to 5000 distinct code rows (i.e. from "every distance equal" to "all different"), 1 to 700 queries
-- both cell-grouped scans and the generic kernels, both found rules."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from freddy_amd import gpu as g
    g.load()
    return g


@pytest.mark.parametrize("seed", list(range(12)))
def test_soak_random_index(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    d, m = 300, 12
    K = int(rng.choice([4, 16, 64, 256, 1024]))
    C = int(rng.choice([1, 2, 5, 9, 40]))
    N = int(rng.choice([50, 700, 9000, 30000]))
    coarse = rng.standard_normal((C, d)).astype(np.float32)
    codebook = (rng.standard_normal((m, K, 25)) * 0.3).astype(np.float32)
    cell = rng.integers(0, C, size=N) if seed % 3 else np.zeros(N, np.int64)
    if C > 2:
        cell[cell == 1] = 0          # an empty list
    order = np.argsort(cell, kind="stable")
    ids = (np.arange(N) * 3 + 7).astype(np.int32)[order]
    n_distinct = int(rng.choice([1, 3, 50, 5000]))
    pool = rng.integers(0, K, size=(n_distinct, m)).astype(np.int16)
    codes = pool[rng.integers(0, n_distinct, size=N)][order]
    list_off = np.zeros(C + 1, np.int32)
    list_off[1:] = np.cumsum(np.bincount(cell, minlength=C))
    ids_sorted = np.concatenate([np.sort(ids[list_off[c]:list_off[c + 1]]) for c in range(C)]).astype(np.int32)
    ot = oracle.ivf_table(coarse, codebook, list_off, ids_sorted, codes)
    idx = gpu.IVFIndex(coarse, codebook, list_off, ids_sorted, codes)
    Q = int(rng.choice([1, 40, 300, 700]))
    qs = (coarse[rng.integers(0, C, size=Q)] + 0.2 * rng.standard_normal((Q, d))).astype(np.float32)
    for fused, variant in ((1, 5), (1, 3), (0, 5)):
        idx.set_option("fused", fused)
        idx.set_option("fused_kernel", variant)
        for k, W in [(1, 1), (5, min(3, C)), (32, min(C, 12))]:
            for rule, sent in [(0, 1000.0), (1, 100.0)]:
                gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
                exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
                util.assert_same_lists(gi, gd, exp, f"seed={seed} fused={fused} kernel={variant} K={K} C={C} N={N} Q={Q} "
                                                    f"k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("seed", list(range(10)))
def test_soak_random_pq_table(gpu, oracle, seed):
    """Batches over random flat PQ tables (tools/soak_pq.py with a fixed seed budget): sizes around the pseudo-list
    boundaries, duplicate-heavy code pools, both paths, subsets."""
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
    if N >= 9000:
        sub = rng.choice(ids, size=int(rng.choice([4200, 6000])), replace=False).astype(np.int32)
        sub = np.concatenate([sub, sub[:30], np.array([2, 4], np.int32)])
        exp = oracle.pq_search_in_batch(ot, qs, 5, sub, use_target_lists=True)
        for mode in (-1, 0):
            idx.set_option("pq_fused", mode)
            gi, gd = idx.search(qs, 5, sentinel=1000.0, subset_ids=sub)
            util.assert_same_lists(gi, gd, exp, f"seed={seed} subset K={K} N={N} Q={Q} pq_fused={mode}")
    assert idx.bound_violations() == 0
    idx.close()
