"""GPU-box helper: long randomised soak of the fused IVFADC kernels against the oracle (the shapes of
tests/test_gpu_parity.py::test_ivfadc_randomised_small_indexes, many more seeds, both cell-grouped scans; the fixed-seed version runs in tests/test_gpu_soak.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: F401
from freddy_amd import gpu
from oracle.oracle import Oracle
import util

oracle = Oracle()
os.environ["FREDDY_GPU_FUSED"] = "1"
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.time()
n = 0
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    d, m = 300, 12
    K = int(rng.choice([4, 16, 64, 256, 1024]))
    C = int(rng.choice([1, 2, 5, 9, 40]))
    N = int(rng.choice([50, 700, 9000, 30000]))
    coarse = rng.standard_normal((C, d)).astype(np.float32)
    codebook = (rng.standard_normal((m, K, 25)) * 0.3).astype(np.float32)
    cell = rng.integers(0, C, size=N) if seed % 3 else np.zeros(N, np.int64)
    if C > 2:
        cell[cell == 1] = 0
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
    # (variant, codes_u8): K <= 256 has three scans behind variant 5 -- 1: the whole-entry-slab kernel (fused8.h, the default),
    # 2: the six-phase kernel's one-byte instantiation, 0: the int16 layout
    for variant, u8 in [(5, 1), (5, 2), (5, 0), (4, 1), (3, 1)]:
        if u8 != 1 and K > 256:
            continue
        idx.set_option("fused_kernel", variant)
        idx.set_option("codes_u8", u8)
        for k, W in [(1, 1), (5, min(3, C)), (32, min(C, 12))]:
            for rule, sent in [(0, 1000.0), (1, 100.0)]:
                gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
                exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
                util.assert_same_lists(gi, gd, exp, f"seed={seed} variant={variant} codes_u8={u8} K={K} C={C} N={N} Q={Q} k={k} W={W} rule={rule}")
                n += 1
    idx.close()
print(f"soak ok: {n} searches over {seeds} random indexes in {time.time() - t0:.1f}s")
