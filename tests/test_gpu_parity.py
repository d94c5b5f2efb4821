"""-m gpu parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded
tables.  Bar: bit-exact (id, rank) AND bit-exact binary32 distance for every ADC result
(the ADC distance is an order-fixed fp32 sum, so there is no tolerance to grant)."""
import functools
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from freddy_amd import gpu as g
    g.load()
    return g


@pytest.mark.parametrize("K", [256, 1024])
def test_pq_search_matches_oracle(gpu, oracle, K):
    N = 20000
    t = util.pq_tables(N=N, K=K)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 16)
    for k in (1, 5, 17):
        gi, gd = idx.search(qs, k, sentinel=100.0)
        exp = np.stack([oracle.pq_search(ot, q, k) for q in qs])
        util.assert_same_lists(gi, gd, exp, f"pq_search K={K} k={k}")
    idx.close()


@pytest.mark.parametrize("K", [256, 1024])
def test_pq_single_query_one_launch(gpu, oracle, K):
    """A single pq_search query through the host-buffer call is ONE launch (one.h: table slices -> grid barrier -> scan ->
    last-arriver merge; option one_launch).  Consecutive calls with different queries on one handle (the second call's table
    must not be served from a cache line of the first), duplicates (equal distances), a sentinel that bites, k up to the
    selection width, a subset; each list equal to the oracle's and to the three-launch path's."""
    N = 60000
    t = util.pq_tables(N=N, K=K)
    codes = t["codes"].copy()
    codes[1000:1300] = codes[999]   # 300 rows with the same code row: equal distances, order decided by the scan position
    ot = oracle.pq_table(t["codebook"], t["ids"], codes)
    idx = gpu.PQIndex(t["codebook"], t["ids"], codes)
    _, qs = util.queries_from_corpus(N, 48, seed=5)
    qs[3] *= np.float32(20.0)   # farther than the sentinel from everything
    idx.profile_enable(True)
    gi, gd = idx.search(qs[:1], 5, sentinel=100.0)
    prof = idx.profile_read()
    idx.profile_enable(False)
    assert "pq_one" in prof, f"the one-launch kernel did not run: {sorted(prof)}"
    for k in (1, 5, 32):
        for i, q in enumerate(qs):
            exp = oracle.pq_search(ot, q, k)[None]
            gi, gd = idx.search(q[None], k, sentinel=100.0)
            util.assert_same_lists(gi, gd, exp, f"one launch K={K} k={k} query {i}")
    rng = np.random.default_rng(9)
    targets = rng.choice(np.arange(1, N + 1), size=9000, replace=False).astype(np.int32)
    for i, q in enumerate(qs[:8]):
        exp = oracle.pq_search_in_batch(ot, q[None], 5, targets, use_target_lists=True)
        gi, gd = idx.search(q[None], 5, sentinel=1000.0, subset_ids=targets)
        util.assert_same_lists(gi, gd, exp, f"one launch, subset, query {i}")
    idx.set_option("one_launch", 0)
    for i, q in enumerate(qs[:8]):
        gi, gd = idx.search(q[None], 5, sentinel=100.0)
        util.assert_same_lists(gi, gd, oracle.pq_search(ot, q, 5)[None], f"three launches, query {i}")
    idx.close()


def test_pq_single_query_one_launch_under_load(gpu, oracle):
    """The one-launch kernel's hand-offs (table slices -> every workgroup, lists -> last arriver) while other streams keep
    the chip unevenly busy (matrix products of several sizes enqueued ahead on two torch streams): the grid's workgroups
    start at different times and on whatever CUs come free; every list equals the oracle's."""
    import torch
    N = 60000
    t = util.pq_tables(N=N, K=1024)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 40, seed=11)
    exp = [oracle.pq_search(ot, q, 5)[None] for q in qs]
    dev = torch.device("cuda", 0)
    a = torch.randn(2048, 2048, device=dev)
    b = torch.randn(512, 512, device=dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for rep in range(3):
        for i, q in enumerate(qs):
            with torch.cuda.stream(streams[0]):
                for _ in range(1 + i % 3):
                    a = (a @ a).clamp_(-1, 1)
            with torch.cuda.stream(streams[1]):
                for _ in range(4):
                    b = (b @ b).clamp_(-1, 1)
            gi, gd = idx.search(q[None], 5, sentinel=100.0)
            util.assert_same_lists(gi, gd, exp[i], f"one launch under load, pass {rep} query {i}")
    torch.cuda.synchronize()
    idx.close()


@pytest.mark.parametrize("K", [256, 1024])
def test_ivfadc_single_query_one_launch(gpu, oracle, K):
    """ivfadc_search for ONE query -- the reference's own call shape (freddy.c:174-393) -- is one launch (one.h
    ivf_one_kernel: coarse distances -> barrier -> the W nearest cells -> the cells' tables -> barrier -> scan ->
    last-arriver merge).  Consecutive calls with different queries; both rules of counting found rows; W from 1 to 32; k up
    to the selection width; duplicate rows (equal distances); a query far from every centroid (no cell below the limit: the
    empty list); a sentinel so low that fewer than k rows are found (the reference probes again: the host falls back to the
    multi-round path); each list equal to the oracle's and to the multi-launch path's."""
    N = 40000
    t = dict(util.ivf_tables(N=N, C=100, K=K))
    codes = t["codes"].copy()
    lo = t["list_off"]
    for c in range(len(lo) - 1):   # the first rows of every list share a code row: equal distances
        n = min(4, int(lo[c + 1] - lo[c]))
        codes[lo[c]:lo[c] + n] = codes[lo[c]]
    t["codes"] = codes
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 40, seed=17)
    qs[5] *= np.float32(30.0)   # no centroid within the cell limit of 100
    idx.profile_enable(True)
    idx.search(qs[:1], 5, 10, sentinel=1000.0, found_rule=0)
    prof = idx.profile_read()
    idx.profile_enable(False)
    assert "ivf_one" in prof, f"the one-launch kernel did not run: {sorted(prof)}"
    for k, W, rule, sent in ((5, 10, 0, 1000.0), (1, 1, 0, 1000.0), (10, 4, 1, 100.0), (32, 32, 0, 1000.0), (5, 3, 1, 0.4), (17, 7, 0, 0.9)):
        for i, q in enumerate(qs):
            exp = oracle.ivfadc_search(ot, q, k, W, sentinel=sent, found_rule=rule)[None]
            gi, gd = idx.search(q[None], k, W, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"one launch K={K} k={k} W={W} rule={rule} sentinel={sent} query {i}")
    idx.set_option("one_launch", 0)
    for i, q in enumerate(qs[:8]):
        gi, gd = idx.search(q[None], 5, 10, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, oracle.ivfadc_search(ot, q, 5, 10, sentinel=1000.0, found_rule=0)[None], f"multi-launch, query {i}")
    idx.close()


def test_pq_search_in_and_batch(gpu, oracle):
    N = 20000
    t = util.pq_tables(N=N, K=256)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 12)
    rng = np.random.default_rng(3)
    targets = rng.choice(np.arange(1, N + 1), size=3000, replace=False).astype(np.int32)
    targets = np.concatenate([targets, targets[:50], np.array([N + 5, -3], np.int32)])  # dups + unknown ids
    k = 5
    gi, gd = idx.search(qs, k, sentinel=1000.0, subset_ids=targets)
    exp_b = oracle.pq_search_in_batch(ot, qs, k, targets, use_target_lists=True)
    exp_n = oracle.pq_search_in_batch(ot, qs, k, targets, use_target_lists=False)
    assert np.array_equal(exp_b, exp_n)
    util.assert_same_lists(gi, gd, exp_b, "pq_search_in_batch")
    exp_1 = oracle.pq_search_in(ot, qs[0], k, targets)
    util.assert_same_lists(gi[:1], gd[:1], exp_1, "pq_search_in")
    # fewer targets than k: sentinel rows survive
    gi, gd = idx.search(qs[:2], 7, sentinel=1000.0, subset_ids=targets[:3])
    exp = np.stack([oracle.pq_search_in(ot, q, 7, targets[:3]) for q in qs[:2]])
    util.assert_same_lists(gi, gd, exp, "pq_search_in short")
    # empty subset
    gi, gd = idx.search(qs[:2], 3, sentinel=1000.0, subset_ids=np.array([N + 9], np.int32))
    assert (gi == -1).all() and (gd == np.float32(1000.0)).all()
    idx.close()


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("K,W", [(256, 1), (256, 3), (256, 10), (1024, 3)])
def test_ivfadc_matches_oracle(gpu, oracle, K, W, fused, monkeypatch):
    """FREDDY_GPU_FUSED=1: ivf_fused_kernel (LUT slabs in LDS); 0: lut_build + adc_scan kernels."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    N = 20000
    t = util.ivf_tables(N=N, C=32, K=K)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 64)
    for k in (1, 5, 20):
        gi, gd = idx.search(qs, k, W, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"ivfadc K={K} W={W} k={k}")
    idx.close()


@pytest.mark.parametrize("variant", ["3", "5"])
@pytest.mark.parametrize("K", [256, 1024])
def test_fused_kernel_variants(gpu, oracle, K, variant, monkeypatch):
    """The cell-grouped scans (FREDDY_GPU_FUSED_KERNEL: 3 = the reference's arithmetic for every row,
    fused3.h; 5 = filter + refine with int16 slabs, fused5.h)
    against the oracle: many queries per cell (entries of
    every size incl. split cells), both found rules, k up to 32."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    monkeypatch.setenv("FREDDY_GPU_FUSED_KERNEL", variant)
    N = 20000
    t = util.ivf_tables(N=N, C=32, K=K)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 200)      # 200 x W items over 32 cells: up to ~60 items per cell
    for k, W, rule in ((5, 4, 0), (32, 2, 0), (3, 1, 1)):
        gi, gd = idx.search(qs, k, W, sentinel=1000.0 if rule == 0 else 100.0,
                            found_rule=gpu.FOUND_ROWS if rule == 0 else gpu.FOUND_ACCEPTED)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=1000.0 if rule == 0 else 100.0, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"variant {variant} K={K} k={k} W={W} rule={rule}")
    idx.close()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_ivfadc_batch_udf_semantics(gpu, oracle, fused, monkeypatch):
    """W=1, sentinel 100.0, found = accepted insertions == ivfadc_batch_search (freddy.c:679-999)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    N = 20000
    t = util.ivf_tables(N=N, C=32, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 100)
    gi, gd = idx.search(qs, 5, 1, sentinel=100.0, found_rule=gpu.FOUND_BATCH_UDF)
    exp = oracle.ivfadc_batch_search(ot, qs, 5)
    util.assert_same_lists(gi, gd, exp, "ivfadc_batch_search")
    idx.close()


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("scale", [14.0, 20.0])
def test_ivfadc_batch_udf_cell_limit(gpu, oracle, fused, scale, monkeypatch):
    """Non-normalised data: ivfadc_batch_search picks its cell by argmin from minDist = 1000 (freddy.c:853-866),
    ivfadc_search's cell list never admits a cell at distance >= 100 (freddy.c:266-283).  With every coarse
    distance in [100, 1000) the batch UDF keeps probing where the W-probe search retires the query."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    N = 20000
    t = dict(util.ivf_tables(N=N, C=32, K=256))
    t["coarse"] = (t["coarse"] * np.float32(scale)).astype(np.float32)
    t["codebook"] = (t["codebook"] * np.float32(scale)).astype(np.float32)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 80)
    qs = (qs * np.float32(scale)).astype(np.float32)
    qs[::2] += np.float32(0.04 * scale)        # pushed further out: every row beyond the sentinel, all 32 cells get probed
    gi, gd = idx.search(qs, 5, 1, sentinel=100.0, found_rule=gpu.FOUND_BATCH_UDF)
    exp = oracle.ivfadc_batch_search(ot, qs, 5)
    util.assert_same_lists(gi, gd, exp, f"ivfadc_batch_search, data x{scale}")
    # the same queries under ivfadc_search's rule (cell list sentinel 100.0)
    gi2, gd2 = idx.search(qs, 5, 1, sentinel=100.0, found_rule=gpu.FOUND_ACCEPTED)
    exp2 = oracle.ivfadc_search_many(ot, qs, 5, 1, sentinel=100.0, found_rule=1)
    util.assert_same_lists(gi2, gd2, exp2, f"ivfadc_search W=1, data x{scale}")
    assert not np.array_equal(exp["id"].reshape(gi.shape), exp2["id"].reshape(gi.shape)), "the two cell limits must be told apart by this test"
    idx.close()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_ivfadc_multi_round_tiny_cells(gpu, oracle, fused, monkeypatch):
    """Cells with fewer than k rows force the reference's extra probing rounds (freddy.c:262,:377)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    N = 600
    x = util.corpus(N)
    from freddy_amd import index_build as ib
    t = ib.build_ivf_index(x, C=150, m=12, K=64, train_size=N, iters=3, seed=9)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    qs = x.numpy()[::7].astype(np.float32)
    for W, k, rule, sent in [(1, 10, 0, 1000.0), (2, 25, 0, 1000.0), (1, 10, 1, 100.0), (3, 40, 1, 100.0), (200, 5, 0, 1000.0)]:
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        exp = oracle.ivfadc_search_many(ot, qs, k, min(W, 150), sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"multi-round W={W} k={k} rule={rule}")
    # k larger than the whole table: every row is returned, the rest stays sentinel
    gi, gd = idx.search(qs[:3], 64 * 8, 150, sentinel=1000.0, found_rule=0)
    exp = oracle.ivfadc_search_many(ot, qs[:3], 64 * 8, 150, sentinel=1000.0, found_rule=0)
    util.assert_same_lists(gi, gd, exp, "k > N")
    idx.close()


def test_large_k(gpu, oracle):
    """k*pvf-sized requests of the SQL post-verification wrappers (freddy--0.0.1.sql:556-591)."""
    N = 20000
    t = util.ivf_tables(N=N, C=32, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 6)
    for k in (33, 100, 500):
        gi, gd = idx.search(qs, k, 3, sentinel=1000.0, found_rule=0)
        exp = oracle.ivfadc_search_many(ot, qs, k, 3, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"large k={k}")
    idx.close()


@pytest.mark.parametrize("k", [513, 600, 2000, 4096])
def test_k_beyond_512_ivfadc(gpu, oracle, k):
    """The reference allocates k entries for any k (freddy.c:236,258).  Lists of more than 512 entries (bigk.h): the 2k smallest
    keys selected 1024 at a time over the same rows, the guarded insertion's result in closed form.  300 rows per cell share one
    code row (the k-th place falls among equal distances), W = 1 needs several probing rounds (the carried list), both found
    rules, a sentinel that bites."""
    N = 20000
    t = dict(util.ivf_tables(N=N, C=32, K=256))
    codes = t["codes"].copy()
    lo = t["list_off"]
    for c in range(len(lo) - 1):
        n = min(300, int(lo[c + 1] - lo[c]))
        codes[lo[c]:lo[c] + n] = codes[lo[c]]
    t["codes"] = codes
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 5, seed=3)
    for W, rule, sent in ((3, 0, 1000.0), (1, 0, 1000.0), (2, 1, 1000.0), (4, 0, 1.5)):
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"k={k} W={W} rule={rule} sentinel={sent}")
    idx.close()


def test_k_4096_large_host_batch_is_chunked_not_refused(gpu, oracle):
    """a host-buffer batch of a few thousand queries at k = 4096 used to size its chunks by the LUT bytes alone and
    then fail with FREDDY_E_NOMEM on the partial-list / selected-key buffers (bigk.h).  max_queries_per_chunk counts them now:
    with a 48 MB budget the 2100 queries go through in several chunks, lists equal to the oracle's."""
    N, k, W = 20000, 4096, 2
    t = util.ivf_tables(N=N, C=32, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx.set_option("lut_budget_mb", 48)
    _, qs = util.queries_from_corpus(N, 2100, seed=17)
    gi, gd = idx.search(qs, k, W, sentinel=1000.0, found_rule=0)
    exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=1000.0, found_rule=0, n_threads=min(32, os.cpu_count() or 1))
    util.assert_same_lists(gi, gd, exp, "k=4096, 2100 queries per host-buffer call")
    idx.close()


@pytest.mark.parametrize("k", [600, 2000])
def test_k_beyond_512_pq(gpu, oracle, k):
    """pq_search and pq_search_in with k > 512 (freddy.c:66,89): equal distances at the k-th place, a subset smaller than k."""
    N = 20000
    t = util.pq_tables(N=N, K=256)
    codes = t["codes"].copy()
    codes[1000:1900] = codes[999]   # 900 rows with the same code row
    ot = oracle.pq_table(t["codebook"], t["ids"], codes)
    idx = gpu.PQIndex(t["codebook"], t["ids"], codes)
    _, qs = util.queries_from_corpus(N, 3, seed=9)
    qs[0] = np.asarray(util.corpus(N)[1200].numpy(), np.float32)   # (a query that is one of the equal rows' neighbours)
    gi, gd = idx.search(qs, k, sentinel=100.0)
    exp = np.stack([oracle.pq_search(ot, q, k) for q in qs])
    util.assert_same_lists(gi, gd, exp, f"pq_search k={k}")
    rng = np.random.default_rng(3)
    for n_t in (3000, k - 50):
        targets = rng.choice(np.arange(1, N + 1), size=n_t, replace=False).astype(np.int32)
        gi, gd = idx.search(qs, k, sentinel=1000.0, subset_ids=targets)
        exp = oracle.pq_search_in_batch(ot, qs, k, targets, use_target_lists=True)
        util.assert_same_lists(gi, gd, exp, f"pq_search_in k={k} targets={n_t}")
    idx.close()


def _join_setup(oracle, gpu, N=20000, k_coarse=8, K=32, m=30):
    t = util.ivpq_tables(N=N, m=m, K=K, k_coarse=k_coarse)
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    return t, ot, idx


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("use_tl", [True, False])
def test_knn_join_matches_oracle(gpu, oracle, method, use_tl):
    """ivpq_search_in (ivpq_search_in.c:61-699): ids and ranks bit-exact; distances bit-exact
    (ADC and exact distances are both order-fixed fp32 sums, so 1e-5 is not even needed)."""
    N = 20000
    t, ot, idx = _join_setup(oracle, gpu, N)
    _, qs = util.queries_from_corpus(N, 120, seed=21)
    rng = np.random.default_rng(5)
    targets = rng.choice(np.arange(1, N + 1), size=4000, replace=False).astype(np.int32)
    for (k, alpha, pvf) in [(5, 3, 20), (5, 100, 20), (3, 1, 4), (10, 10, 7)]:
        gi, gd, git = idx.knn_join(qs, k, targets, alpha, pvf, method, use_target_lists=use_tl, confidence=0.8)
        exp, eit = oracle.ivpq_search_in(ot, qs, k, targets, alpha, pvf, method, use_target_lists=use_tl, confidence=0.8)
        assert git == eit, (git, eit)
        util.assert_same_lists(gi, gd, exp, f"knn_join method={method} tl={use_tl} k={k} alpha={alpha} pvf={pvf}")
    # the query batch in a pinned buffer (freddy_gpu_host_alloc: read by the kernels where it is, no staging copy), a changed and
    # a repeated target array (the repeated one finds its buckets in place)
    pb = gpu.PinnedBuffer(qs.shape)
    pb.array[:] = qs
    for tg in (targets[::-1].copy(), targets, targets, targets[:1000].copy()):
        gi, gd, git = idx.knn_join(pb.array, 5, tg, 10, 20, method, use_target_lists=use_tl, confidence=0.8)
        exp, eit = oracle.ivpq_search_in(ot, qs, 5, tg, 10, 20, method, use_target_lists=use_tl, confidence=0.8)
        assert git == eit, (git, eit)
        util.assert_same_lists(gi, gd, exp, f"knn_join pinned queries method={method} tl={use_tl} targets={len(tg)}")
    pb.close()
    # a VIEW into a pinned buffer that starts one float in (4-byte aligned only): the kernel's 16-byte loads must not see it --
    # the call stages it like pageable memory
    pb = gpu.PinnedBuffer((qs.size + 1,))
    view = pb.array[1:].reshape(qs.shape)
    view[:] = qs
    assert view.ctypes.data % 16 == 4
    gi, gd, git = idx.knn_join(view, 5, targets, 10, 20, method, use_target_lists=use_tl, confidence=0.8)
    exp, eit = oracle.ivpq_search_in(ot, qs, 5, targets, 10, 20, method, use_target_lists=use_tl, confidence=0.8)
    assert git == eit, (git, eit)
    util.assert_same_lists(gi, gd, exp, f"knn_join pinned view at an odd offset method={method} tl={use_tl}")
    del view
    pb.close()
    idx.close()


def test_knn_join_post_verification_beyond_1024_candidates(gpu, oracle):
    """k * pvf > 1024 (the SQL default pvf = 20 with k > 51; ivpq_search_in.c:238-259 allocates k * pvf entries for any value):
    the candidates of the post verification selected 1024 per pass over the query's rows (join_query_kernel<16, true>).  Queries
    with more candidate rows than k * pvf and with fewer; pair codes; lists and iteration counts equal the oracle's."""
    N = 20000
    t, ot, idx = _join_setup(oracle, gpu, N)
    _, qs = util.queries_from_corpus(N, 24, seed=23)
    rng = np.random.default_rng(8)
    targets = rng.choice(np.arange(1, N + 1), size=15000, replace=False).astype(np.int32)
    for (k, alpha, pvf, tg) in [(20, 150, 100, targets), (100, 30, 20, targets), (20, 300, 400, targets), (60, 5, 20, targets[:3000])]:
        for use_tl in (True, False):
            gi, gd, git = idx.knn_join(qs, k, tg, alpha, pvf, 2, use_target_lists=use_tl, confidence=0.8)
            exp, eit = oracle.ivpq_search_in(ot, qs, k, tg, alpha, pvf, 2, use_target_lists=use_tl, confidence=0.8)
            assert git == eit, (git, eit)
            util.assert_same_lists(gi, gd, exp, f"knn_join k={k} alpha={alpha} pvf={pvf} tl={use_tl} targets={len(tg)}")
    gi, gd, git = idx.knn_join(qs, 20, targets, 150, 100, 2, double_threshold=20)
    exp, eit = oracle.ivpq_search_in(ot, qs, 20, targets, 150, 100, 2, double_threshold=20)
    assert git == eit
    util.assert_same_lists(gi, gd, exp, "knn_join k*pvf=2000, pair codes")
    idx.close()


def test_knn_join_edge_cases(gpu, oracle):
    N = 20000
    t, ot, idx = _join_setup(oracle, gpu, N)
    _, qs = util.queries_from_corpus(N, 40, seed=22)
    rng = np.random.default_rng(6)
    targets = rng.choice(np.arange(1, N + 1), size=300, replace=False).astype(np.int32)
    # few targets (k*alpha > |targets| -> confidence 0 -> every cell), duplicates, unknown ids
    t2 = np.concatenate([targets[:7], targets[:3], np.array([N + 100], np.int32)])
    for method in (0, 1, 2):
        for tg, k, alpha in [(t2, 5, 3), (targets, 5, 1000), (targets[:2], 5, 1), (targets, 200, 2)]:
            pvf = 3
            gi, gd, git = idx.knn_join(qs, k, tg, alpha, pvf, method)
            exp, eit = oracle.ivpq_search_in(ot, qs, k, tg, alpha, pvf, method)
            assert git == eit
            util.assert_same_lists(gi, gd, exp, f"edge method={method} k={k} alpha={alpha} T={tg.size}")
    # pair LUT ("double codes", ivpq_search_in.c:262-279): alpha*k > threshold
    for method in (0, 2):
        gi, gd, git = idx.knn_join(qs, 5, targets, 10, 4, method, double_threshold=20)
        exp, eit = oracle.ivpq_search_in(ot, qs, 5, targets, 10, 4, method, double_threshold=20)
        assert git == eit
        util.assert_same_lists(gi, gd, exp, f"double codes method={method}")
    # confidence sweep changes the probed cell sets
    for conf in (0.1, 0.5, 0.95, 0.999):
        gi, gd, git = idx.knn_join(qs, 5, targets, 4, 5, 2, confidence=conf)
        exp, eit = oracle.ivpq_search_in(ot, qs, 5, targets, 4, 5, 2, confidence=conf)
        assert git == eit
        util.assert_same_lists(gi, gd, exp, f"confidence={conf}")
    idx.close()


@pytest.mark.parametrize("method", [0, 1, 2])
def test_knn_join_alpha_doubling_rounds(gpu, oracle, method):
    """Low confidence + few targets: queries come back with fewer than k results and are
    re-queued with alpha doubled (ivpq_search_in.c:639-680), incl. the target-list skip rule."""
    N = 20000
    t, ot, idx = _join_setup(oracle, gpu, N)
    _, qs = util.queries_from_corpus(N, 40, seed=22)
    rng = np.random.default_rng(6)
    targets = rng.choice(np.arange(1, N + 1), size=300, replace=False).astype(np.int32)
    seen_multi = False
    for (T, k, alpha, conf) in [(300, 5, 1, 0.3), (100, 3, 1, 0.5), (60, 5, 1, 0.2), (300, 20, 1, 0.05)]:
        for use_tl in (True, False):
            gi, gd, git = idx.knn_join(qs, k, targets[:T], alpha, 3, method, use_target_lists=use_tl, confidence=conf)
            exp, eit = oracle.ivpq_search_in(ot, qs, k, targets[:T], alpha, 3, method, use_target_lists=use_tl,
                                             confidence=conf)
            assert git == eit
            seen_multi |= eit > 1
            util.assert_same_lists(gi, gd, exp, f"rounds method={method} T={T} k={k} conf={conf} tl={use_tl}")
    assert seen_multi
    idx.close()


def test_ivfadc_long_lists_overflow_units(gpu, oracle):
    """Lists longer than 4096 rows are split into several work units of the fused kernel."""
    N = 20000
    t = util.ivf_tables(N=N, C=3, K=256)      # ~6.7k rows per list
    assert np.diff(t["list_off"]).max() > 4096
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 50)
    for W, k in [(1, 5), (2, 10), (3, 32)]:
        gi, gd = idx.search(qs, k, W)
        exp = oracle.ivfadc_search_many(ot, qs, k, W)
        util.assert_same_lists(gi, gd, exp, f"long lists W={W} k={k}")
    idx.close()


@pytest.mark.parametrize("fused", ["1", "0", "auto"])
def test_ivfadc_randomised_small_indexes(gpu, oracle, fused, monkeypatch):
    """FREDDY_GPU_FUSED=1 forces the fused kernel even for tiny batches, 0 the generic kernels,
    unset lets the library choose.  Many small random tables with deliberately nasty shapes: empty cells, cells longer than one
    4096-row chunk, more than 16 queries probing one cell (several fused groups per cell), heavy
    distance ties (few distinct code rows), k up to 32, W up to C."""
    if fused != "auto":
        monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    rng = np.random.default_rng(123)
    d, m = 300, 12
    for trial in range(12):
        K = int(rng.choice([4, 16, 64, 256]))
        C = int(rng.choice([1, 2, 5, 9]))
        N = int(rng.choice([50, 700, 9000]))
        coarse = rng.standard_normal((C, d)).astype(np.float32)
        codebook = (rng.standard_normal((m, K, 25)) * 0.3).astype(np.float32)
        cell = rng.integers(0, C, size=N) if trial % 3 else np.zeros(N, np.int64)   # every 3rd: all rows in one cell
        if C > 2:
            cell[cell == 1] = 0                                                      # cell 1 stays empty
        order = np.argsort(cell, kind="stable")
        ids = (np.arange(N) * 3 + 7).astype(np.int32)[order]                         # non-contiguous ids
        n_distinct = int(rng.choice([1, 3, 50]))
        pool = rng.integers(0, K, size=(n_distinct, m)).astype(np.int16)
        codes = pool[rng.integers(0, n_distinct, size=N)][order]
        list_off = np.zeros(C + 1, np.int32)
        list_off[1:] = np.cumsum(np.bincount(cell, minlength=C))
        ids_sorted = np.concatenate([np.sort(ids[list_off[c]:list_off[c + 1]]) for c in range(C)]).astype(np.int32)
        ot = oracle.ivf_table(coarse, codebook, list_off, ids_sorted, codes)
        idx = gpu.IVFIndex(coarse, codebook, list_off, ids_sorted, codes)
        Q = int(rng.choice([1, 40, 300]))
        qs = (coarse[rng.integers(0, C, size=Q)] + 0.2 * rng.standard_normal((Q, d))).astype(np.float32)
        for k, W in [(1, 1), (5, min(3, C)), (32, C)]:
            for rule, sent in [(0, 1000.0), (1, 100.0)]:
                gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
                exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
                util.assert_same_lists(gi, gd, exp, f"random trial={trial} K={K} C={C} N={N} Q={Q} k={k} W={W} rule={rule}")
        idx.close()


def test_knn_join_non_contiguous_ids(gpu, oracle):
    """Row ids with gaps (the O(1) id -> row shortcut for consecutive ids must not be taken)."""
    N = 20000
    t = dict(util.ivpq_tables(N=N))
    t["ids"] = (t["ids"].astype(np.int64) * 3 + 7).astype(np.int32)
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    _, qs = util.queries_from_corpus(N, 50, seed=31)
    rng = np.random.default_rng(9)
    targets = t["ids"][rng.choice(N, 3000, replace=False)]
    targets = np.concatenate([targets, targets[:20], np.array([8, 9, 5, 10**8], np.int32)])   # dups, gaps, unknown
    for method in (0, 2):
        gi, gd, git = idx.knn_join(qs, 5, targets, 10, 4, method)
        exp, eit = oracle.ivpq_search_in(ot, qs, 5, targets, 10, 4, method)
        assert git == eit
        util.assert_same_lists(gi, gd, exp, f"non-contiguous ids method={method}")
    idx.close()


def test_exact_knn_matches_oracle(gpu, oracle):
    """SURVEY 8f-1 (next row): k_nearest_neighbour / knn_in_exact.  The similarity is the same
    order-fixed binary32 chain as cosine_similarity_bytea, so ids, ranks and similarities are
    bit-exact (ties by ascending id)."""
    N = 20000
    x = util.corpus(N).numpy()
    ids = (np.arange(N) * 2 + 3).astype(np.int32)
    idx = gpu.VectorIndex(ids, x)
    qs = x[::997][:16].copy()
    qs[3] = -qs[3]                       # negative similarities
    for k in (1, 5, 64, 200):
        gi, gs = idx.search(qs, k)
        for qi, q in enumerate(qs):
            exp = oracle.exact_knn(x, ids, q, k)
            assert gi[qi].tolist() == exp["id"].tolist(), (k, qi)
            assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32)), (k, qi)
    rng = np.random.default_rng(2)
    sub = np.concatenate([ids[rng.choice(N, 700, replace=False)], np.array([4, 10**8], np.int32)])
    gi, gs = idx.search(qs, 10, subset_ids=sub)
    for qi, q in enumerate(qs):
        exp = oracle.exact_knn(x, ids, q, 10, sub)
        assert gi[qi].tolist() == exp["id"].tolist()
        assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32))
    gi, gs = idx.search(qs[:2], 8, subset_ids=ids[:3])          # fewer rows than k
    assert (gi[:, 3:] == -1).all() and np.isneginf(gs[:, 3:]).all()
    exp = oracle.exact_knn(x, ids, qs[0], 8, ids[:3])
    assert gi[0, :3].tolist() == exp["id"].tolist()
    idx.close()


def _exact_expect(oracle, x, ids, qs, k):
    return [oracle.exact_knn(x, ids, q, k) for q in qs]


def test_exact_knn_lists_beyond_1024_entries(gpu, oracle):
    """ORDER BY ... FETCH FIRST k takes any k (freddy--0.0.1.sql:426-454): lists of 1025 .. 4096 entries are selected 1024 keys per
    pass over the same rows (exact.h FLOOR).  Duplicate rows put equal similarities across a pass boundary; a subset smaller than k
    pads with (-1, -inf); k = 4097 is refused."""
    N = 9000
    x = util.corpus(N).numpy()
    x[3000:4100] = x[2999]                            # 1101 copies: equal similarities around the 1024-th place for a query near them
    ids = (np.arange(N) * 2 + 5).astype(np.int32)
    idx = gpu.VectorIndex(ids, x)
    qs = np.stack([x[2999], x[17], -x[500]]).astype(np.float32)
    for k in (1025, 2048, 3000, 4096):
        gi, gs = idx.search(qs, k)
        for qi, q in enumerate(qs):
            exp = oracle.exact_knn(x, ids, q, k)
            assert gi[qi].tolist() == exp["id"].tolist(), (k, qi)
            assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32)), (k, qi)
    sub = ids[100:1500]                               # 1400 rows, k = 2000: two passes, the second runs out of rows
    gi, gs = idx.search(qs[:2], 2000, subset_ids=sub)
    exp = oracle.exact_knn(x, ids, qs[0], 2000, sub)
    assert gi[0, :1400].tolist() == exp["id"].tolist() and (gi[:, 1400:] == -1).all() and np.isneginf(gs[:, 1400:]).all()
    with pytest.raises(Exception):
        idx.search(qs[:1], 4097)
    idx.close()


@pytest.mark.parametrize("scale", [1.0, 3000.0, 2e-4])
def test_exact_knn_filter_refine_matches_oracle(gpu, oracle, scale):
    """exact2.h: f16-split MFMA similarities with a proven bracket rank the rows, the reference's chain runs for the rows
    that can reach the k-th largest similarity -- ids, ranks and similarity bits equal fo_exact_knn / the all-exact kernel.
    `scale`: unnormalised tables (large / tiny elements exercise the power-of-two operand scaling)."""
    N = 24000
    x = (util.corpus(N).numpy() * np.float32(scale)).astype(np.float32)
    x[5000:5040] = x[100:140]                       # duplicate rows: equal similarities, ties by id
    ids = (np.arange(N) * 3 + 7).astype(np.int32)
    idx = gpu.VectorIndex(ids, x)
    qs = x[::331][:70].copy()                       # 70 queries: a full pass of 64 (two MFMA tiles) and one of 6 (one tile)
    qs[3] = -qs[3]
    qs[9] = x[100]                                  # a query with 2 exact copies in the table
    for k in (1, 5, 32):
        idx.set_option("exact_filter", -1)
        gi, gs = idx.search(qs, k)
        idx.set_option("exact_filter", 0)
        hi, hs = idx.search(qs, k)                  # the all-exact kernels
        assert np.array_equal(gi, hi) and np.array_equal(gs.view(np.uint32), hs.view(np.uint32)), k
        for qi in (0, 3, 9, 33, 64, 69):
            exp = oracle.exact_knn(x, ids, qs[qi], k)
            assert gi[qi].tolist() == exp["id"].tolist(), (k, qi)
            assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32)), (k, qi)
    assert idx.bound_violations() == 0
    idx.close()


def test_exact_knn_filter_bracket_holds_for_every_row(gpu, oracle):
    """Option exact_refine_all: EVERY row goes through the refine stage, which has both the MFMA value and the reference's
    similarity in hand -- the bracket |a - s| <= eps(q) is checked for every (row, query) pair, and the lists still equal
    the oracle's."""
    N = 16384 + 77
    rng = np.random.default_rng(5)
    x = util.corpus(N).numpy()
    x[1000:1100] *= np.float32(37.5)                # a few long rows: the norm bound X is their norm
    x[2000:2050] = 0.0
    ids = np.arange(1, N + 1, dtype=np.int32)
    idx = gpu.VectorIndex(ids, x)
    qs = np.concatenate([x[::777][:20], rng.standard_normal((4, x.shape[1])).astype(np.float32) * np.float32(0.01)])
    idx.set_option("check_brackets", 4)
    gi, gs = idx.search(qs, 5)
    assert idx.bound_violations() == 0
    assert int(idx.lib.freddy_gpu_filter_bound_checked(idx.h)) == qs.shape[0] * N
    for qi in range(qs.shape[0]):
        exp = oracle.exact_knn(x, ids, qs[qi], 5)
        assert gi[qi].tolist() == exp["id"].tolist(), qi
        assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32)), qi
    idx.close()


def test_exact_knn_filter_after_append_rows(gpu, oracle):
    """freddy_gpu_append_rows on a vector handle extends the fragment-order copy the filter reads (the strip the old last
    row sat in is rewritten; a new largest element changes the power-of-two scale and lays everything out again)."""
    N0, N1, N2 = 9000, 9500, 10007
    x = util.corpus(N2).numpy()
    x[N1 + 5] *= np.float32(300.0)                  # arrives with the second append: a smaller scale for the whole table
    ids = np.arange(1, N2 + 1, dtype=np.int32)
    idx = gpu.VectorIndex(ids[:N0], x[:N0])
    qs = np.concatenate([x[::1111][:9], x[N0 + 3:N0 + 5], x[N1 + 5:N1 + 6]])
    for n in (N1, N2):
        lo = N0 if n == N1 else N1
        idx.append_rows(ids[lo:n], vectors=x[lo:n])
        gi, gs = idx.search(qs, 5)
        for qi in range(qs.shape[0]):
            exp = oracle.exact_knn(x[:n], ids[:n], qs[qi], 5)
            assert gi[qi].tolist() == exp["id"].tolist(), (n, qi)
            assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32)), (n, qi)
    assert idx.bound_violations() == 0
    idx.close()


def test_exact_knn_filter_edge_cases(gpu, oracle):
    x = util.corpus(300).numpy()
    ids = (np.arange(300) + 1).astype(np.int32)
    idx = gpu.VectorIndex(ids, x)
    idx.set_option("exact_filter", 1)               # forced: fewer rows than the sample, than a strip, than k
    for n_q, k in ((1, 5), (3, 32)):
        gi, gs = idx.search(x[:n_q], k)
        for qi in range(n_q):
            exp = oracle.exact_knn(x, ids, x[qi], k)
            assert gi[qi].tolist() == exp["id"].tolist()
            assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32))
    idx.close()
    idx = gpu.VectorIndex(ids[:3], x[:3])
    idx.set_option("exact_filter", 1)
    gi, gs = idx.search(x[:2], 8)                    # fewer rows than k
    assert (gi[:, 3:] == -1).all() and np.isneginf(gs[:, 3:]).all()
    assert gi[0, :3].tolist() == oracle.exact_knn(x[:3], ids[:3], x[0], 8)["id"].tolist()
    # a dense neighbourhood: more rows within the bracket of the k-th than the candidate buffer holds -> all-exact kernels
    N = 20000
    y = np.tile(util.corpus(1).numpy(), (N, 1))
    y[::2] *= np.float32(1.0 - 1e-7)
    yid = np.arange(1, N + 1, dtype=np.int32)
    idx2 = gpu.VectorIndex(yid, y)
    gi, gs = idx2.search(y[:2], 5)
    for qi in range(2):
        exp = oracle.exact_knn(y, yid, y[qi], 5)
        assert gi[qi].tolist() == exp["id"].tolist()
        assert np.array_equal(gs[qi].view(np.uint32), exp["dist"].view(np.uint32))
    # a query that is not finite: the all-exact kernels answer (same bits as before)
    bad = y[:1].copy(); bad[0, 7] = np.inf
    idx2.set_option("exact_filter", 0)
    hi, hs = idx2.search(bad, 5)
    idx2.set_option("exact_filter", -1)
    gi, gs = idx2.search(bad, 5)
    assert np.array_equal(gi, hi) and np.array_equal(gs.view(np.uint32), hs.view(np.uint32))
    idx2.close()
    idx.close()


@pytest.mark.parametrize("K,m", [(256, 12), (1024, 12), (32, 30)])
def test_encode_matches_oracle(gpu, oracle, K, m):
    """Next row 8f-2, encoding step of the index build: cells and PQ codes bit-identical to the oracle
    (exact 1-NN by squareDistance, lowest index on ties), flat PQ and IVFADC (codes of the residuals)."""
    N = 3000
    x = util.corpus(N).numpy()
    rng = np.random.default_rng(K + m)
    s = 300 // m
    cb = (rng.standard_normal((m, K, s)) * 0.05).astype(np.float32)
    cb[0, K - 1] = cb[0, 1]                              # duplicated codeword
    cb[2, 5] = x[11, 2 * s:3 * s]                        # an exact hit
    cell, codes = gpu.encode(cb, x)
    assert cell is None
    assert np.array_equal(codes, oracle.encode_pq(cb, x))
    assert codes[11, 2] == 5 and (codes[:, 0] != K - 1).all()
    coarse = x[rng.choice(N, 40, replace=False)].copy()
    coarse[17] = coarse[3]
    cell, codes = gpu.encode(cb, x, coarse=coarse)
    exp_cell = oracle.assign_coarse(coarse, x)
    assert np.array_equal(cell, exp_cell) and (cell != 17).all()
    res = np.stack([oracle.vec_minus(x[i], coarse[exp_cell[i]]) for i in range(N)])
    assert np.array_equal(codes, oracle.encode_pq(cb, res))


@pytest.mark.parametrize("fused", ["1", "0"])
def test_ivfadc_many_probes(gpu, oracle, fused, monkeypatch):
    """W > 64 takes the probe plan's wide selection (V = 4) and its serial replay branch; W = C probes
    every cell (exhaustive), so the result equals the best over the whole table."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    N = 20000
    t = util.ivf_tables(N=N, C=128, K=64)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 24)
    for W in (65, 100, 128):
        gi, gd = idx.search(qs, 4, W, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
        exp = oracle.ivfadc_search_many(ot, qs, 4, W, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"W={W}")
    idx.close()


def test_hip_path_reproduces_committed_golden_vectors(gpu, monkeypatch):
    """The HIP path against the committed vectors of tests/golden/ (no oracle involved on this side)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "freddy_small.npz"))
    x, ids, qs = g["x"], g["ids"], g["queries"]

    def same(gi, gd, exp, what):
        assert np.array_equal(gi, exp["id"]), what
        assert np.array_equal(gd.view(np.uint32), exp["dist"].view(np.uint32)), what

    _, codes = gpu.encode(g["pq_codebook"], x)
    assert np.array_equal(codes, g["pq_codes"])
    pq = gpu.PQIndex(g["pq_codebook"], ids, g["pq_codes"])
    same(*pq.search(qs, 5, sentinel=100.0), g["pq_search"], "pq_search")
    same(*pq.search(qs, 4, sentinel=1000.0, subset_ids=g["subset"]), g["pq_search_in"], "pq_search_in")
    gi, gg = pq.grouping(x[[9, 199, 349]], g["grouping_input"])
    assert np.array_equal(gi, g["grouping_ids"]) and np.array_equal(gg, g["grouping_group"])
    pq.close()
    ivf = gpu.IVFIndex(g["coarse"], g["codebook"], g["list_off"], g["ivf_ids"], g["ivf_codes"])
    for fused in ("1", "0"):
        monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
        same(*ivf.search(qs, 5, 3, sentinel=1000.0, found_rule=gpu.FOUND_ROWS), g["ivfadc_w3_k5"], f"ivfadc fused={fused}")
        same(*ivf.search(qs, 7, 1, sentinel=100.0, found_rule=gpu.FOUND_ACCEPTED), g["ivfadc_w1_k7_accepted"], f"batch rule fused={fused}")
        same(*ivf.search(qs, 20, 8, sentinel=1000.0, found_rule=gpu.FOUND_ROWS), g["ivfadc_w8_k20"], f"W=8 fused={fused}")
    ivf.close()
    iv = gpu.IVPQIndex(g["ivpq_codebook"], g["ivpq_coarse"], ids, g["ivpq_coarse_id"], g["ivpq_codes"], x, g["ivpq_stats"])
    for method in (0, 1, 2):
        gi, gd, it = iv.knn_join(qs, 5, g["targets"], 3, 4, method)
        same(gi, gd, g[f"knn_join_m{method}"], f"knn_join method {method}")
        assert it == int(g[f"knn_join_m{method}_iterations"])
    iv.close()
    vi = gpu.VectorIndex(ids, x)
    gi, gs = vi.search(qs, 6)
    same(gi, gs, g["exact_knn"], "exact kNN")
    vi.close()


# ---------------------------------------------------------------------------------------
# filter + refine scan (fused5.h + refine.h, the default for batches): the cases its bounds have to survive
# ---------------------------------------------------------------------------------------
def _fr_setup(gpu, oracle, K=256, scale=1.0, dup_rows=0):
    t = dict(util.ivf_tables(N=20000, C=32, K=K))
    if scale != 1.0:
        t["coarse"] = (t["coarse"] * np.float32(scale)).astype(np.float32)
        t["codebook"] = (t["codebook"] * np.float32(scale)).astype(np.float32)
    if dup_rows:
        # the first dup_rows rows of every list get the list's first code row: that many exactly equal distances
        codes = t["codes"].copy()
        lo = t["list_off"]
        for c in range(len(lo) - 1):
            n = min(dup_rows, int(lo[c + 1] - lo[c]))
            codes[lo[c]:lo[c] + n] = codes[lo[c]]
        t["codes"] = codes
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(20000, 120)
    return t, ot, idx, (qs * np.float32(scale)).astype(np.float32)


def test_filter_refine_many_equal_distances(gpu, oracle, monkeypatch):
    """Hundreds of rows with the SAME exact distance around the k-th place: more rows inside the error
    margin than the merge keeps in registers (every key is revisited), order decided by id."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle, dup_rows=300)
    for k, W, rule, sent in ((5, 3, 0, 1000.0), (32, 2, 0, 1000.0), (10, 4, 1, 100.0)):
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"equal distances k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0, "a refined row's distance left its proven bracket"
    idx.close()


def test_exact_stage_reads_the_last_codeword_and_the_last_centroid(gpu, oracle, monkeypatch):
    """The merge's exact stage reads a chain's codeword / centroid slice as 16-byte units whose last one overlaps its neighbour (one
    wave per query, refine.h): rows that carry the LAST code at every position, in the LAST cell, so the last unit of the last
    codeword and of the last centroid row are read -- both merge instantiations (one batch at a time / batches in flight)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    rng = np.random.default_rng(77)
    d, m, K, C, N = 300, 12, 1024, 7, 6000
    coarse = rng.standard_normal((C, d)).astype(np.float32)
    codebook = (rng.standard_normal((m, K, 25)) * 0.3).astype(np.float32)
    cell = np.sort(rng.integers(0, C, size=N))
    codes = rng.integers(0, K, size=(N, m)).astype(np.int16)
    last = np.flatnonzero(cell == C - 1)
    codes[last[: len(last) // 2]] = K - 1                      # half of the last cell's rows: the last code everywhere
    codes[last[len(last) // 2:], m - 1] = K - 1                # the others: the last code at the last position
    list_off = np.zeros(C + 1, np.int32)
    list_off[1:] = np.cumsum(np.bincount(cell, minlength=C))
    ids = (np.arange(N) * 5 + 3).astype(np.int32)
    ot = oracle.ivf_table(coarse, codebook, list_off, ids, codes)
    idx = gpu.IVFIndex(coarse, codebook, list_off, ids, codes)
    # queries at the last centroid shifted by the last codewords: their nearest rows are the rows above
    tail = np.concatenate([codebook[p_, K - 1] for p_ in range(m)])
    qs = (coarse[C - 1] + tail + 0.05 * rng.standard_normal((300, d))).astype(np.float32)
    for share in (1, 4):
        idx.set_option("scan_share", share)
        for k, W in ((5, 2), (32, C)):
            gi, gd = idx.search(qs, k, W)
            exp = oracle.ivfadc_search_many(ot, qs, k, W)
            util.assert_same_lists(gi, gd, exp, f"last codeword / centroid, scan_share={share} k={k} W={W}")
    assert idx.bound_violations() == 0
    idx.close()


def test_filter_refine_sentinel_inside_the_data(gpu, oracle, monkeypatch):
    """The guard dist < sentinel (freddy.c:128-131) with the sentinel in the middle of the distances and
    exactly ON a row's distance: rows whose bound straddles it are decided by the exact stage, and with
    the batch rule the number of accepted rows decides whether a query goes to another round."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle)
    base = oracle.ivfadc_search_many(ot, qs, 8, 2, sentinel=1000.0, found_rule=0)
    d = base["dist"].reshape(len(qs), 8)
    for sent in (float(d[0, 3]), float(d[5, 0]), float(np.median(d[:, 7])), float(np.nextafter(d[9, 2], np.float32(0)))):
        for rule in (0, 1):
            gi, gd = idx.search(qs, 8, 2, sentinel=sent, found_rule=rule)
            exp = oracle.ivfadc_search_many(ot, qs, 8, 2, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"sentinel {sent!r} rule={rule}")
    assert idx.bound_violations() == 0, "a refined row's distance left its proven bracket"
    idx.close()


@pytest.mark.parametrize("scale", [1e-12, 30.0, 1e4])
def test_filter_refine_scaled_data(gpu, oracle, scale, monkeypatch):
    """Tiny magnitudes (squares near the denormals), distances above the sentinels (x30: around 100 and
    1000; x1e4: everything rejected): the bounds scale with the data."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle, scale=scale)
    for k, W, rule, sent in ((5, 3, 0, 1000.0), (5, 2, 1, 100.0)):
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"scale {scale} k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0, "a refined row's distance left its proven bracket"
    idx.close()


def test_filter_refine_overflowing_bound(gpu, oracle, monkeypatch):
    """Queries so large that the error bound itself overflows: every row goes to the exact stage (and
    is rejected there, its distance being +inf) -- same lists as the reference arithmetic gives."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle)
    qs = qs[:40].copy()
    qs[::4] *= np.float32(3e19)
    gi, gd = idx.search(qs, 5, 2, sentinel=1000.0, found_rule=0)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 2, sentinel=1000.0, found_rule=0)
    util.assert_same_lists(gi, gd, exp, "overflowing bound")
    assert idx.bound_violations() == 0, "a refined row's distance left its proven bracket"
    idx.close()


def test_filter_refine_bracket_holds_for_every_row(gpu, oracle, monkeypatch):
    """Debug switches make the scan keep EVERY probed row and the merge refine every one of them, so the
    kernel's self-check compares the proven bracket [d_lo, d_lo + E] with the reference's distance for all
    rows, not only for the few a normal run refines.  Results must not change either."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    for scale in (1.0, 1e-12, 8.0):   # (x8: coarse distances stay below the probe plan's limit of 100)
        t, ot, idx, qs = _fr_setup(gpu, oracle, scale=scale)
        idx.set_option("check_brackets", 1)
        qs = qs[:48]
        gi, gd = idx.search(qs, 5, 2, sentinel=1000.0, found_rule=0)
        exp = oracle.ivfadc_search_many(ot, qs, 5, 2, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"every row refined, scale {scale}")
        assert idx.bound_checked() > 20000, f"scale {scale}: {idx.bound_checked()} brackets checked"   # every probed row ...
        assert idx.bound_violations() == 0             # ... and none was violated
        idx.close()


# ---------------------------------------------------------------------------------------
# item-wise scan of thin cells (sparse5.h): the same brackets, regions and lists as the cell-grouped scan
# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("K", [256, 1024])
def test_sparse_item_scan_matches_oracle(gpu, oracle, K, monkeypatch):
    """Option sparse_items < 0 forces the item-wise scan for every cell with at most that many items: all cells (-16: the
    cell-grouped scan gets nothing), a mix (-2), none (0); every rule of counting found rows; multi-round searches.
    Cells of exactly two items are one unit (their rows read once)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle, K=K, dup_rows=3)
    for force in (-16, -2, 0):
        idx.set_option("sparse_items", force)
        for k, W, rule, sent in ((5, 3, 0, 1000.0), (10, 4, 1, 100.0), (5, 1, 2, 100.0), (32, 2, 0, 1000.0)):
            gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
            exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"sparse_items {force} K={K} k={k} W={W} rule={rule}")
        # few queries: most probed cells have one or two items (the pair units' case)
        for nq in (2, 7):
            gi, gd = idx.search(qs[:nq], 5, 3, sentinel=1000.0, found_rule=0)
            exp = oracle.ivfadc_search_many(ot, qs[:nq], 5, 3, sentinel=1000.0, found_rule=0)
            util.assert_same_lists(gi, gd, exp, f"sparse_items {force} K={K} {nq} queries")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.gpu
def test_sparse_item_scan_bracket_holds_for_every_row(gpu, oracle, monkeypatch):
    """As test_filter_refine_bracket_holds_for_every_row, every cell scanned item by item."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    for scale in (1.0, 1e-12, 8.0):
        t, ot, idx, qs = _fr_setup(gpu, oracle, scale=scale)
        idx.set_option("sparse_items", -16)
        idx.set_option("check_brackets", 1)
        g2, d2 = idx.search(qs[:12], 5, 2, sentinel=1000.0, found_rule=0)   # (many cells with exactly two items: pair units)
        util.assert_same_lists(g2, d2, oracle.ivfadc_search_many(ot, qs[:12], 5, 2, sentinel=1000.0, found_rule=0), f"12 queries, scale {scale}")
        qs = qs[:48]
        gi, gd = idx.search(qs, 5, 2, sentinel=1000.0, found_rule=0)
        exp = oracle.ivfadc_search_many(ot, qs, 5, 2, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"every row refined, item-wise scan, scale {scale}")
        assert idx.bound_checked() > 20000, f"scale {scale}: {idx.bound_checked()} brackets checked"
        assert idx.bound_violations() == 0
        idx.close()


@pytest.mark.gpu
def test_one_byte_code_layout_equals_the_int16_layout(gpu, oracle, monkeypatch):
    """K <= 256 (the reference's shipped default indexes): the integer-slab scans read packed8 -- one byte per code, 16 instead of
    28 B per row -- unless option codes_u8 = 0 keeps the int16 layout.  codes_u8 = 1 (default) is the scan that keeps the whole work
    entry's slab in LDS and reads the compact copy of the query table (fused8.h, round 6), 2 the six-phase kernel's one-byte
    instantiation (fused5.h).  Same lists from all three, for the cell-grouped scan and the item-wise one, both found rules, with
    every row's bracket checked, and after rows were appended (the byte array is rebuilt)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle, K=256, dup_rows=3)
    for sparse in (0, -16):
        idx.set_option("sparse_items", sparse)
        for k, W, rule, sent in ((5, 3, 0, 1000.0), (10, 4, 1, 100.0), (32, 2, 0, 1000.0)):
            exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
            for u8 in (1, 2, 0):
                idx.set_option("codes_u8", u8)
                gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
                util.assert_same_lists(gi, gd, exp, f"codes_u8={u8} sparse_items={sparse} k={k} W={W} rule={rule}")
    # every probed row through the exact stage: the brackets of the byte layout
    idx.set_option("sparse_items", 0); idx.set_option("codes_u8", 1)
    idx.set_option("check_brackets", 1)
    before = idx.bound_checked()
    gi, gd = idx.search(qs[:48], 5, 2, sentinel=1000.0, found_rule=0)
    util.assert_same_lists(gi, gd, oracle.ivfadc_search_many(ot, qs[:48], 5, 2, sentinel=1000.0, found_rule=0), "every row, byte layout")
    assert idx.bound_checked() - before > 20000
    idx.set_option("check_brackets", 0)
    # appended rows
    rng = np.random.default_rng(3)
    n_new = 500
    new_ids = (np.arange(n_new) + int(t["ids"].max()) + 1).astype(np.int32)
    new_cell = rng.integers(0, 32, n_new).astype(np.int32)
    new_codes = rng.integers(0, 256, (n_new, 12)).astype(np.int16)
    idx.append_rows(new_ids, coarse_id=new_cell, codes=new_codes)
    order = np.argsort(new_cell, kind="stable")
    lo = t["list_off"]
    ids2, codes2, off2 = [], [], [0]
    for c in range(32):
        sel = order[new_cell[order] == c]
        ids2.append(np.concatenate([t["ids"][lo[c]:lo[c + 1]], new_ids[sel]]))
        codes2.append(np.concatenate([t["codes"][lo[c]:lo[c + 1]], new_codes[sel]]))
        off2.append(off2[-1] + len(ids2[-1]))
    ot2 = oracle.ivf_table(t["coarse"], t["codebook"], np.array(off2, np.int32), np.concatenate(ids2), np.concatenate(codes2))
    exp = oracle.ivfadc_search_many(ot2, qs, 5, 3, sentinel=1000.0, found_rule=0)
    for u8 in (1, 2, 0):
        idx.set_option("codes_u8", u8)
        gi, gd = idx.search(qs, 5, 3, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"after append, codes_u8={u8}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.gpu
def test_sparse_item_scan_by_its_own_rule(gpu, oracle, monkeypatch):
    """More cells than the batch has probes and enough probes to fill the chip: the library picks the item-wise scan for
    the thin cells itself (default options); the lists equal the oracle's and those of the cell-grouped scan alone."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t = dict(util.ivf_tables(N=120000, C=3000, K=256))
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(120000, 1024)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 5, sentinel=1000.0, found_rule=0, n_threads=8)
    idx.profile_enable(True)
    gi, gd = idx.search(qs, 5, 5, sentinel=1000.0, found_rule=0)
    prof = idx.profile_read()
    idx.profile_enable(False)
    util.assert_same_lists(gi, gd, exp, "thin cells by rule")
    assert "sparse_items" in prof, f"the item-wise scan did not run: {sorted(prof)}"
    idx.set_option("sparse_items", 0)
    g2, d2 = idx.search(qs, 5, 5, sentinel=1000.0, found_rule=0)
    assert np.array_equal(gi, g2) and np.array_equal(gd.view(np.uint32), d2.view(np.uint32))
    assert idx.bound_violations() == 0
    idx.close()


# ---------------------------------------------------------------------------------------
# coarse-cell selection as filter + refine (coarse.h): MFMA distances with a proven bracket, the reference's
# squareDistance for the candidate cells only
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("scale", [1.0, 1e-12, 8.0, 1e4])
def test_coarse_filter_refine_every_cell(gpu, oracle, scale, monkeypatch):
    """Debug switch coarse_refine_all: EVERY (query, cell) pair gets both the MFMA value and the reference's
    distance, the kernel counts the pairs outside the bracket; the lists equal the oracle's and the ones of
    the all-exact coarse kernel (option coarse_approx = 0)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle, K=256, scale=scale)
    C = len(t["list_off"]) - 1
    for k, W, rule, sent in ((5, 3, 0, 1000.0), (3, 1, 1, 100.0), (8, 10, 0, 1000.0)):
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        idx.set_option("coarse_approx", 1)
        idx.set_option("check_brackets", 2)
        before = idx.coarse_bound_checked()
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"coarse refine all, scale {scale} k={k} W={W}")
        assert idx.coarse_bound_checked() - before >= len(qs) * C      # first round: every cell of every query
        idx.set_option("check_brackets", 0)
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"coarse filter + refine, scale {scale} k={k} W={W}")
        idx.set_option("coarse_approx", 0)
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"coarse all exact, scale {scale} k={k} W={W}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("kind", ["duplicate_centroids", "one_huge_centroid", "far_queries", "tiny_cells_many_rounds"])
def test_coarse_filter_refine_adversarial(gpu, oracle, kind, monkeypatch):
    """Equal coarse distances (identical centroids: the lowest cell index has to win exactly as in the
    reference's strict '<' / updateTopK), a centroid whose norm dwarfs the others (the bracket's width follows
    the LARGEST centroid norm: many candidates), queries far from everything (distances beyond the cell
    limit of 100), and extra probing rounds (cells already used are masked)."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    rng = np.random.default_rng(11)
    if kind == "tiny_cells_many_rounds":
        N = 900
        x = util.corpus(N)
        from freddy_amd import index_build as ib
        t = dict(ib.build_ivf_index(x, C=200, m=12, K=64, train_size=N, iters=3, seed=9))
        qs = np.repeat(x.numpy()[::9].astype(np.float32), 1, axis=0)
    else:
        t = dict(util.ivf_tables(N=20000, C=32, K=256))
        _, qs = util.queries_from_corpus(20000, 90)
    coarse = t["coarse"].copy()
    if kind == "duplicate_centroids":
        coarse[5] = coarse[17]
        coarse[30] = coarse[2]
        coarse[3] = coarse[2]
    elif kind == "one_huge_centroid":
        coarse[9] *= np.float32(300.0)
    elif kind == "far_queries":
        qs = qs.copy()
        qs[::3] += np.float32(0.5)     # coarse distances ~75 + : some beyond the limit of 100, some just below
        qs[1::3] *= np.float32(9.5)
    t["coarse"] = coarse
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    for k, W, rule, sent in ((5, 3, 0, 1000.0), (10, 1, 1, 100.0), (12, 4, 0, 1000.0)):
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        for approx in (1, 0):
            idx.set_option("coarse_approx", approx)
            gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"{kind} approx={approx} k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0
    idx.close()


def _every_row_check(idx, oracle, ot, qs, k, W, what, kernel=5):
    """Every probed row kept by the scan and refined by the merge: the kernel's self-check then compares the
    proven bracket with the reference's distance for ALL of them.  Returns the lists."""
    idx.set_option("fused", 1)
    idx.set_option("fused_kernel", kernel)
    idx.set_option("check_brackets", 1)
    before = idx.bound_checked()
    gi, gd = idx.search(qs, k, W, sentinel=1000.0, found_rule=0)
    checked = idx.bound_checked() - before
    rows = idx.last_scanned_rows()
    assert checked == rows, f"{what}: {checked} brackets checked, {rows} rows probed"
    assert idx.bound_violations() == 0, f"{what}: a row's distance left its proven bracket"
    idx.set_option("check_brackets", 0)
    exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=1000.0, found_rule=0)
    util.assert_same_lists(gi, gd, exp, what)
    return gi, gd


@pytest.mark.parametrize("kernel", [5])
def test_filter_refine_bracket_every_row_K1024(gpu, oracle, kernel, monkeypatch):
    """The instantiations the benchmark runs -- ivf_filter5_kernel<12, true> / ivf_filter_kernel<12, true>, K = 1024 --
    in the every-row mode: brackets checked == rows probed, none violated, lists equal to the oracle's, to the exact
    scan's (fused3.h) and to the normal filter + refine run."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    for scale in (1.0, 1e-12, 8.0, 1e4):
        t, ot, idx, qs = _fr_setup(gpu, oracle, K=1024, scale=scale)
        qs = qs[:96]
        gi, gd = _every_row_check(idx, oracle, ot, qs, 5, 3, f"K=1024 every row, scale {scale}, kernel {kernel}", kernel)
        idx.set_option("fused_kernel", 3)
        ei, ed = idx.search(qs, 5, 3, sentinel=1000.0, found_rule=0)
        idx.set_option("fused_kernel", kernel)
        ni, nd = idx.search(qs, 5, 3, sentinel=1000.0, found_rule=0)
        assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32)), "exact scan differs"
        assert np.array_equal(gi, ni) and np.array_equal(gd.view(np.uint32), nd.view(np.uint32)), "normal run differs"
        idx.close()


def _adversarial_tables(kind, K):
    """Index tables that stress the FIXED-POINT term of the filter's cheap distance (16-bit query x codeword
    table with one scale per (query, position), scale = 2 |q_p| max|c_p| / 32767)."""
    t = dict(util.ivf_tables(N=20000, C=32, K=K))
    cb = t["codebook"].copy()
    if kind == "huge_codeword":
        # one codeword per position 200x longer than the others: max|c_p| -- and with it the scale of every
        # table entry of that position -- grows 200x, i.e. typical entries keep ~7 bits
        for p in range(cb.shape[0]):
            cb[p, (7 * p + 3) % K] *= np.float32(200.0)
    elif kind == "huge_codeword_used":
        # ... and rows that actually carry such a codeword (their distances are huge, the others' are not)
        for p in range(cb.shape[0]):
            cb[p, (7 * p + 3) % K] *= np.float32(50.0)
        codes = t["codes"].copy()
        codes[::97, 5] = (7 * 5 + 3) % K
        t["codes"] = codes
    t["codebook"] = cb
    return t


@pytest.mark.parametrize("kind", ["huge_codeword", "huge_codeword_used", "dominant_subvector", "K300", "K40"])
def test_filter_refine_adversarial_fixed_point(gpu, oracle, kind, monkeypatch):
    """Inputs chosen against the fixed-point table: coarse scales (one huge codeword per position), queries
    whose norm sits in ONE sub-vector (one position's scale dwarfs the rest), and K < 512 / K not a
    multiple of anything (padding slots of the [512]-pair table layout).  Normal run and every-row run."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    K = {"K300": 300, "K40": 40}.get(kind, 1024 if kind != "dominant_subvector" else 256)
    t = _adversarial_tables(kind, K)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(20000, 72)
    if kind == "dominant_subvector":
        qs = qs.copy()
        for i in range(len(qs)):
            p = i % 12
            qs[i, p * 25:(p + 1) * 25] *= np.float32(40.0 if i % 2 else 6.0)
    for k, W, rule, sent in ((5, 3, 0, 1000.0), (10, 2, 1, 100.0)):
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        for kernel in (5,):
            idx.set_option("fused_kernel", kernel)
            gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"{kind} kernel={kernel} k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0
    for kernel in (5,):
        _every_row_check(idx, oracle, ot, qs[:40], 5, 2, f"{kind}, every row, kernel {kernel}", kernel)
    idx.close()


def test_row_order_inside_a_list_is_free(gpu, oracle, monkeypatch):
    """The pin-time arrangement of the rows inside a list (against LDS bank conflicts in the scans) must
    not be observable: the oracle's lists (canonical scan order = ascending id) on every scan path."""
    arrange = "1"
    t = util.ivf_tables(N=20000, C=32, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(20000, 150)
    exp = oracle.ivfadc_search_many(ot, qs, 10, 3, sentinel=1000.0, found_rule=0)
    for fused, variant in ((1, 5), (1, 3), (0, 5)):
        idx.set_option("fused", fused)
        idx.set_option("fused_kernel", variant)
        gi, gd = idx.search(qs, 10, 3, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"arrange={arrange} fused={fused} kernel={variant}")
    idx.close()


def test_kmeans_matches_oracle(gpu, oracle):
    """Quantizer training (quantizer_creation.py:13-52) as a native Lloyd k-means: centroids and assignment equal
    the restatement bit for bit (fixed summation order), for the coarse quantizer shape and a PQ sub-codebook."""
    x = util.corpus(20000).numpy()
    rng = np.random.default_rng(21)
    for vecs, k, iters in ((x[:6000], 40, 4), (np.ascontiguousarray(x[:5000, 50:75]), 256, 3), (x[:300], 7, 6)):
        init = rng.choice(len(vecs), k, replace=False).astype(np.int32)
        gc, ga = gpu.kmeans(vecs, k, iters, init)
        oc, oa = oracle.kmeans(vecs, k, iters, init)
        assert np.array_equal(ga, oa)
        assert np.array_equal(gc.view(np.uint32), oc.view(np.uint32))
        assert len(np.unique(ga)) > k // 2
    # duplicated initial rows: the second copy of a centroid loses every tie (strict "<") and, empty, keeps its value
    init = np.array([5, 5, 9], np.int32)
    gc, ga = gpu.kmeans(x[:200], 3, 0, init)
    assert (ga != 1).all()
    for iters in (1, 2):
        gc, ga = gpu.kmeans(x[:200], 3, iters, init)
        oc, oa = oracle.kmeans(x[:200], 3, iters, init)
        assert np.array_equal(ga, oa) and np.array_equal(gc.view(np.uint32), oc.view(np.uint32))
    # a trained PQ codebook encodes its own training vectors (end to end through the index build ABI)
    cb = gpu.train_pq_codebook(x[:4000], 12, 64, iters=3, seed=4)
    _, codes = gpu.encode(cb, x[:500])
    assert np.array_equal(codes, oracle.encode_pq(cb, x[:500]))


def test_searches_on_two_streams_overlap_safely(gpu, oracle):
    """Two batches in flight on two HIP streams (what bench.py does): every stream gets its own workspace inside
    the library, so interleaved device-pointer searches give exactly the lists of the same searches run alone."""
    import torch
    dev = torch.device("cuda", 0)
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qa = util.queries_from_corpus(N, 300)
    qb = np.ascontiguousarray(qa[::-1] * np.float32(1.01))
    exp = [oracle.ivfadc_search_many(ot, q, 5, 4, sentinel=1000.0, found_rule=0) for q in (qa, qb)]
    dq = [torch.from_numpy(q).to(dev) for q in (qa, qb)]
    res = [torch.zeros((2, 300, 5), dtype=torch.int32, device=dev) for _ in range(2)]
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize(dev)
    for rounds in range(6):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                res[i].zero_()
                idx.search_dev(dq[i].data_ptr(), 300, 5, 4, 1000.0, gpu.FOUND_ROWS, res[i][0].data_ptr(), res[i][1].data_ptr(),
                               st.data_ptr(), streams[i].cuda_stream)
    torch.cuda.synchronize(dev)
    for i in (0, 1):
        util.assert_same_lists(res[i][0].cpu().numpy(), res[i][1].view(torch.float32).cpu().numpy(), exp[i], f"stream {i}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("K", [256, 1024])
def test_running_bound_changes_survivors_not_lists(gpu, oracle, K, monkeypatch):
    """Option running_bound (default on): the scan's work entries share, per query, the smallest (threshold + coarse distance)
    any finished (item, chunk) has reported, and cut at it -- opportunistically, without waiting, so which rows survive the
    filter depends on timing and the lists must not: the oracle's, with the option on and off, for every found rule, several
    probing rounds (tiny cells) and repeated runs."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    t, ot, idx, qs = _fr_setup(gpu, oracle, K=K, dup_rows=3)
    for k, W, rule, sent in ((5, 6, 0, 1000.0), (10, 4, 1, 100.0), (5, 1, 2, 100.0), (32, 3, 0, 1000.0)):
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        for rb in (1, 0, 1, 1):
            idx.set_option("running_bound", rb)
            gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
            util.assert_same_lists(gi, gd, exp, f"running_bound {rb} K={K} k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("K", [256, 1024])
def test_in_flight_instantiations_match_oracle(gpu, oracle, K, monkeypatch):
    """scan_share > 1 -- the caller keeps batches in flight -- selects the small-footprint instantiations: ONE wave per query
    in the cell-selection plan (probe_plan2_kernel<0, false, 1>: seven candidates per round, the whole 300-dimensional query
    staged by 64 lanes) and in the merge (merge_refine_kernel<25, 12, 1>).  Same lists as the oracle; every probed row's bracket holds."""
    import torch
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    dev = torch.device("cuda", 0)
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=K)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qa = util.queries_from_corpus(N, 300, seed=23)
    dq = torch.from_numpy(qa).to(dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(dev)
    idx.set_option("scan_share", 4)
    for k, W, rule, sent in ((5, 10, 0, 1000.0), (10, 4, 1, 100.0), (5, 32, 0, 1000.0)):
        exp = oracle.ivfadc_search_many(ot, qa, k, W, sentinel=sent, found_rule=rule)
        for plan_waves in (0,):
            res = torch.zeros((2, 300, k), dtype=torch.int32, device=dev)
            with torch.cuda.stream(stream):
                idx.search_dev(dq.data_ptr(), 300, k, W, sent, rule, res[0].data_ptr(), res[1].data_ptr(), st.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            if int(st[0].item()) != 0:   # (a query that needs another probing round: finished by the synchronous call -- not this test's subject)
                st.zero_()
                continue
            util.assert_same_lists(res[0].cpu().numpy(), res[1].view(torch.float32).cpu().numpy(), exp,
                                   f"in flight K={K} k={k} W={W} rule={rule} plan_waves={plan_waves}")
    assert idx.bound_violations() == 0
    idx.set_option("scan_share", 1)
    idx.close()


def test_search_dev_can_be_captured_into_a_graph(gpu, oracle):
    """freddy_gpu_ivfadc_search_dev only enqueues (kernel launches and memsets on the caller's stream, no allocation once the stream's
    workspace exists, no synchronisation): a caller may capture a batch's chain into a hipGraph and replay it.  Replays with NEW
    queries in the same buffers give the oracle's lists.  (Round 6 measured the replay against the plain calls with four batches
    in flight: 0.0997 against 0.1003 ms per step -- the chain is not launch-bound; tools/lab/graph_replay.py.)"""
    import torch
    dev = torch.device("cuda", 0)
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=1024)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qa = util.queries_from_corpus(N, 300, seed=29)
    _, qb = util.queries_from_corpus(N, 300, seed=31)
    dq = torch.from_numpy(qa).to(dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    res = torch.zeros((2, 300, 5), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(dev)
    call = idx.bind_search_dev(dq.data_ptr(), 300, 5, 4, 1000.0, gpu.FOUND_ROWS, res[0].data_ptr(), res[1].data_ptr(), st.data_ptr(), stream.cuda_stream)
    call()                                  # (the stream's workspace is allocated by its first search)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        call()
    for qs in (qa, qb, qa):
        dq.copy_(torch.from_numpy(qs).to(dev))
        res.zero_()
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(stream):
            g.replay()
        torch.cuda.synchronize(dev)
        if int(st[0].item()) != 0:
            st.zero_()
            continue
        exp = oracle.ivfadc_search_many(ot, qs, 5, 4, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(res[0].cpu().numpy(), res[1].view(torch.float32).cpu().numpy(), exp, "graph replay")
    assert idx.bound_violations() == 0
    del g
    idx.close()


@pytest.mark.parametrize("share", [0, 3, 8])
def test_scan_share_gives_the_same_lists(gpu, oracle, share):
    """Option scan_share: the persistent scan on n_cus / share workgroups (the caller's statement of its
    batches in flight; values below 1 mean 1).  The number of workgroups that pull work entries changes nothing in the lists."""
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=1024)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx.set_option("scan_share", share)
    _, qs = util.queries_from_corpus(N, 400)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 6, sentinel=1000.0, found_rule=0)
    for _ in range(2):
        got_i, got_d = idx.search(qs, 5, 6, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
        util.assert_same_lists(got_i, got_d, exp, f"scan_share={share}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("K", [256, 1024])
def test_pq_batch_through_the_cell_grouped_scan(gpu, oracle, K):
    """Batches over the flat PQ table take the filter + refine scan over pseudo-lists of 4096 rows with zero centroids
    (option pq_fused, automatic from 16 queries on): same lists as pq_search (freddy.c:28-152) for every k
    the selection width admits, with the guard dist < sentinel biting, for a forced 3-query batch, with the path switched
    off, and after rows were appended (the pseudo-lists are rebuilt).  1 % of the rows are duplicates: equal distances."""
    N = 20000   # 4 full pseudo-lists + one of 3616 rows
    t = util.pq_tables(N=N, K=K)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 40, seed=21)
    qs[7] *= np.float32(20.0)   # a query farther than the sentinel 100.0 from everything: the empty list survives
    for k in (1, 5, 32):
        exp = np.stack([oracle.pq_search(ot, q, k) for q in qs])
        for mode in (-1, 0):
            idx.set_option("pq_fused", mode)
            gi, gd = idx.search(qs, k, sentinel=100.0)
            util.assert_same_lists(gi, gd, exp, f"pq batch K={K} k={k} pq_fused={mode}")
            if k == 5:
                assert (gi[7] == -1).all() and (gi[:7] >= 0).all()
    idx.set_option("pq_fused", 1)
    gi, gd = idx.search(qs[:3], 5, sentinel=100.0)
    util.assert_same_lists(gi, gd, np.stack([oracle.pq_search(ot, q, 5) for q in qs[:3]]), "forced 3-query batch")
    assert idx.bound_violations() == 0
    # append: rows 20001.. with the codes of existing rows (more equal distances)
    idx.set_option("pq_fused", -1)
    extra = 700
    new_ids = np.arange(N + 1, N + 1 + extra, dtype=np.int32)
    new_codes = np.ascontiguousarray(t["codes"][100:100 + extra])
    idx.append_rows(new_ids, codes=new_codes)
    ot2 = oracle.pq_table(t["codebook"], np.concatenate([t["ids"], new_ids]), np.concatenate([t["codes"], new_codes]))
    gi, gd = idx.search(qs, 5, sentinel=100.0)
    util.assert_same_lists(gi, gd, np.stack([oracle.pq_search(ot2, q, 5) for q in qs]), "pq batch after append_rows")
    assert idx.bound_violations() == 0
    idx.close()


def test_pq_subset_batch_through_the_cell_grouped_scan(gpu, oracle):
    """pq_search_in_batch (freddy.c:414-653) with an input set of at least one pseudo-list: the gathered rows get a view of
    their own (pseudo-lists, row terms, ids; refreshed by every call) and the batch takes the filter + refine scan.  Same
    lists with the path switched off, for a second call with a different set, and for a set below 4096 rows (old kernels)."""
    N = 20000
    t = util.pq_tables(N=N, K=256)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 24, seed=33)
    rng = np.random.default_rng(5)
    for n_sub, k in ((9000, 5), (4500, 17), (1500, 5)):
        targets = rng.choice(np.arange(1, N + 1), size=n_sub, replace=False).astype(np.int32)
        targets = np.concatenate([targets, targets[:40], np.array([N + 9, -2], np.int32)])   # duplicates + unknown ids
        exp = oracle.pq_search_in_batch(ot, qs, k, targets, use_target_lists=True)
        for mode in (-1, 0):
            idx.set_option("pq_fused", mode)
            gi, gd = idx.search(qs, k, sentinel=1000.0, subset_ids=targets)
            util.assert_same_lists(gi, gd, exp, f"pq_search_in_batch n_sub={n_sub} k={k} pq_fused={mode}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("kind", ["plain", "duplicate_centroids", "refine_all", "13000_cells"])
def test_coarse_filter_refine_beyond_1024_cells(gpu, oracle, kind, monkeypatch):
    """More than 1024 coarse cells: the plan streams a query's approximate distances twice (per-lane minima, then the
    candidates as a bitmap in LDS; probe_plan2_kernel<0, true>) instead of holding them in registers.  C = 1500 cells over
    30 000 rows; identical centroids (exactly equal coarse distances, hundreds of candidates when every cell is refined),
    probing rounds beyond the first, W up to 16; against the oracle and the all-exact coarse kernel."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", "1")
    from freddy_amd import index_build as ib
    # (13000_cells: the 40 M-row benchmark's cell count -- 102 tiles of 128 cells for the plan's two-level selection: the
    # threshold from the tiles' minima, candidates from the tiles that can reach it; mostly tiny or EMPTY lists)
    N, C = (60000, 13000) if kind == "13000_cells" else (30000, 1500)
    x = util.corpus(N)
    t = dict(ib.build_ivf_index(x, C=C, m=12, K=256, train_size=30000 if kind == "13000_cells" else 8000, iters=3, seed=4))
    coarse = t["coarse"].copy()
    if kind == "duplicate_centroids":
        coarse[700] = coarse[3]
        coarse[1499] = coarse[3]
        coarse[1100] = coarse[1024]
    t["coarse"] = coarse
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 70, seed=3)
    for k, W, rule, sent in ((5, 10, 0, 1000.0), (30, 3, 0, 1000.0), (10, 1, 1, 100.0), (5, 16, 0, 1000.0)):
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        idx.set_option("coarse_approx", 1)
        if kind == "refine_all":
            idx.set_option("check_brackets", 2)
            before = idx.coarse_bound_checked()
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"{kind}: streamed plan k={k} W={W} rule={rule}")
        if kind == "refine_all":
            assert idx.coarse_bound_checked() - before >= len(qs) * C
            idx.set_option("check_brackets", 0)
        idx.set_option("coarse_approx", 0)
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        util.assert_same_lists(gi, gd, exp, f"{kind}: all-exact coarse kernel k={k} W={W} rule={rule}")
    assert idx.bound_violations() == 0
    idx.close()


def test_more_streams_than_workspaces(gpu, oracle):
    """A handle keeps twelve workspaces, one per searching stream; further streams take over the least recently used slot
    after the device drained (workspace_for).  Fifteen streams, three rounds, interleaved: every stream's lists are the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qa = util.queries_from_corpus(N, 280)
    ns = 15
    qs = [np.ascontiguousarray(np.roll(qa, 11 * i, axis=0)) for i in range(ns)]
    exp = [oracle.ivfadc_search_many(ot, q, 5, 4, sentinel=1000.0, found_rule=0) for q in qs]
    dq = [torch.from_numpy(q).to(dev) for q in qs]
    res = [torch.zeros((2, 280, 5), dtype=torch.int32, device=dev) for _ in qs]
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in qs]
    torch.cuda.synchronize(dev)
    for rounds in range(3):
        for i in range(ns):
            with torch.cuda.stream(streams[i]):
                res[i].zero_()
                idx.search_dev(dq[i].data_ptr(), 280, 5, 4, 1000.0, gpu.FOUND_ROWS, res[i][0].data_ptr(), res[i][1].data_ptr(),
                               st.data_ptr(), streams[i].cuda_stream)
    torch.cuda.synchronize(dev)
    for i in range(ns):
        util.assert_same_lists(res[i][0].cpu().numpy(), res[i][1].view(torch.float32).cpu().numpy(), exp[i], f"stream {i} of {ns}")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("k_coarse", [32, 8])
def test_knn_join_device_traversal(gpu, oracle, k_coarse):
    """The multi-index traversal on the device (join_traverse_kernel: sorted keys + the running statistics sum, every
    stop checked by the host's libm) against the oracle's literal heap (index_utils.c:252-443): 1024 cells (the
    reference's default 2 x 32 multi-index) and 64; the same calls with every traversal forced onto the host heap."""
    N = 20000
    t = util.ivpq_tables(N=N, k_coarse=k_coarse)
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    _, qs = util.queries_from_corpus(N, 300, seed=41)
    rng = np.random.default_rng(12)
    targets = rng.choice(np.arange(1, N + 1), size=5000, replace=False).astype(np.int32)
    cases = [(5, 100, 20, 2, 0.8, True), (5, 3, 4, 0, 0.8, True), (5, 1, 3, 2, 0.3, True), (3, 1, 3, 0, 0.05, False),
             (5, 1000, 3, 0, 0.8, True), (10, 7, 5, 1, 0.95, True)]
    for host in (0, 1, 2):
        # 0: device traversal, the host's libm re-evaluates a stop only where the device's value is within 1e-5 of the
        # confidence; 2: the same with the margin at 2.0 -- EVERY proposed stop checked by libm; 1: the host heap
        idx.set_option("join_host_traversal", 1 if host == 1 else 0)
        idx.set_option("join_libm_margin_ppm", 2_000_000 if host == 2 else 10)
        handed_back = 0
        for (k, alpha, pvf, method, conf, tl) in cases:
            gi, gd, git = idx.knn_join(qs, k, targets, alpha, pvf, method, use_target_lists=tl, confidence=conf)
            exp, eit = oracle.ivpq_search_in(ot, qs, k, targets, alpha, pvf, method, use_target_lists=tl, confidence=conf)
            assert git == eit, (git, eit)
            util.assert_same_lists(gi, gd, exp, f"traversal host={host} Kc={k_coarse} k={k} alpha={alpha} conf={conf}")
            tr = idx.last_track()
            handed_back += tr["host_traversals"]
            if host == 1:
                assert tr["host_traversals"] >= qs.shape[0]
            if host == 2:
                assert tr["libm_checks"] >= 0.9 * qs.shape[0]
        if host != 1:   # equal float sums among a query's nearest cells are rare: the device keeps nearly every traversal
            assert handed_back < 0.2 * len(cases) * qs.shape[0], handed_back
    idx.close()


def test_knn_join_device_traversal_hands_ties_to_the_host(gpu, oracle):
    """Duplicate multi-index centroids make equal keys among the nearest cells: the heap's order among equals depends
    on its history, so the device hands those queries to the host heap -- the lists stay the oracle's."""
    N = 20000
    t = dict(util.ivpq_tables(N=N, k_coarse=32))
    coarse = t["coarse"].copy()            # [2][32][150]
    coarse[0, 5] = coarse[0, 3]; coarse[1, 9] = coarse[1, 2]; coarse[1, 10] = coarse[1, 2]
    t["coarse"] = coarse
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    _, qs = util.queries_from_corpus(N, 200, seed=43)
    targets = np.random.default_rng(13).choice(np.arange(1, N + 1), size=6000, replace=False).astype(np.int32)
    total = 0
    for (k, alpha, method, conf) in [(5, 50, 2, 0.8), (5, 2, 0, 0.9), (4, 1, 0, 0.2)]:
        gi, gd, git = idx.knn_join(qs, k, targets, alpha, 5, method, confidence=conf)
        exp, eit = oracle.ivpq_search_in(ot, qs, k, targets, alpha, 5, method, confidence=conf)
        assert git == eit
        util.assert_same_lists(gi, gd, exp, f"ties k={k} alpha={alpha} conf={conf}")
        total += idx.last_track()["host_traversals"]
    assert total > 0
    idx.close()


# ---------------------------------------------------------------------------------------
# every other index shape: the cell-grouped exact scan of multi.h (the reference's primitives are shape-generic,
# index_utils.c:445-455, :1126-1133; it ships m = 5 / K = 256 / 25-d / 32 cells: index_creation/config/ivfadc_complete_config.json)
# ---------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _shape_tables(d, m, K, C, N):
    import torch
    from freddy_amd import index_build as ib
    torch.manual_seed(0)
    x = ib.make_corpus(N, d=d, seed=17 + d + m, n_clusters=120, latent=min(10, d), dup_frac=0.01, device="cpu")
    t = ib.build_ivf_index(x, C=C, m=m, K=K, train_size=min(N, 6000), iters=4, seed=3)
    return x, t


@pytest.mark.parametrize("d,m,K,C", [(25, 5, 256, 32), (300, 6, 256, 24), (300, 10, 64, 40), (300, 15, 128, 16), (300, 30, 32, 32), (300, 12, 1024, 3000)])
def test_other_shapes_cell_grouped_scan_matches_oracle(gpu, oracle, d, m, K, C):
    """Shapes the filter + refine scan is not built for take ivf_multi_kernel (<= 8 items of a cell share its rows; exact LUTs
    interleaved in LDS; the sums are the reference's distances) from 256 (query, cell) items on; option fused = 1 forces it for
    small batches, fused = 0 keeps lut_build + adc_scan (the yardstick).  All of them: the oracle's lists -- every found rule,
    odd and even m (code dwords hold two positions), k up to 32, several probing rounds (tiny cells), duplicated rows (ties),
    and the filter scan's own shape with more than 1024 cells' worth of LDS-free... (m = 12 / K = 1024 stays with fused5.h: the
    last case checks that the dispatch leaves it there)."""
    N = 24000
    x, t = _shape_tables(d, m, K, C, N)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    rng = np.random.default_rng(3)
    qs = x[rng.choice(N, size=300, replace=False)].numpy().astype(np.float32)
    for k, W, rule, sent in ((5, 4, 0, 1000.0), (10, 3, 1, 100.0), (5, 1, 2, 100.0), (32, 2, 0, 1000.0)):
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule)
        kernels = {}
        for fused in (-1, 1, 0):
            idx.set_option("fused", fused)
            idx.profile_enable(True)
            gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
            kernels[fused] = set(idx.profile_read())
            idx.profile_enable(False)
            util.assert_same_lists(gi, gd, exp, f"shape d={d} m={m} K={K} C={C} fused={fused} k={k} W={W} rule={rule}")
        if (m, d // m) != (12, 25):
            assert "ivf_multi_scan" in kernels[1] and "ivf_multi_scan" not in kernels[0] and "adc_scan" in kernels[0], kernels
            if 300 * W >= 256:
                assert "ivf_multi_scan" in kernels[-1], kernels
        else:
            assert "ivf_multi_scan" not in kernels[1] and "ivf_filter" in kernels[1], kernels
    # few queries with fused = 1: entries of one or two items
    idx.set_option("fused", 1)
    for nq in (1, 3, 17):
        gi, gd = idx.search(qs[:nq], 5, 3, sentinel=1000.0, found_rule=0)
        exp = oracle.ivfadc_search_many(ot, qs[:nq], 5, 3, sentinel=1000.0, found_rule=0)
        util.assert_same_lists(gi, gd, exp, f"shape d={d} m={m} K={K} {nq} queries")
    idx.close()
