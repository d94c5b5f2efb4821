"""-m gpu parity at BASELINE.json's full sizes (configs[1..3]).  The corpora are built on the
GPU with torch (index creation is an input, not the path under test); the oracle is fast
enough on the host cores to be run at these sizes for the checked queries, so the bar stays
bit-exact (id, rank, distance)."""
import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def test_config2_pq_search_1m(oracle):
    """PQ search (pq_search / knn_in_pq), 1M x 300d, m=12, K=1024, k=5."""
    from freddy_amd import gpu, index_build as ib
    N = 1_000_000
    x = ib.make_corpus(N, seed=11, device=_dev())
    t = ib.build_pq_index(x, m=12, K=1024, train_size=100000, iters=6, seed=1)
    idx = gpu.PQIndex(t["codebook"], t["ids"], t["codes"])
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    qs = x[torch.arange(0, N, N // 16, device=x.device)[:16]].cpu().numpy()
    gi, gd = idx.search(qs, 5, sentinel=100.0)
    exp = np.stack([oracle.pq_search(ot, q, 5) for q in qs])
    util.assert_same_lists(gi, gd, exp, "config 2 pq_search (16 queries: the cell-grouped scan over 245 pseudo-lists)")
    assert (gi[:, 0] >= 1).all() and (np.diff(gd, axis=1) >= 0).all()
    assert idx.bound_violations() == 0
    idx.set_option("pq_fused", 0)
    gi0, gd0 = idx.search(qs, 5, sentinel=100.0)
    util.assert_same_lists(gi0, gd0, exp, "config 2 pq_search (LUT build + adc_scan_kernel)")
    idx.set_option("pq_fused", -1)
    gi1, gd1 = idx.search(qs[:1], 5, sentinel=100.0)
    util.assert_same_lists(gi1, gd1, exp[:1], "config 2 pq_search, one query")
    # knn_in_pq: 5,000 targets
    targets = np.random.default_rng(1).choice(np.arange(1, N + 1), 5000, replace=False).astype(np.int32)
    gi, gd = idx.search(qs, 5, sentinel=1000.0, subset_ids=targets)
    exp = oracle.pq_search_in_batch(ot, qs, 5, targets)
    util.assert_same_lists(gi, gd, exp, "config 2 pq_search_in_batch")
    assert np.isin(gi, targets).all()
    idx.close()


def test_config3_ivfadc_batch_3m(oracle):
    """IVFADC batch, 3M x 300d, 1000 coarse cells, nprobe=10, batch=1024 queries."""
    from freddy_amd import gpu, index_build as ib
    N = 3_000_000
    x = ib.make_corpus(N, seed=20260101, device=_dev())
    t = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    qid = np.sort(np.random.default_rng(7).choice(np.arange(1, N + 1), 1024, replace=False))
    qs = x[torch.from_numpy(qid - 1).to(x.device)].cpu().numpy()
    import os
    for W, sent, rule in [(10, 1000.0, 0), (1, 100.0, 1)]:
        gi, gd = idx.search(qs, 5, W, sentinel=sent, found_rule=rule)
        exp = oracle.ivfadc_search_many(ot, qs, 5, W, sentinel=sent, found_rule=rule, n_threads=os.cpu_count() or 1)
        util.assert_same_lists(gi, gd, exp, f"config 3 W={W}")
    # the query is an indexed vector: its own id must be in its result (at ADC distance rank 1 up to ties)
    assert (gi == qid[:, None]).any(axis=1).mean() > 0.99
    # The benchmarked instantiation in the every-row mode (ivf_filter_kernel<12, true>, all ~31 M probed rows
    # kept by the scan and refined by the merge): the proven bracket [d_lo, d_lo + E] is compared with the
    # reference's distance for EVERY probed row of this batch, and the lists must not change.
    gi0, gd0 = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=0)
    for kernel in (5,):   # the filter + refine scan
        idx.set_option("fused_kernel", kernel)
        gi1, gd1 = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=0)
        assert np.array_equal(gi0, gi1) and np.array_equal(gd0.view(np.uint32), gd1.view(np.uint32)), f"kernel {kernel}"
        idx.set_option("check_brackets", 1)
        before = idx.bound_checked()
        gi1, gd1 = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=0)
        rows = idx.last_scanned_rows()
        assert rows > 25_000_000
        assert idx.bound_checked() - before == rows, f"{idx.bound_checked() - before} brackets checked, {rows} rows probed"
        assert idx.bound_violations() == 0
        assert np.array_equal(gi0, gi1) and np.array_equal(gd0.view(np.uint32), gd1.view(np.uint32))
        idx.set_option("check_brackets", 0)
    idx.set_option("fused_kernel", 5)
    # the coarse filter + refine's bracket for EVERY (query, cell) pair of the batch (1024 x 1000)
    idx.set_option("check_brackets", 2)
    before = idx.coarse_bound_checked()
    gi2, gd2 = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=0)
    assert idx.coarse_bound_checked() - before == 1024 * 1000
    assert idx.bound_violations() == 0
    assert np.array_equal(gi0, gi2) and np.array_equal(gd0.view(np.uint32), gd2.view(np.uint32))
    idx.set_option("check_brackets", 0)
    idx.set_option("coarse_approx", 0)     # every coarse distance exact (coarse_tile_kernel)
    gi2, gd2 = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=0)
    assert np.array_equal(gi0, gi2) and np.array_equal(gd0.view(np.uint32), gd2.view(np.uint32))
    idx.set_option("coarse_approx", 1)
    # ... and the exact scan (the reference's arithmetic for every row, fused3.h) on the same batch
    idx.set_option("fused_kernel", 3)
    gi3, gd3 = idx.search(qs, 5, 10, sentinel=1000.0, found_rule=0)
    assert np.array_equal(gi0, gi3) and np.array_equal(gd0.view(np.uint32), gd3.view(np.uint32))
    idx.set_option("fused_kernel", 5)
    # The TIMED configuration of bench.py at full size: four batches in flight on four streams, scan_share = 4 (four
    # 64-workgroup scans side by side), the one-wave merge -- four DIFFERENT 1024-query sets, every list against the oracle.
    dev = x.device
    sets = []
    for i in range(4):
        ids_i = np.sort(np.random.default_rng(100 + i).choice(np.arange(1, N + 1), 1024, replace=False))
        sets.append(x[torch.from_numpy(ids_i - 1).to(dev)].contiguous())
    exp4 = [oracle.ivfadc_search_many(ot, s_.cpu().numpy(), 5, 10, sentinel=1000.0, found_rule=0, n_threads=os.cpu_count() or 1) for s_ in sets]
    res = [torch.zeros((2, 1024, 5), dtype=torch.int32, device=dev) for _ in sets]
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in sets]
    idx.set_option("scan_share", 4)
    torch.cuda.synchronize(dev)
    for rounds in range(5):
        for i in range(4):
            with torch.cuda.stream(streams[i]):
                res[i].zero_()
                idx.search_dev(sets[i].data_ptr(), 1024, 5, 10, 1000.0, gpu.FOUND_ROWS, res[i][0].data_ptr(), res[i][1].data_ptr(),
                               st.data_ptr(), streams[i].cuda_stream)
    torch.cuda.synchronize(dev)
    for i in range(4):
        util.assert_same_lists(res[i][0].cpu().numpy(), res[i][1].view(torch.float32).cpu().numpy(), exp4[i], f"config 3, four in flight, stream {i}")
    assert int(st[0].item()) == 0
    idx.set_option("scan_share", 1)
    # ... and the host-buffer pipeline (the call pg/freddy_srf.c makes) on all 4096 queries at once: four lanes
    allq = torch.cat(sets).cpu().numpy()
    gi4, gd4 = idx.search(allq, 5, 10, sentinel=1000.0, found_rule=0)
    util.assert_same_lists(gi4, gd4, {"id": np.concatenate([e["id"] for e in exp4]), "dist": np.concatenate([e["dist"] for e in exp4])},
                           "config 3, host-buffer pipeline, 4096 queries")
    assert idx.bound_violations() == 0
    idx.close()


def test_config4_knn_join_5000x100000(oracle):
    """knn_join (ivpq_search_in): 5,000 queries x 100,000 targets, k=5, alpha=100, pvf=20, method 2."""
    from freddy_amd import gpu, index_build as ib
    N = 1_000_000
    x = ib.make_corpus(N, seed=5, device=_dev())
    t = ib.build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=6, seed=3)
    idx = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    rng = np.random.default_rng(4)
    qid = rng.choice(np.arange(1, N + 1), 5000, replace=False)
    targets = rng.choice(np.arange(1, N + 1), 100000, replace=False).astype(np.int32)
    qs = t["vectors"][qid - 1]
    for method in (2, 0):
        gi, gd, git = idx.knn_join(qs, 5, targets, 100, 20, method)
        exp, eit = oracle.ivpq_search_in(ot, qs, 5, targets, 100, 20, method)
        assert git == eit
        util.assert_same_lists(gi, gd, exp, f"config 4 method={method}")
        assert np.isin(gi[gi >= 0], targets).all()
    idx.close()
