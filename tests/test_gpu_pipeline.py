"""-m gpu tests of the host-buffer pipeline behind freddy_gpu_ivfadc_search (the call pg/freddy_srf.c makes: one
synchronous call per batch, freddy.c:679-999) and of the multi-device handle: however a batch is cut into sub-batches,
spread over lanes or devices, the lists are the oracle's -- bit-exact (id, rank, distance)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from freddy_amd import gpu as g
    g.load()
    return g


@pytest.mark.parametrize("batch,lanes", [(128, 4), (500, 3), (333, 1), (4096, 4)])
def test_host_pipeline_cuts_change_nothing(gpu, oracle, batch, lanes):
    """2 000 queries through sub-batches of `batch` queries on `lanes` lanes; cell-grouped scan (K = 1024 shape)."""
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=1024)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx.set_option("pipeline_batch", batch)
    idx.set_option("pipeline_lanes", lanes)
    _, qs = util.queries_from_corpus(N, 2000, seed=3)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 6, sentinel=1000.0, found_rule=0, n_threads=8)
    for rep in range(2):   # (the second call reuses lanes, staging buffers and workspaces)
        got_i, got_d = idx.search(qs, 5, 6, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
        util.assert_same_lists(got_i, got_d, exp, f"pipeline batch={batch} lanes={lanes} call {rep}")
    # a different, shorter batch on the same handle afterwards
    got_i, got_d = idx.search(qs[100:731], 5, 6, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
    util.assert_same_lists(got_i, got_d, {"id": exp["id"][100:731], "dist": exp["dist"][100:731]}, "shorter batch")
    assert idx.bound_violations() == 0
    idx.close()


@pytest.mark.parametrize("Q", [512, 545, 1024, 1999])
def test_one_lane_call_launches_its_cell_selection_by_pieces(gpu, oracle, Q):
    """A host-buffer call of ONE sub-batch (>= 512 queries) stages its queries in four pieces of whole 32-query tiles and launches
    the MFMA cell selection + query table per piece (option coarse_pieces, round 6): lists equal with the option on and off, from
    pageable and from pinned query buffers, and equal to the oracle's; batch sizes that are not multiples of 32 or 16."""
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=1024)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, Q, seed=11)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 6, sentinel=1000.0, found_rule=0, n_threads=8)
    pb = gpu.PinnedBuffer(qs.shape)
    pb.array[:] = qs
    for pieces in (1, 0, 1):
        idx.set_option("coarse_pieces", pieces)
        for src, what in ((qs, "pageable"), (pb.array, "pinned")):
            got_i, got_d = idx.search(src, 5, 6, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
            util.assert_same_lists(got_i, got_d, exp, f"Q={Q} coarse_pieces={pieces} {what}")
    assert idx.bound_violations() == 0
    pb.close()
    idx.close()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_host_pipeline_with_extra_probing_rounds(gpu, oracle, fused, monkeypatch):
    """Cells with fewer than k rows force the reference's extra probing rounds (freddy.c:262, :377, :971) -- inside the
    pipeline they run where the host waits for the lane: sub-batches with and without stragglers, every found rule."""
    monkeypatch.setenv("FREDDY_GPU_FUSED", fused)
    N = 600
    x = util.corpus(N)
    from freddy_amd import index_build as ib
    t = ib.build_ivf_index(x, C=150, m=12, K=64, train_size=N, iters=3, seed=9)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx.set_option("pipeline_batch", 40)
    qs = np.ascontiguousarray(np.concatenate([x.numpy(), x.numpy()[::-1] * np.float32(1.02)]).astype(np.float32))   # 1 200 queries, 30 sub-batches
    for W, k, rule, sent in [(1, 10, 0, 1000.0), (2, 25, 0, 1000.0), (1, 10, 2, 100.0), (3, 40, 1, 100.0)]:
        gi, gd = idx.search(qs, k, W, sentinel=sent, found_rule=rule)
        exp = oracle.ivfadc_search_many(ot, qs, k, W, sentinel=sent, found_rule=rule, n_threads=8)
        util.assert_same_lists(gi, gd, exp, f"pipeline multi-round W={W} k={k} rule={rule}")
    idx.close()


def test_pinned_query_buffer_skips_the_staging_copy(gpu, oracle):
    """Queries written into a freddy_gpu_host_alloc buffer go to the device straight from there."""
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx.set_option("pipeline_batch", 256)
    _, qs = util.queries_from_corpus(N, 900, seed=5)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 4, sentinel=1000.0, found_rule=0, n_threads=8)
    pb = gpu.PinnedBuffer(qs.shape)
    pb.array[:] = qs
    got_i, got_d = idx.search(pb.array, 5, 4, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
    util.assert_same_lists(got_i, got_d, exp, "pinned query buffer")
    pb.array[:] = qs[::-1]   # the buffer is the caller's again after the call returned
    got_i, got_d = idx.search(pb.array, 5, 4, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
    util.assert_same_lists(got_i, got_d, {"id": exp["id"][::-1], "dist": exp["dist"][::-1]}, "pinned query buffer, second batch")
    idx.close()
    pb.close()


@pytest.mark.parametrize("n_dev", [2, 3])
def test_multi_device_handle_splits_the_batch(gpu, oracle, n_dev):
    """freddy_gpu_pin_ivf_multi with the same device listed n times (one GPU on the test box): the host batch is split
    contiguously over the replicas, one host thread each; insert_batch's device side reaches every replica."""
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=256)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"], devices=[0] * n_dev)
    assert idx.replicas == n_dev
    idx.set_option("pipeline_batch", 200)
    _, qs = util.queries_from_corpus(N, 1001, seed=9)
    exp = oracle.ivfadc_search_many(ot, qs, 5, 4, sentinel=1000.0, found_rule=0, n_threads=8)
    for rep in range(2):
        got_i, got_d = idx.search(qs, 5, 4, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
        util.assert_same_lists(got_i, got_d, exp, f"{n_dev} replicas, call {rep}")
    got_i, got_d = idx.search(qs[:3], 5, 4, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)   # fewer queries than 2 per replica: one device
    util.assert_same_lists(got_i, got_d, {"id": exp["id"][:3], "dist": exp["dist"][:3]}, "tiny batch")
    # new rows reach every replica: a query equal to a new row's vector must find it on whichever replica serves it
    new_ids = np.arange(N + 1, N + 9, dtype=np.int32)
    src = np.arange(0, 8 * 1000, 1000)
    row_of = {int(i): r for r, i in enumerate(t["ids"])}
    rows = np.array([row_of[int(i) + 1] for i in src])
    cell = np.searchsorted(t["list_off"], rows, side="right").astype(np.int32) - 1
    idx.append_rows(new_ids, coarse_id=cell, codes=t["codes"][rows])
    ids2 = np.concatenate([t["ids"], new_ids]); codes2 = np.concatenate([t["codes"], t["codes"][rows]]); cells2 = np.concatenate([
        np.repeat(np.arange(64, dtype=np.int32), np.diff(t["list_off"])), cell])
    order = np.lexsort((ids2, cells2))
    lo2 = np.concatenate([[0], np.cumsum(np.bincount(cells2, minlength=64))]).astype(np.int32)
    ot2 = oracle.ivf_table(t["coarse"], t["codebook"], lo2, ids2[order], codes2[order])
    exp2 = oracle.ivfadc_search_many(ot2, qs, 5, 4, sentinel=1000.0, found_rule=0, n_threads=8)
    got_i, got_d = idx.search(qs, 5, 4, sentinel=1000.0, found_rule=gpu.FOUND_ROWS)
    util.assert_same_lists(got_i, got_d, exp2, "after append_rows on every replica")
    assert idx.bound_violations() == 0
    idx.close()


def test_dev_entry_point_explicit_scan_share(gpu, oracle):
    """The *_dev contract: the caller states its batches in flight (option scan_share); four streams, four different
    batches, interleaved; then one batch at a time with scan_share = 1."""
    import torch
    dev = torch.device("cuda", 0)
    N = 60000
    t = util.ivf_tables(N=N, C=64, K=1024)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    qs = [util.queries_from_corpus(N, 400, seed=30 + i)[1] for i in range(4)]
    exp = [oracle.ivfadc_search_many(ot, q, 5, 6, sentinel=1000.0, found_rule=0, n_threads=8) for q in qs]
    dq = [torch.from_numpy(q).to(dev) for q in qs]
    res = [torch.zeros((2, 400, 5), dtype=torch.int32, device=dev) for _ in qs]
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in qs]
    torch.cuda.synchronize(dev)
    for share in (4, 1):
        idx.set_option("scan_share", share)
        for rounds in range(3):
            for i in range(4):
                with torch.cuda.stream(streams[i]):
                    res[i].zero_()
                    idx.search_dev(dq[i].data_ptr(), 400, 5, 6, 1000.0, gpu.FOUND_ROWS, res[i][0].data_ptr(), res[i][1].data_ptr(),
                                   st.data_ptr(), streams[i].cuda_stream)
                if share == 1:
                    torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        for i in range(4):
            util.assert_same_lists(res[i][0].cpu().numpy(), res[i][1].view(torch.float32).cpu().numpy(), exp[i], f"share {share}, stream {i}")
    assert idx.bound_violations() == 0
    idx.close()
