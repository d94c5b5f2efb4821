"""-m gpu tests through the host-side UDF mirror (libfreddy_host.so): the calls read like the
reference's SQL (`SELECT * FROM ivfadc_batch_search('{...}'::int[], k)`), results are compared
with the oracle drivers of the same UDFs."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

N = 20000


@pytest.fixture(scope="module")
def db(oracle):
    from freddy_amd import udf
    x = util.corpus(N).numpy()
    ids_all = np.arange(1, N + 1, dtype=np.int32)
    s = udf.Session()
    # rows are handed over shuffled: the host has to establish the canonical (id) order itself
    perm = np.random.default_rng(1).permutation(N)
    s.load_vecs_norm(ids_all[perm], x[perm])
    pq = util.pq_tables(N=N, K=256)
    s.load_pq(pq["codebook"], pq["ids"][perm], pq["codes"][perm])
    ivf = util.ivf_tables(N=N, C=32, K=256)
    cell_of = np.repeat(np.arange(32), np.diff(ivf["list_off"])).astype(np.int32)
    p2 = np.random.default_rng(2).permutation(N)
    s.load_ivfadc(ivf["coarse"], ivf["codebook"], ivf["ids"][p2], cell_of[p2], ivf["codes"][p2])
    iv = util.ivpq_tables(N=N)
    s.load_ivpq(iv["codebook"], iv["coarse"], iv["ids"][perm], iv["coarse_id"][perm], iv["codes"][perm], iv["stats"])
    tabs = dict(x=x, pq=oracle.pq_table(pq["codebook"], pq["ids"], pq["codes"]),
                ivf=oracle.ivf_table(ivf["coarse"], ivf["codebook"], ivf["list_off"], ivf["ids"], ivf["codes"]),
                ivpq=oracle.ivpq_table(iv["codebook"], iv["coarse"], iv["ids"], iv["coarse_id"], iv["codes"],
                                       iv["vectors"], iv["stats"]))
    yield s, tabs
    s.close()


def same(rows, exp):
    assert np.array_equal(rows["id"], exp["id"].ravel())
    assert np.array_equal(rows["distance"].view(np.uint32), exp["dist"].ravel().view(np.uint32))


def test_pq_search_and_pq_search_in(db, oracle):
    s, t = db
    q = t["x"][123]
    same(s.pq_search(q, 5), oracle.pq_search(t["pq"], q, 5))
    ids = [5, 17, 17, 900, 19999, 20001, -4]
    same(s.pq_search_in(q, 4, ids), oracle.pq_search_in(t["pq"], q, 4, ids))
    rows = s.pq_search_in(q, 3, [])                       # empty IN-list: k sentinel rows
    assert (rows["id"] == -1).all() and (rows["distance"] == np.float32(1000.0)).all()


def test_pq_search_in_batch(db, oracle):
    s, t = db
    qids = np.array([11, 500, 7777], np.int32)
    qs = t["x"][qids - 1]
    targets = np.arange(1, N + 1, 13).astype(np.int32)
    rows = s.pq_search_in_batch(qs, qids, 5, targets, True)
    exp = oracle.pq_search_in_batch(t["pq"], qs, 5, targets)
    same(rows, exp)
    assert rows["query_id"].tolist() == np.repeat(qids, 5).tolist()
    from freddy_amd import udf
    with pytest.raises(udf.FreddyError, match="Number of query vectors and query vector ids differs"):
        s.pq_search_in_batch(qs, qids[:2], 5, targets, True)


def test_ivfadc_search_uses_get_w(db, oracle):
    s, t = db
    q = t["x"][4321]
    for w in (1, 3, 8):
        s.set_w(w)
        same(s.ivfadc_search(q, 5), oracle.ivfadc_search(t["ivf"], q, 5, w))
    s.set_w(3)


def test_ivfadc_batch_search(db, oracle):
    """query ids arrive unordered, with a duplicate and an unknown id: rows come back in table
    order of the found ids (freddy.c:767-804, 986)."""
    s, t = db
    asked = np.array([900, 17, 17, 15000, 25000, 3], np.int32)
    rows = s.ivfadc_batch_search(asked, 5)
    found = np.array([3, 17, 900, 15000], np.int32)
    assert rows["query_id"].tolist() == np.repeat(found, 5).tolist()
    same(rows, oracle.ivfadc_batch_search(t["ivf"], t["x"][found - 1], 5))


@pytest.mark.parametrize("method", [0, 1, 2])
def test_ivpq_search_in_and_knn_join(db, oracle, method):
    s, t = db
    qids = np.arange(100, 160, dtype=np.int32)
    qs = t["x"][qids - 1]
    targets = np.random.default_rng(3).choice(np.arange(1, N + 1), 3000, replace=False).astype(np.int32)
    rows = s.ivpq_search_in(qs, qids, 5, targets, 10, 4, method, True, 0.8, 10000000)
    exp, _ = oracle.ivpq_search_in(t["ivpq"], qs, 5, targets, 10, 4, method)
    same(rows, exp)
    s.set_alpha(10); s.set_pvf(4); s.set_method_flag(method)
    same(s.knn_join(qs, qids, 5, targets), exp)
    from freddy_amd import udf
    with pytest.raises(udf.FreddyError, match="Unknown computation method"):
        s.ivpq_search_in(qs, qids, 5, targets, 10, 4, 7, True, 0.8, 10000000)


def test_k_nearest_neighbour_and_knn_in_exact(db, oracle):
    """Next row 8f-1: `SELECT * FROM k_nearest_neighbour(vec, k)` / `knn_in_exact(vec, k, '{..}'::int[])`
    (freddy--0.0.1.sql:426-439, 1041-1054), keyed by row id."""
    s, t = db
    ids_all = np.arange(1, N + 1, dtype=np.int32)
    q = t["x"][777]
    rows = s.k_nearest_neighbour(q, 7)
    exp = oracle.exact_knn(t["x"], ids_all, q, 7)
    same(rows, exp)
    assert rows["id"][0] == 778
    sub = [5, 17, 17, 900, 19999, 20001, -4]
    rows = s.knn_in_exact(q, 10, sub)
    exp = oracle.exact_knn(t["x"], ids_all, q, 10, sub)
    assert len(rows) == 4                                   # FETCH FIRST returns only existing rows
    same(rows, exp)
    assert len(s.knn_in_exact(q, 3, [])) == 0


def test_grouping_pq(db, oracle):
    """Next row 8f-3: `SELECT * FROM grouping_pq('{ids}'::int[], '{groups}'::int[])` (freddy.c:1176-1401)."""
    s, t = db
    from freddy_amd import udf
    groups = np.array([9000, 12, 4400, 17017], np.int32)          # unsorted: the UDF sorts them (freddy.c:1241)
    asked = np.concatenate([np.arange(1, N + 1, 7), [5, 5, N + 10, -1]]).astype(np.int32)
    rows = s.grouping_pq(asked, groups)
    sg = np.sort(groups)
    ids, grp = oracle.grouping_pq(t["pq"], t["x"][sg - 1], asked)
    assert np.array_equal(rows["id"], ids)
    assert np.array_equal(rows["group_id"], np.where(grp >= 0, sg[np.maximum(grp, 0)], -1))
    assert set(np.unique(rows["group_id"]).tolist()) <= set(sg.tolist())
    assert len(np.unique(rows["group_id"])) > 1
    assert len(s.grouping_pq([], groups)) == 0
    with pytest.raises(udf.FreddyError, match="Group ids do not exist"):
        s.grouping_pq(asked, [12, N + 5])
    with pytest.raises(udf.FreddyError, match="Group ids do not exist"):
        s.grouping_pq(asked, [12, 12])


@pytest.mark.parametrize("which", ["pq", "ivfadc"])
def test_analogy_3cosadd(db, oracle, which):
    """analogy_3cosadd_pq / _ivfadc (freddy--0.0.1.sql:1317-1346, 1428-1460) composed from the oracle's
    pieces: vec ops, pq_search / ivfadc_search with k = get_pvf() + 3, exact re-ranking."""
    s, t = db
    x = t["x"]
    s.set_pvf(6); s.set_w(3)
    for (i1, i2, i3) in ((10, 200, 3000), (4321, 4322, 77), (15000, 8, 9)):
        raw = oracle.vec_plus(oracle.vec_minus(x[i3 - 1], x[i1 - 1]), x[i2 - 1])
        unit = oracle.vec_normalize(raw)
        cands = oracle.pq_search(t["pq"], unit, 9) if which == "pq" else oracle.ivfadc_search(t["ivf"], unit, 9, 3)
        best, bid = None, -1
        for cid in cands["id"].tolist():
            if cid < 0 or cid in (i1, i2, i3):
                continue
            sim = oracle.cosine_similarity_bytea(raw, x[cid - 1])
            if best is None or sim > best or (sim == best and cid < bid):
                best, bid = sim, cid
        got = s.analogy_3cosadd_pq(i1, i2, i3) if which == "pq" else s.analogy_3cosadd_ivfadc(i1, i2, i3)
        assert got == bid and got > 0
    assert s.analogy_3cosadd_pq(1, 2, N + 99) == -1              # unknown word: NULL
    s.set_pvf(20)


def test_index_file_import(db, oracle, tmp_path):
    """Next row 8f-2: tables -> FRDYIDX1 file -> a fresh session pinned from the file answers like the
    session loaded from the arrays."""
    from freddy_amd import udf
    s, t = db
    pq = util.pq_tables(N=N, K=256)
    ivf = util.ivf_tables(N=N, C=32, K=256)
    iv = util.ivpq_tables(N=N)
    ids_all = np.arange(1, N + 1, dtype=np.int32)
    path = tmp_path / "all.fidx"
    udf.write_index_file(path, udf.table_arrays(pq=pq, ivfadc=ivf, ivpq=iv, vecs_norm=(ids_all, t["x"])))
    s2 = udf.Session()
    s2.import_index(path)
    q = t["x"][321]
    same(s2.pq_search(q, 5), oracle.pq_search(t["pq"], q, 5))
    same(s2.ivfadc_search(q, 5), oracle.ivfadc_search(t["ivf"], q, 5, 3))
    qids = np.arange(100, 140, dtype=np.int32)
    targets = np.arange(1, N + 1, 9).astype(np.int32)
    exp, _ = oracle.ivpq_search_in(t["ivpq"], t["x"][qids - 1], 5, targets, 10, 4, 2)
    same(s2.ivpq_search_in(t["x"][qids - 1], qids, 5, targets, 10, 4, 2, True, 0.8, 10000000), exp)
    rows = s2.ivfadc_batch_search([17, 900], 5)
    same(rows, oracle.ivfadc_batch_search(t["ivf"], t["x"][[16, 899]], 5))
    s2.close()


@pytest.mark.parametrize("which", ["pq", "ivfadc"])
def test_k_nearest_neighbour_callers(db, oracle, which):
    """The plpgsql callers of pq_search / ivfadc_search (freddy--0.0.1.sql:520-531, 575-591, 610-641):
    similarity from the emitted distance, and post verification by exact cosine similarity."""
    s, t = db
    s.set_w(3); s.set_pvf(4)
    q = t["x"][2468]
    ids_all = t["x"]
    cand = (lambda k: oracle.pq_search(t["pq"], q, k)) if which == "pq" else (lambda k: oracle.ivfadc_search(t["ivf"], q, k, 3))
    plain = s.k_nearest_neighbour_pq(q, 6) if which == "pq" else s.k_nearest_neighbour_ivfadc(q, 6)
    exp = [e for e in cand(6) if e["id"] >= 0]
    assert plain["id"].tolist() == [int(e["id"]) for e in exp]
    sims = [np.float32(1.0 - np.float64(oracle.emit_roundtrip(e["dist"])) / 2.0) for e in exp]
    assert np.array_equal(plain["distance"].view(np.uint32), np.array(sims, np.float32).view(np.uint32))
    pv = s.k_nearest_neighbour_pq_pv(q, 5) if which == "pq" else s.k_nearest_neighbour_ivfadc_pv(q, 5)
    c = [int(e["id"]) for e in cand(20) if e["id"] >= 0]
    scored = sorted(((-float(oracle.cosine_similarity_bytea(q, ids_all[i - 1])), i) for i in c))[:5]
    assert pv["id"].tolist() == [i for _, i in scored]
    assert np.array_equal(pv["distance"].view(np.uint32),
                          np.array([oracle.cosine_similarity_bytea(q, ids_all[i - 1]) for _, i in scored], np.float32).view(np.uint32))
    assert pv["id"][0] == 2469                                   # the query's own row wins the exact re-ranking
    s.set_pvf(20)


def test_knn_in_pq_and_ivfadc_batch_callers(db, oracle):
    """knn_in_pq / k_nearest_neighbour_ivfadc_batch (freddy--0.0.1.sql:830-843, 535-553): the SRF rows with
    similarity = (1.0 - (emitted distance / 2.0))::float4, filler rows dropped by the joins."""
    s, t = db
    sim = lambda d: np.float32(1.0 - np.float64(oracle.emit_roundtrip(d)) / 2.0)
    q = t["x"][55]
    ids = [5, 17, 900, 19999, 20001]
    rows = s.knn_in_pq(q, 6, ids)
    exp = [e for e in oracle.pq_search_in(t["pq"], q, 6, ids) if e["id"] >= 0]
    assert len(rows) == 4 and rows["id"].tolist() == [int(e["id"]) for e in exp]
    assert np.array_equal(rows["distance"].view(np.uint32), np.array([sim(e["dist"]) for e in exp], np.float32).view(np.uint32))
    rows = s.k_nearest_neighbour_ivfadc_batch([900, 3, 17], 4)
    found = np.array([3, 17, 900], np.int32)
    exp = oracle.ivfadc_batch_search(t["ivf"], t["x"][found - 1], 4)
    keep = exp["id"].ravel() >= 0
    assert rows["query_id"].tolist() == np.repeat(found, 4)[keep].tolist()
    assert rows["id"].tolist() == exp["id"].ravel()[keep].tolist()
    assert np.array_equal(rows["distance"].view(np.uint32),
                          np.array([sim(d) for d in exp["dist"].ravel()[keep]], np.float32).view(np.uint32))


# ---------------------------------------------------------------------------------------
# insert_batch: quantisation on the device, HBM index mutation
# ---------------------------------------------------------------------------------------
def _new_vectors(n, seed=5):
    x = util.corpus(20000).numpy()
    rng = np.random.default_rng(seed)
    v = x[rng.choice(20000, n, replace=False)] + 0.05 * rng.standard_normal((n, 300)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_insert_quantize_matches_oracle(oracle):
    """Codes / coarse ids of new vectors exactly as the reference's insert_batch finds them (freddy.c:1557-1623,
    index_utils.c:925-939): PQ, coarse + residual, ivpq, multi-index coarse codes -- bit for bit."""
    from freddy_amd import gpu
    pq, ivf, ivpq = util.pq_tables(), util.ivf_tables(), util.ivpq_tables()
    v = _new_vectors(37)
    got = gpu.insert_quantize(v, pq_codebook=pq["codebook"], residual_codebook=ivf["codebook"], coarse=ivf["coarse"],
                              ivpq_codebook=ivpq["codebook"], coarse_multi=ivpq["coarse"])
    m, K, s_ = pq["codebook"].shape
    _, _, codes, _ = oracle.update_codebook(pq["codebook"], np.ones(m * K, np.int32), v)
    assert np.array_equal(codes, got["pq_codes"])
    cq, res = oracle.insert_coarse(ivf["coarse"], v)
    assert np.array_equal(cq, got["coarse_id"])
    m, K, s_ = ivf["codebook"].shape
    _, _, codes, _ = oracle.update_codebook(ivf["codebook"], np.ones(m * K, np.int32), res)
    assert np.array_equal(codes, got["residual_codes"])
    m, K, s_ = ivpq["codebook"].shape
    _, _, codes, _ = oracle.update_codebook(ivpq["codebook"], np.ones(m * K, np.int32), v)
    assert np.array_equal(codes, got["ivpq_codes"])
    multi = oracle.insert_coarse_multi(ivpq["coarse"], v)
    P = ivpq["coarse"].shape[0]
    assert np.array_equal(multi, got["coarse_multi_codes"][:, 0].astype(np.int32) + P * got["coarse_multi_codes"][:, 1].astype(np.int32))
    # a vector 100 or farther from every centroid: undefined in the reference, an error here
    with pytest.raises(gpu.FreddyGpuError):
        gpu.insert_quantize(v * np.float32(40.0), pq_codebook=pq["codebook"])


def test_append_rows_and_update_codebook_equal_a_fresh_pin(oracle):
    """HBM mutation: rows appended to pinned pq / ivf / ivpq / vector indexes and a replaced codebook give
    the same results as pinning the final tables from scratch (and as the oracle on those tables)."""
    from freddy_amd import gpu
    N, n_new = 20000, 333
    pq, ivf, ivpq = util.pq_tables(), util.ivf_tables(), util.ivpq_tables()
    x = util.corpus(N).numpy()
    v = _new_vectors(n_new, seed=9)
    new_ids = np.arange(N + 5, N + 5 + 2 * n_new, 2, dtype=np.int32)
    q = gpu.insert_quantize(v, pq_codebook=pq["codebook"], residual_codebook=ivf["codebook"], coarse=ivf["coarse"],
                            ivpq_codebook=ivpq["codebook"], coarse_multi=ivpq["coarse"])
    _, qs = util.queries_from_corpus(N, 40)
    qs = np.concatenate([qs, v[:24]])
    rng = np.random.default_rng(2)

    def nudged(cb):   # a "running mean" style change of some entries
        out = cb.copy()
        out[:, ::3] += np.float32(0.01) * rng.standard_normal(out[:, ::3].shape).astype(np.float32)
        return out

    # pq
    idx = gpu.PQIndex(pq["codebook"], pq["ids"], pq["codes"])
    idx.append_rows(new_ids, codes=q["pq_codes"])
    cb2 = nudged(pq["codebook"])
    idx.update_codebook(cb2)
    ids2, codes2 = np.concatenate([pq["ids"], new_ids]), np.concatenate([pq["codes"], q["pq_codes"]])
    ot = oracle.pq_table(cb2, ids2, codes2)
    gi, gd = idx.search(qs, 7, sentinel=100.0)
    util.assert_same_lists(gi, gd, np.stack([oracle.pq_search(ot, qq, 7) for qq in qs]), "pq after append")
    gi, gd = idx.search(qs, 5, sentinel=1000.0, subset_ids=new_ids[::2])
    util.assert_same_lists(gi, gd, oracle.pq_search_in_batch(ot, qs, 5, new_ids[::2]), "pq_search_in on appended ids")
    idx.close()
    # ivf: appended rows join the END of their cell's list (two batches: the second finds partially filled blocks)
    idx = gpu.IVFIndex(ivf["coarse"], ivf["codebook"], ivf["list_off"], ivf["ids"], ivf["codes"])
    h = n_new // 2
    idx.append_rows(new_ids[:h], coarse_id=q["coarse_id"][:h], codes=q["residual_codes"][:h])
    idx.append_rows(new_ids[h:], coarse_id=q["coarse_id"][h:], codes=q["residual_codes"][h:])
    cb2 = nudged(ivf["codebook"])
    idx.update_codebook(cb2)
    C = len(ivf["list_off"]) - 1
    cell_old = np.repeat(np.arange(C), np.diff(ivf["list_off"]))
    cell_all = np.concatenate([cell_old, q["coarse_id"]])
    ids_all, codes_all = np.concatenate([ivf["ids"], new_ids]), np.concatenate([ivf["codes"], q["residual_codes"]])
    order = np.lexsort((ids_all, cell_all))
    lo = np.zeros(C + 1, np.int32)
    lo[1:] = np.cumsum(np.bincount(cell_all, minlength=C))
    ot = oracle.ivf_table(ivf["coarse"], cb2, lo, ids_all[order], codes_all[order])
    fresh = gpu.IVFIndex(ivf["coarse"], cb2, lo, ids_all[order], codes_all[order])
    for fused in (1, 0):
        idx.set_option("fused", fused)
        fresh.set_option("fused", fused)
        for k, W in ((5, 3), (10, 1)):
            gi, gd = idx.search(qs, k, W)
            fi, fd = fresh.search(qs, k, W)
            util.assert_same_lists(gi, gd, oracle.ivfadc_search_many(ot, qs, k, W), f"ivf after append fused={fused} W={W}")
            assert np.array_equal(gi, fi) and np.array_equal(gd.view(np.uint32), fd.view(np.uint32))
    assert idx.bound_violations() == 0
    assert (gi >= N + 5).any(), "appended rows must be reachable"
    idx.close(); fresh.close()
    # ivpq (kNN-join) with vectors
    P = ivpq["coarse"].shape[0]
    multi = q["coarse_multi_codes"][:, 0].astype(np.int32) + ivpq["coarse"].shape[1] * q["coarse_multi_codes"][:, 1].astype(np.int32)
    idx = gpu.IVPQIndex(ivpq["codebook"], ivpq["coarse"], ivpq["ids"], ivpq["coarse_id"], ivpq["codes"], ivpq["vectors"], ivpq["stats"])
    idx.append_rows(new_ids, coarse_id=multi, codes=q["ivpq_codes"], vectors=v)
    ot = oracle.ivpq_table(ivpq["codebook"], ivpq["coarse"], np.concatenate([ivpq["ids"], new_ids]),
                           np.concatenate([ivpq["coarse_id"], multi]), np.concatenate([ivpq["codes"], q["ivpq_codes"]]),
                           np.concatenate([ivpq["vectors"], v]), ivpq["stats"])
    targets = np.concatenate([rng.choice(ivpq["ids"], 600, replace=False), new_ids[::3]]).astype(np.int32)
    for method in (0, 2):
        gi, gd, it = idx.knn_join(qs, 5, targets, 3, 4, method)
        exp, eit = oracle.ivpq_search_in(ot, qs, 5, targets, 3, 4, method)
        assert it == eit
        util.assert_same_lists(gi, gd, exp, f"knn_join after append, method {method}")
    idx.close()
    # raw vectors (exact kNN)
    base_ids = np.arange(1, N + 1, dtype=np.int32)
    idx = gpu.VectorIndex(base_ids, x)
    idx.append_rows(new_ids, vectors=v)
    fresh = gpu.VectorIndex(np.concatenate([base_ids, new_ids]), np.concatenate([x, v]))
    gi, gs = idx.search(qs, 6)
    fi, fs = fresh.search(qs, 6)
    assert np.array_equal(gi, fi) and np.array_equal(gs.view(np.uint32), fs.view(np.uint32))
    gi, gs = idx.search(qs, 4, subset_ids=new_ids[:50])
    fi, fs = fresh.search(qs, 4, subset_ids=new_ids[:50])
    assert np.array_equal(gi, fi) and np.array_equal(gs.view(np.uint32), fs.view(np.uint32))
    idx.close(); fresh.close()
    # ids must ascend beyond what is pinned
    idx = gpu.PQIndex(pq["codebook"], pq["ids"], pq["codes"])
    with pytest.raises(gpu.FreddyGpuError):
        idx.append_rows(np.array([5], np.int32), codes=q["pq_codes"][:1])
    idx.close()


def _fresh_session():
    from freddy_amd import udf
    x = util.corpus(N).numpy()
    ids_all = np.arange(1, N + 1, dtype=np.int32)
    s = udf.Session()
    s.load_vecs_norm(ids_all, x)
    pq, ivf, iv = util.pq_tables(N=N, K=256), util.ivf_tables(N=N, C=32, K=256), util.ivpq_tables(N=N)
    cell_of = np.repeat(np.arange(32), np.diff(ivf["list_off"])).astype(np.int32)
    s.load_pq(pq["codebook"], pq["ids"], pq["codes"])
    s.load_ivfadc(ivf["coarse"], ivf["codebook"], ivf["ids"], cell_of, ivf["codes"])
    s.load_ivpq(iv["codebook"], iv["coarse"], iv["ids"], iv["coarse_id"], iv["codes"], iv["stats"])
    return s, x, pq, ivf, iv, cell_of


def test_insert_batch_udf(oracle):
    """insert_batch (freddy.c:1403-1658) through the host mirror: afterwards every search answers as the oracle
    does on the tables the reference would hold -- new rows with max(id) + 1, codebook entries nudged by
    updateCodebook's own arithmetic (index_utils.c:908-991), every float stored through "%f"."""
    s, x, pq, ivf, iv, cell_of = _fresh_session()
    rng = np.random.default_rng(3)
    counts = {name: rng.integers(1, 400, size=t["codebook"].shape[:2]).astype(np.int32) for name, t in (("pq", pq), ("ivf", ivf), ("iv", iv))}
    s.set_codebook_counts(0, counts["pq"]); s.set_codebook_counts(1, counts["ivf"]); s.set_codebook_counts(2, counts["iv"])
    v = _new_vectors(29, seed=13)
    new_ids = s.insert_batch(v)
    assert new_ids.tolist() == list(range(N + 1, N + 30))
    stored = oracle.text_roundtrip(v)
    # --- the tables after the reference's insert_batch ---
    pq_cb, _, pq_codes, pq_incs = oracle.update_codebook(pq["codebook"], counts["pq"].reshape(-1), v)
    cq, res = oracle.insert_coarse(ivf["coarse"], v)
    res_cb, _, res_codes, _ = oracle.update_codebook(ivf["codebook"], counts["ivf"].reshape(-1), res)
    iv_cb, _, iv_codes, _ = oracle.update_codebook(iv["codebook"], counts["iv"].reshape(-1), v)
    multi = oracle.insert_coarse_multi(iv["coarse"], v)
    assert pq_incs.sum() == 29 * pq["codebook"].shape[0] and (pq_cb != pq["codebook"]).any()
    ot_pq = oracle.pq_table(pq_cb, np.concatenate([pq["ids"], new_ids]), np.concatenate([pq["codes"], pq_codes]))
    cell_all = np.concatenate([cell_of, cq])
    ids_all, codes_all = np.concatenate([ivf["ids"], new_ids]), np.concatenate([ivf["codes"], res_codes])
    order = np.lexsort((ids_all, cell_all))
    lo = np.zeros(33, np.int32); lo[1:] = np.cumsum(np.bincount(cell_all, minlength=32))
    ot_ivf = oracle.ivf_table(ivf["coarse"], res_cb, lo, ids_all[order], codes_all[order])
    ot_iv = oracle.ivpq_table(iv_cb, iv["coarse"], np.concatenate([iv["ids"], new_ids]), np.concatenate([iv["coarse_id"], multi]),
                              np.concatenate([iv["codes"], iv_codes]), np.concatenate([x, stored]), iv["stats"])
    # --- searches ---
    for q in list(v[:6]) + [x[77], x[4000]]:
        same(s.pq_search(q, 6), oracle.pq_search(ot_pq, q, 6))
        same(s.ivfadc_search(q, 6), oracle.ivfadc_search(ot_ivf, q, 6, 3))
    qs = np.concatenate([v[:10], x[100:110]])
    qids = np.arange(1, 21, dtype=np.int32)
    targets = np.concatenate([np.arange(1, N + 1, 23), new_ids]).astype(np.int32)
    for method in (0, 2):
        rows = s.ivpq_search_in(qs, qids, 4, targets, 3, 5, method, True, 0.8, 10000000)
        exp, _ = oracle.ivpq_search_in(ot_iv, qs, 4, targets, 3, 5, method)
        same(rows, exp)
    hit = s.pq_search(v[0], 6)["id"]
    assert (hit > N).any(), "an inserted vector must find its own row"
    # exact kNN sees the rows as stored ("%f" text, six decimals)
    got = s.k_nearest_neighbour(stored[3], 3)
    exp = oracle.exact_knn(np.concatenate([x, stored]), np.arange(1, N + 30, dtype=np.int32), stored[3], 3)
    assert np.array_equal(got["id"], exp["id"]) and np.array_equal(got["distance"].view(np.uint32), exp["dist"].view(np.uint32))
    # a second batch continues the id sequences and meets the updated codebooks
    new2 = s.insert_batch(v[:3] * np.float32(0.999))
    assert new2.tolist() == [N + 30, N + 31, N + 32]
    s.close()


def _sim_of(oracle, dist):
    return np.float32(1.0 - float(oracle.emit_roundtrip(dist)) / 2.0)


def _cluster_reference(knn_rows, vectors, n, k, draws):
    """generic_cluster (freddy--0.0.1.sql:1086-1209) restated; knn_rows(centroids) -> [(similarity, qid, tid)]."""
    it = iter(draws)
    pick = lambda upper: int(min(max(np.rint(next(it) * upper + 0.5), 1), upper))
    d = vectors.shape[1]
    centroids = np.stack([vectors[pick(n) - 1] for _ in range(k)]).astype(np.float32)
    clusters, lens = np.zeros(n, np.int32), np.zeros(k, np.int64)
    for J in range(1, 11):
        processed = np.zeros(n, bool)
        rows = sorted(knn_rows(centroids), key=lambda r: (-r[0], r[1], r[2]))
        for sim, qid, tid in rows:
            if not processed[tid - 1]:
                clusters[tid - 1] = qid; lens[qid - 1] += 1; processed[tid - 1] = True
        if J < 10:
            for I in range(1, k + 1):
                if lens[I - 1] == 0:
                    for _ in range(10):
                        pick(n)
                else:
                    members = np.flatnonzero(clusters == I)
                    samples = []
                    for _ in range(10):
                        xx = pick(int(lens[I - 1]))
                        if xx <= len(members):
                            samples.append(vectors[members[xx - 1]])
                    if samples:
                        c = np.zeros(d, np.float32)
                        for smp in samples:
                            c = (c + smp / np.float32(len(samples))).astype(np.float32)
                        centroids[I - 1] = c
                    lens[I - 1] = 0
    return clusters


@pytest.mark.parametrize("which", ["exact", "pq", "ivpq"])
def test_cluster_functions(db, oracle, which):
    """cluster_exact / cluster_pq / cluster_ivpq = generic_cluster with the three batch kNN functions; the random()
    draws are supplied, so the whole 10-round k-means must reproduce the restatement built from the oracle."""
    s, t = db
    x = t["x"]
    rng = np.random.default_rng(31)
    tokens = np.sort(rng.choice(np.arange(1, N + 1), 90, replace=False)).astype(np.int32)
    vecs = x[tokens - 1]
    k = 5
    draws = rng.random(k + 9 * k * 10)
    n = len(tokens)
    s.set_alpha(3); s.set_pvf(4); s.set_method_flag(0)

    def rows_exact(cent):
        out = []
        for qi, c in enumerate(cent):
            e = oracle.exact_knn(x, np.arange(1, N + 1, dtype=np.int32), c, n, tokens)
            out += [(np.float32(e["dist"][r]), qi + 1, int(np.searchsorted(tokens, e["id"][r])) + 1) for r in range(len(e))]
        return out

    def rows_pq(cent):
        e = oracle.pq_search_in_batch(t["pq"], cent, n, tokens)
        return [(_sim_of(oracle, e["dist"][qi, r]), qi + 1, int(np.searchsorted(tokens, e["id"][qi, r])) + 1)
                for qi in range(len(cent)) for r in range(n) if e["id"][qi, r] >= 0]

    def rows_ivpq(cent):
        e, _ = oracle.ivpq_search_in(t["ivpq"], cent, n, tokens, 3, 4, 0)
        e = e.reshape(len(cent), n)
        return [(_sim_of(oracle, e["dist"][qi, r]), qi + 1, int(np.searchsorted(tokens, e["id"][qi, r])) + 1)
                for qi in range(len(cent)) for r in range(n) if e["id"][qi, r] >= 0]

    exp = _cluster_reference({"exact": rows_exact, "pq": rows_pq, "ivpq": rows_ivpq}[which], vecs, n, k, draws)
    got = getattr(s, "cluster_" + which)(tokens, k, draws)
    assert np.array_equal(got, exp), f"cluster_{which}: {np.flatnonzero(got != exp)[:10]}"
    assert set(np.unique(got)) <= set(range(0, k + 1)) and (got > 0).mean() > 0.9
    s.set_pvf(20)


@pytest.mark.parametrize("which", ["pq", "ivpq"])
def test_analogy_3cosadd_in(db, oracle, which):
    """analogy_3cosadd_in_pq / _in_ivpq (freddy--0.0.1.sql:1348-1426): candidates from pq_search_in(q, pvf + 3, ids)
    resp. ivpq_search_in(ARRAY[q], '{0}', 4, ids, ...), exact re-ranking, the three inputs excluded."""
    s, t = db
    x = t["x"]
    s.set_pvf(6); s.set_alpha(3); s.set_method_flag(0)
    rng = np.random.default_rng(8)
    input_ids = np.sort(rng.choice(np.arange(1, N + 1), 400, replace=False)).astype(np.int32)
    for (i1, i2, i3) in ((10, 200, 3000), (4321, 4322, 77)):
        raw = oracle.vec_plus(oracle.vec_minus(x[i3 - 1], x[i1 - 1]), x[i2 - 1])
        unit = oracle.vec_normalize(raw)
        if which == "pq":
            cands = oracle.pq_search_in(t["pq"], unit, 9, input_ids)["id"].tolist()
        else:
            e, _ = oracle.ivpq_search_in(t["ivpq"], unit[None, :], 4, input_ids, 3, 6, 0)
            cands = e["id"].ravel().tolist()
        best, bid = None, -1
        for cid in cands:
            if cid < 0 or cid in (i1, i2, i3):
                continue
            sim = oracle.cosine_similarity_bytea(raw, x[cid - 1])
            if best is None or sim > best or (sim == best and cid < bid):
                best, bid = sim, cid
        got = getattr(s, "analogy_3cosadd_in_" + which)(i1, i2, i3, input_ids)
        assert got == bid
    s.set_pvf(20)
