"""CPU tests of the oracle (oracle/freddy_oracle.c).

The reference ships no tests or golden vectors and cannot be built here, so the oracle is
"parity unpinned" (see its header).  What these tests can do is hold it to
  * hand-derived known answers for the rules read off the reference source,
  * independent pure-Python/numpy re-statements of the same rules (written from the rule,
    not from the C), and
  * the algebraic claims the HIP kernels rely on (selection-then-replay == full replay).
"""
import math

import numpy as np
import pytest

import util
from oracle.oracle import ENTRY

f32 = np.float32


# ---------------------------------------------------------------------------------------------
# independent models
# ---------------------------------------------------------------------------------------------
def py_sqdist(a, b):
    """index_utils.c:500-508 in explicit binary32 steps."""
    acc = f32(0.0)
    for x, y in zip(np.asarray(a, f32), np.asarray(b, f32)):
        t = f32(x - y)
        p = f32(t * t)
        acc = f32(acc + p)
    return acc


def py_insert(tk, d, i):
    """index_utils.c:19-33: walk from the tail to the first strictly smaller entry."""
    k = len(tk)
    slot = k - 1
    while slot >= 0 and not (tk[slot][1] < d):
        slot -= 1
    slot += 1
    if slot >= k:
        return
    for j in range(k - 2, slot - 1, -1):
        tk[j + 1] = tk[j]
    tk[slot] = (i, d)


def py_stream(dists, ids, k, sentinel, state=None):
    tk = [(-1, f32(sentinel))] * k if state is None else list(state)
    maxd = tk[k - 1][1]
    for d, i in zip(dists, ids):
        d = f32(d)
        if d < maxd:
            py_insert(tk, d, int(i))
            maxd = tk[k - 1][1]
    return tk


def as_list(entries):
    return [(int(e["id"]), f32(e["dist"])) for e in entries]


# ---------------------------------------------------------------------------------------------
# a1 squareDistance
# ---------------------------------------------------------------------------------------------
def test_sqdist_known_answers(oracle):
    assert oracle.sqdist([1, 2, 3], [0, 0, 0]) == f32(14.0)
    assert oracle.sqdist([], []) == f32(0.0)
    # (1+2^-12)^2 = 1 + 2^-11 + 2^-24 -> ties-to-even drops the last bit
    assert oracle.sqdist([1 + 2.0 ** -12], [0]) == f32(1 + 2.0 ** -11)


@pytest.mark.parametrize("n", [10, 25, 150, 300])
def test_sqdist_matches_stepwise_binary32(oracle, n):
    rng = np.random.default_rng(n)
    for scale in (1.0, 1e-3, 1e3, 1e-20):
        a = (rng.standard_normal(n) * scale).astype(f32)
        b = (rng.standard_normal(n) * scale).astype(f32)
        assert oracle.sqdist(a, b).view(np.uint32) == py_sqdist(a, b).view(np.uint32)


def test_sqdist_is_not_fused(oracle):
    """Find inputs where an FMA chain (one rounding per term) differs from the reference's
    mul-then-add (two roundings); the oracle must follow the two-rounding chain."""
    rng = np.random.default_rng(1)
    found = 0
    for _ in range(400):
        a = rng.standard_normal(25).astype(f32)
        b = rng.standard_normal(25).astype(f32)
        acc_f = f32(0.0)
        for x, y in zip(a, b):
            t = f32(x - y)
            acc_f = f32(np.float64(acc_f) + np.float64(t) * np.float64(t))   # exact product: fused
        two = py_sqdist(a, b)
        if acc_f.view(np.uint32) != two.view(np.uint32):
            found += 1
            assert oracle.sqdist(a, b).view(np.uint32) == two.view(np.uint32)
    assert found > 20


# ---------------------------------------------------------------------------------------------
# a2/a3/a4 LUT and ADC
# ---------------------------------------------------------------------------------------------
def test_lut_entry_order_independent_and_stepwise(oracle):
    rng = np.random.default_rng(2)
    m, K, s = 6, 16, 5
    cb = rng.standard_normal((m, K, s)).astype(f32)
    q = rng.standard_normal(m * s).astype(f32)
    lut = oracle.lut(q, cb)
    for p in range(m):
        for c in range(K):
            assert lut[p * K + c].view(np.uint32) == py_sqdist(q[p * s:(p + 1) * s], cb[p, c]).view(np.uint32)
    pos, code = np.divmod(np.arange(m * K), K)
    perm = rng.permutation(m * K)
    lut2 = oracle.lut_entries(q, K, pos[perm], code[perm], cb.reshape(m * K, s)[perm])
    assert np.array_equal(lut.view(np.uint32), lut2.view(np.uint32))


def test_lut_double_and_adc(oracle):
    rng = np.random.default_rng(3)
    m, K, s = 6, 8, 4
    cb = rng.standard_normal((m, K, s)).astype(f32)
    q = rng.standard_normal(m * s).astype(f32)
    lut = oracle.lut(q, cb)
    lut2 = oracle.lut_double(q, cb)
    for i in range(m // 2):
        for c0 in range(K):
            for c1 in range(K):
                exp = f32(lut[2 * i * K + c0] + lut[(2 * i + 1) * K + c1])
                assert lut2[K * K * i + c0 + K * c1].view(np.uint32) == exp.view(np.uint32)
    codes = rng.integers(0, K, size=m).astype(np.int16)
    acc = f32(0)
    for l in range(m):
        acc = f32(acc + lut[K * l + codes[l]])
    assert oracle.adc(lut, codes, K).view(np.uint32) == acc.view(np.uint32)
    # the pair table changes the rounding: ((a+b)+(c+d)) != (((a+b)+c)+d) in general
    pair = (codes[0::2] + K * codes[1::2]).astype(np.int16)
    acc2 = f32(0)
    for l in range(m // 2):
        acc2 = f32(acc2 + lut2[K * K * l + pair[l]])
    assert oracle.adc(lut2, pair, K * K).view(np.uint32) == acc2.view(np.uint32)


# ---------------------------------------------------------------------------------------------
# a5 / a-T  top-k insertion and its tie contract
# ---------------------------------------------------------------------------------------------
def test_topk_hand_derived_tie_cases(oracle):
    # later equal candidate goes in FRONT of earlier equals; boundary equal is rejected
    tk = oracle.topk_stream([3, 1, 2, 1, 1, 0.5], [10, 11, 12, 13, 14, 15], 3, 100.0)
    assert as_list(tk) == [(15, f32(0.5)), (14, f32(1)), (13, f32(1))]
    # full list of equals: a further equal is rejected (strict <), a smaller one evicts the EARLIEST equal
    tk = oracle.topk_stream([1, 1, 1, 1], [1, 2, 3, 4], 3, 100.0)
    assert as_list(tk) == [(3, f32(1)), (2, f32(1)), (1, f32(1))]
    tk = oracle.topk_stream([1, 1, 1, 0.5], [1, 2, 3, 4], 3, 100.0)
    assert as_list(tk) == [(4, f32(0.5)), (3, f32(1)), (2, f32(1))]
    # fewer candidates than k: sentinel rows stay
    tk = oracle.topk_stream([2.0], [7], 3, 1000.0)
    assert as_list(tk) == [(7, f32(2)), (-1, f32(1000)), (-1, f32(1000))]
    # candidates at / above the sentinel and NaN are never accepted
    tk = oracle.topk_stream([100.0, 250.0, float("nan"), 99.5], [1, 2, 3, 4], 2, 100.0)
    assert as_list(tk) == [(4, f32(99.5)), (-1, f32(100))]


@pytest.mark.parametrize("k", [1, 5, 16])
def test_topk_matches_python_model_with_heavy_ties(oracle, k):
    rng = np.random.default_rng(k)
    for _ in range(40):
        n = int(rng.integers(0, 200))
        d = rng.integers(0, 12, size=n).astype(f32) / f32(4)
        ids = rng.permutation(1000)[:n]
        assert as_list(oracle.topk_stream(d, ids, k, 100.0)) == py_stream(d, ids, k, 100.0)


def test_selection_then_replay_equals_full_replay():
    """The claim the HIP kernels rest on: keep only the 2k smallest (distance, scan position)
    keys of a stream, replay the reference insertion over them in scan order -> same list as
    replaying the whole stream.  Also with a carried list of fewer than k real entries (the
    state after a probing round that retrieved < k rows)."""
    rng = np.random.default_rng(0)
    for trial in range(1500):
        k = int(rng.integers(1, 9))
        n = int(rng.integers(0, 120))
        levels = int(rng.integers(1, 10))
        d = (rng.integers(0, levels, size=n) / 4).astype(f32)
        pos = np.arange(n)
        carried = None
        if trial % 3 == 0:
            c = int(rng.integers(0, k))
            cd = (rng.integers(0, levels, size=c) / 4).astype(f32)
            carried = py_stream(cd, -2 - np.arange(c), k, 100.0)
        full = py_stream(d, pos, k, 100.0, carried)
        order = np.lexsort((pos, d.view(np.uint32)))[:2 * k]
        keep = np.sort(order)
        part = py_stream(d[keep], pos[keep], k, 100.0, carried)
        assert full == part


def _replay_closed_form(carried, d, pos, k, sentinel, kmax=4096):
    """What bigk_replay_kernel (postgres-word2vec_amd/csrc/bigk.h) computes, restated: the list the guarded insertion leaves
    behind, without performing the insertions.  carried: the k (id, dist) entries of an earlier round or None; (d, pos): the
    new rows (the 2k smallest keys are enough), arrival = ascending position."""
    E = []   # (distance bits, arrival, id)
    if carried is not None:
        for s, (i, dd) in enumerate(carried):
            E.append((int(f32(dd).view(np.uint32)), k - 1 - s, i))
    for dd, p in zip(d, pos):
        E.append((int(f32(dd).view(np.uint32)), kmax + int(p), int(p)))
    E.sort(key=lambda e: (e[0], e[1]))
    sb = int(f32(sentinel).view(np.uint32))
    S = [e for e in E if e[0] < sb]       # (bit order = float order for non-negative floats; NaN bits are larger)
    runs_reversed = lambda rows: sorted(rows, key=lambda e: (e[0], -e[1]))
    if len(S) <= k:
        out = runs_reversed(S)
    else:
        ds = S[k - 1][0]
        lt = [e for e in S if e[0] < ds]
        ties = [e for e in S if e[0] == ds]
        A = sum(1 for j, t in enumerate(ties[:k]) if sum(1 for e in lt if e[1] < t[1]) + j < k)
        e_ = A + len(lt) - k
        out = runs_reversed(lt) + list(reversed(ties[e_:A]))
    res = [(e[2], np.uint32(e[0]).view(f32)) for e in out]
    return res + [(-1, f32(sentinel))] * (k - len(res))


def test_big_k_closed_form_equals_replay():
    """k > 512 (bigk.h): the list after the reference's guarded insertions, in closed form -- rows below the k-th smallest
    distance d* all stay; a row AT d* is accepted iff fewer than k rows with d <= d* arrived before it, and the earliest accepted
    ones are pushed out again by later rows below d*; equal distances end up in descending arrival order; a carried list counts
    as a prefix of the stream, last slot first.  Tie-heavy streams, sentinel and NaN rows, carried lists with fewer than k real
    entries, against the insertion loop."""
    rng = np.random.default_rng(1)
    for trial in range(4000):
        k = int(rng.integers(1, 12))
        n = int(rng.integers(0, 120))
        levels = int(rng.integers(1, 10))
        d = (rng.integers(0, levels, size=n) / 4).astype(f32)
        if trial % 7 == 0 and n:
            d[rng.integers(0, n)] = f32(100.0)
        if trial % 11 == 0 and n:
            d[rng.integers(0, n)] = f32(np.nan)
        pos = np.sort(rng.permutation(1000)[:n])
        carried = None
        if trial % 2 == 0:
            c = int(rng.integers(0, 3 * k))
            cd = (rng.integers(0, levels, size=c) / 4).astype(f32)
            carried = py_stream(cd, -2 - np.arange(c), k, 100.0)
        full = py_stream(d, pos, k, 100.0, carried)
        order = np.lexsort((pos, d.view(np.uint32)))[:2 * k]   # the 2k smallest (distance, position) keys, in any order
        got = _replay_closed_form(carried, d[order], pos[order], k, 100.0)
        assert [(i, float(x)) for i, x in full] == [(i, float(x)) for i, x in got], (trial, k)


def test_pv_buffer_closed_form():
    """updateTopKPVFast/reorderTopKPV (ivpq_search_in.c:40-57) with a stable sort keep exactly
    the `keep` smallest (distance, arrival) entries, ascending."""
    rng = np.random.default_rng(4)
    for _ in range(200):
        keep = int(rng.integers(1, 12))
        batch = 20 + keep
        n = int(rng.integers(0, 150))
        d = (rng.integers(0, 15, size=n) / 4).astype(f32)
        buf, fill, maxd = [(-1, f32(1000.0))] * batch, 0, f32(1000.0)

        def reorder(buf, fill):
            head = sorted(buf[:fill], key=lambda e: e[1])   # python sort is stable
            return head + buf[fill:], keep, head[keep - 1][1] if keep - 1 < len(head) else buf[keep - 1][1]

        for i in range(n):
            if d[i] < maxd:
                buf[fill] = (i, d[i]); fill += 1
                if fill == batch - 1:
                    buf, fill, maxd = reorder(buf, fill)
        buf, fill, maxd = reorder(buf, fill)
        got = [e for e in buf[:keep] if e[0] != -1]
        order = np.lexsort((np.arange(n), d.view(np.uint32)))[:keep]
        assert got == [(int(i), d[i]) for i in order]


# ---------------------------------------------------------------------------------------------
# a10 confidence + multi-index traversal
# ---------------------------------------------------------------------------------------------
def py_confidence(expect, size, p, stat_size):
    if expect > size:
        return f32(0)
    p = f32(p)
    mu = f32(size * p)   # int * float -> float
    sig = f32(math.sqrt(float(f32(size * p)) * (1.0 - float(p))) *
              (float(f32(f32(stat_size) - f32(size))) / (float(f32(stat_size)) - 1.0)))
    if sig == 0:
        z = (float(f32(expect)) - 0.5 - float(mu))
        e = math.copysign(1.0, z) if z != 0 else float("nan")
    else:
        e = math.erf((float(f32(expect)) - 0.5 - float(mu)) / (float(sig) * math.sqrt(2)))
    return f32(1.0 - 0.5 * (1.0 + e))


def test_confidence_hyp(oracle):
    assert oracle.confidence_hyp(10, 5, 0.5, 100) == 0          # expect > size
    assert oracle.confidence_hyp(5, 100, 0.0, 1000) == 0        # nothing probed yet
    for (e, n, p, S) in [(15, 1000, 0.02, 3000000), (500, 100000, 0.004, 3000000), (25, 4000, 0.01, 20000),
                         (5, 300, 0.05, 20000), (3, 100, 0.9, 20000), (500, 100000, 0.0049, 3000000)]:
        got, exp = oracle.confidence_hyp(e, n, p, S), py_confidence(e, n, p, S)
        assert abs(float(got) - float(exp)) <= 2e-7, (e, n, p, S, got, exp)
    # monotone in p
    vals = [float(oracle.confidence_hyp(50, 10000, p, 1000000)) for p in np.linspace(0.001, 0.02, 30)]
    assert all(b >= a for a, b in zip(vals, vals[1:]))


def test_multi_index_emits_cells_in_ascending_distance(oracle):
    t = util.ivpq_tables(N=6000, k_coarse=8)
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], None, t["stats"])
    _, qs = util.queries_from_corpus(6000, 12, seed=3)
    Kc, half = 8, 150
    # confidence > 1 can never be reached -> every cell is emitted, in ascending (D0+D1)
    cells, last = oracle.multi_index_select(ot, qs, np.arange(12), 3000, 10, 2.0)
    assert last
    for qi, cl in enumerate(cells):
        assert sorted(cl.tolist()) == list(range(Kc * Kc))
        d0 = [py_sqdist(qs[qi, :half], t["coarse"][0, c]) for c in range(Kc)]
        d1 = [py_sqdist(qs[qi, half:], t["coarse"][1, c]) for c in range(Kc)]
        dist = [f32(f32(f32(0) + d0[c % Kc]) + d1[c // Kc]) for c in cl]
        assert all(b >= a for a, b in zip(dist, dist[1:]))
    # reachable confidence: a prefix of that order, stopping exactly when the confidence is met
    part, last2 = oracle.multi_index_select(ot, qs, np.arange(12), 3000, 10, 0.8)
    assert not last2
    for qi, cl in enumerate(part):
        assert cl.tolist() == cells[qi][:len(cl)].tolist()
        prob = f32(0)
        for n_used, c in enumerate(cl):
            assert oracle.confidence_hyp(10, 3000, prob, int(t["stats"][-1])) < f32(0.8)
            prob = f32(prob + t["stats"][c])
        assert oracle.confidence_hyp(10, 3000, prob, int(t["stats"][-1])) >= f32(0.8)


# ---------------------------------------------------------------------------------------------
# drivers
# ---------------------------------------------------------------------------------------------
def test_pq_search_drivers_against_python_model(oracle):
    N = 3000
    t = util.pq_tables(N=N, K=64)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 3)
    K = 64
    for q in qs:
        lut = oracle.lut(q, t["codebook"])
        d = np.array([oracle.adc(lut, c, K) for c in t["codes"]], f32)
        assert as_list(oracle.pq_search(ot, q, 5)) == py_stream(d, t["ids"], 5, 100.0)
        sub = np.array([5, 17, 17, 900, 2999, 3000, 4000, -1], np.int32)
        rows = np.unique(sub[(sub >= 1) & (sub <= N)]) - 1
        assert as_list(oracle.pq_search_in(ot, q, 4, sub)) == py_stream(d[rows], t["ids"][rows], 4, 1000.0)
    sub = np.arange(1, N + 1, 7).astype(np.int32)
    a = oracle.pq_search_in_batch(ot, qs, 5, sub, use_target_lists=True)
    b = oracle.pq_search_in_batch(ot, qs, 5, sub, use_target_lists=False)
    assert np.array_equal(a, b)
    for i, q in enumerate(qs):
        assert np.array_equal(a[i], oracle.pq_search_in(ot, q, 5, sub))


def test_ivfadc_single_vs_python_model(oracle):
    N, C, K, m = 3000, 12, 64, 12
    t = util.ivf_tables(N=N, C=C, K=K)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(N, 6)
    for q in qs:
        for W in (1, 3):
            cd = np.array([oracle.sqdist(q, c) for c in t["coarse"]], f32)
            sel = [c for c, _ in py_stream(cd, np.arange(C), W, 100.0) if c >= 0]
            rows = np.concatenate([np.arange(t["list_off"][c], t["list_off"][c + 1]) for c in sel])
            rows = rows[np.argsort(t["ids"][rows], kind="stable")]          # canonical order = ascending id
            cell_of = np.repeat(np.arange(C), np.diff(t["list_off"]))
            luts = {c: oracle.lut((q - t["coarse"][c]).astype(f32), t["codebook"]) for c in sel}
            d = np.array([oracle.adc(luts[cell_of[r]], t["codes"][r], K) for r in rows], f32)
            assert as_list(oracle.ivfadc_search(ot, q, 5, W)) == py_stream(d, t["ids"][rows], 5, 1000.0)


def test_ivfadc_batch_udf_equals_w1_generalisation(oracle):
    for (N, C, k) in [(3000, 12, 5), (600, 150, 10)]:
        t = util.ivf_tables(N=N, C=C, K=64)
        ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
        _, qs = util.queries_from_corpus(N, 40)
        a = oracle.ivfadc_batch_search(ot, qs, k)
        b = oracle.ivfadc_search_many(ot, qs, k, 1, sentinel=100.0, found_rule=1)
        assert np.array_equal(a, b)
        c = oracle.ivfadc_search_many(ot, qs, k, 1, sentinel=100.0, found_rule=1, n_threads=4)
        assert np.array_equal(a, c)


def py_knn_join(oracle, t, ot, qs, k, targets, alpha, pvf, method, use_tl, conf):
    """Independent driver model of ivpq_search_in (cell selection taken from the oracle's a10)."""
    N, K = t["ids"].size, t["codebook"].shape[1]
    ids = t["ids"]
    tg = np.unique(targets[(targets >= ids[0]) & (targets <= ids[-1])])
    trows = np.searchsorted(ids, tg)
    trows = trows[ids[trows] == tg]
    Q = qs.shape[0]
    out = [[(-1, f32(1000.0))] * k for _ in range(Q)]
    tcount = np.zeros(Q, int)
    active = list(range(Q))
    alpha0, iters = alpha, 0
    pvf = max(pvf, 1)
    while active:
        iters += 1
        cells, last = oracle.multi_index_select(ot, qs, np.array(active, np.int32), targets.size, k * alpha, conf)
        nxt = []
        for x, q in enumerate(active):
            cset = set(cells[x].tolist())
            rows = [r for r in trows if t["coarse_id"][r] in cset]
            tcount[q] += len(rows)
            if use_tl and tcount[q] < k * alpha0 and not last:
                tcount[q] = 0
            else:
                lut = oracle.lut(qs[q], t["codebook"])
                adc = np.array([oracle.adc(lut, t["codes"][r], K) for r in rows], f32)
                ex = np.array([oracle.sqdist(qs[q], t["vectors"][r]) for r in rows], f32)
                rid = ids[rows] if rows else np.zeros(0, np.int32)
                if method == 0:
                    out[q] = py_stream(adc, rid, k, 1000.0)
                elif method == 1:
                    out[q] = py_stream(ex, rid, k, 1000.0)
                else:
                    order = np.lexsort((np.arange(len(rows)), adc.view(np.uint32)))[:k * pvf]
                    order = [o for o in order if adc[o] < f32(1000.0)]
                    out[q] = py_stream(ex[order], rid[order], k, 1000.0)
            if not last and out[q][k - 1][1] == f32(1000.0):
                out[q] = [(-1, f32(1000.0))] * k
                nxt.append(q)
        active = [] if last else nxt
        alpha += alpha
    return out, iters


@pytest.mark.parametrize("method", [0, 1, 2])
def test_knn_join_driver_against_python_model(oracle, method):
    N = 6000
    t = util.ivpq_tables(N=N, k_coarse=8)
    ot = oracle.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    _, qs = util.queries_from_corpus(N, 10, seed=9)
    rng = np.random.default_rng(8)
    targets = rng.choice(np.arange(1, N + 1), size=250, replace=False).astype(np.int32)
    multi = False
    for (k, alpha, pvf, conf, use_tl) in [(5, 3, 4, 0.8, True), (5, 1, 3, 0.3, True), (5, 1, 3, 0.3, False),
                                          (3, 1, 2, 0.1, True), (4, 40, 2, 0.8, True)]:
        exp, eit = py_knn_join(oracle, t, ot, qs, k, targets, alpha, pvf, method, use_tl, conf)
        got, git = oracle.ivpq_search_in(ot, qs, k, targets, alpha, pvf, method, use_target_lists=use_tl, confidence=conf)
        assert git == eit
        multi |= git > 1
        for q in range(qs.shape[0]):
            assert as_list(got[q]) == exp[q], (method, k, alpha, pvf, conf, use_tl, q)
    assert multi


def test_emit_text_roundtrip(oracle):
    assert oracle.emit_roundtrip(0.1234567) == f32(0.123457)
    assert oracle.emit_roundtrip(100.0) == f32(100.0)
    assert oracle.emit_roundtrip(1e-8) == f32(0.0)


def test_exact_knn_oracle(oracle):
    """cosine_similarity_bytea (core_functions.c:67-81) and ORDER BY ... DESC FETCH FIRST k."""
    x = util.corpus(3000).numpy()
    ids = np.arange(1, 3001, dtype=np.int32)
    q = x[10]
    sims = np.empty(3000, f32)
    for r in range(3000):
        acc = f32(0)
        for a_, b_ in zip(q, x[r]):
            acc = f32(acc + f32(a_ * b_))
        sims[r] = acc
        if r < 50:
            assert oracle.cosine_similarity_bytea(q, x[r]).view(np.uint32) == acc.view(np.uint32)
    for k in (1, 5, 64):
        order = np.lexsort((ids, -sims.astype(np.float64)))[:k]
        got = oracle.exact_knn(x, ids, q, k)
        assert got["id"].tolist() == ids[order].tolist()
        assert np.array_equal(got["dist"].view(np.uint32), sims[order].view(np.uint32))
    sub = np.array([5, 11, 12, 9000, 11, -3], np.int32)
    got = oracle.exact_knn(x, ids, q, 5, sub)
    assert got["id"].tolist() == [11, 12, 5] or set(got["id"].tolist()) == {11, 12, 5}
    assert len(got) == 3
    # duplicated rows tie exactly: ascending id decides
    x2 = np.concatenate([x[:20], x[5:6], x[5:6]])
    ids2 = np.arange(1, 23, dtype=np.int32)
    got = oracle.exact_knn(x2, ids2, x[5], 3)
    assert got["id"].tolist() == [6, 21, 22]


def test_vec_ops_and_grouping_pq_oracle(oracle):
    """core_functions.c vec_minus/plus/normalize_bytea and grouping_pq (freddy.c:1176-1401) against
    step-wise binary32 Python models."""
    x = util.corpus(3000).numpy()
    a, b, c = x[1], x[2], x[3]
    raw = np.array([f32(f32(cc - aa) + bb) for aa, bb, cc in zip(a, b, c)], f32)
    assert np.array_equal(oracle.vec_plus(oracle.vec_minus(c, a), b).view(np.uint32), raw.view(np.uint32))
    sq = f32(0)
    for v in raw:
        sq = f32(sq + f32(v * v))
    length = f32(np.sqrt(np.float64(sq)))
    unit = np.array([f32(v / length) for v in raw], f32)
    assert np.array_equal(oracle.vec_normalize(raw).view(np.uint32), unit.view(np.uint32))

    t = util.pq_tables(N=3000, K=64)
    ot = oracle.pq_table(t["codebook"], t["ids"], t["codes"])
    gvec = x[[10, 500, 2222]]
    asked = [7, 7, 2999, 1, 5000, -2, 1500]
    ids, grp = oracle.grouping_pq(ot, gvec, asked)
    assert ids.tolist() == [1, 7, 1500, 2999]                      # table order, de-duplicated, unknown ids dropped
    luts = [oracle.lut(g, t["codebook"]) for g in gvec]
    K = t["codebook"].shape[1]
    for i, g in zip(ids, grp):
        best, md = -1, f32(100)
        for gi, L in enumerate(luts):
            acc = f32(0)
            for j, code in enumerate(t["codes"][i - 1]):
                acc = f32(acc + L[j * K + code])
            if acc < md:
                md, best = acc, gi
        assert best == g
    # a duplicated group vector: the FIRST of the equally near groups wins (strict <)
    ids2, grp2 = oracle.grouping_pq(ot, np.stack([gvec[1], gvec[1], gvec[0]]), [501])
    assert grp2.tolist() in ([0], [2]) and 1 not in grp2.tolist()


def test_encode_oracle(oracle):
    """fo_encode_pq / fo_assign_coarse: exact 1-NN by squareDistance, lowest index on ties."""
    rng = np.random.default_rng(5)
    m, K, s = 4, 16, 5
    cb = rng.standard_normal((m, K, s)).astype(f32)
    cb[1, 9] = cb[1, 3]                                  # duplicated codeword: the lower code wins
    v = rng.standard_normal((50, m * s)).astype(f32)
    v[7, s:2 * s] = cb[1, 3]
    codes = oracle.encode_pq(cb, v)
    for i in range(50):
        for p in range(m):
            ds = [py_sqdist(v[i, p * s:(p + 1) * s], cb[p, j]) for j in range(K)]
            assert codes[i, p] == int(np.argmin(np.array(ds, f32)))   # argmin returns the first minimum
    assert codes[7, 1] == 3
    coarse = rng.standard_normal((9, m * s)).astype(f32)
    coarse[6] = coarse[2]
    cell = oracle.assign_coarse(coarse, v)
    for i in range(50):
        ds = np.array([py_sqdist(v[i], coarse[c]) for c in range(9)], f32)
        assert cell[i] == int(np.argmin(ds)) and cell[i] != 6


def golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "freddy_small.npz"))


def test_oracle_reproduces_committed_golden_vectors(oracle):
    """tests/golden/freddy_small.npz (inputs + the oracle's outputs when it was written; see
    make_golden.py for what it does and does not pin): the oracle built here gives the same bits."""
    g = golden()
    x, ids, qs = g["x"], g["ids"], g["queries"]
    same = lambda a, b: np.array_equal(a["id"], b["id"]) and np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32))
    assert np.array_equal(oracle.encode_pq(g["pq_codebook"], x), g["pq_codes"])
    pq = oracle.pq_table(g["pq_codebook"], ids, g["pq_codes"])
    assert same(np.stack([oracle.pq_search(pq, q, 5) for q in qs]), g["pq_search"])
    assert same(np.stack([oracle.pq_search_in(pq, q, 4, g["subset"]) for q in qs]), g["pq_search_in"])
    ivf = oracle.ivf_table(g["coarse"], g["codebook"], g["list_off"], g["ivf_ids"], g["ivf_codes"])
    assert same(oracle.ivfadc_search_many(ivf, qs, 5, 3, sentinel=1000.0, found_rule=0), g["ivfadc_w3_k5"])
    assert same(oracle.ivfadc_search_many(ivf, qs, 7, 1, sentinel=100.0, found_rule=1), g["ivfadc_w1_k7_accepted"])
    assert same(oracle.ivfadc_search_many(ivf, qs, 20, 8, sentinel=1000.0, found_rule=0), g["ivfadc_w8_k20"])
    ivpq = oracle.ivpq_table(g["ivpq_codebook"], g["ivpq_coarse"], ids, g["ivpq_coarse_id"], g["ivpq_codes"], x, g["ivpq_stats"])
    for method in (0, 1, 2):
        exp, it = oracle.ivpq_search_in(ivpq, qs, 5, g["targets"], 3, 4, method)
        assert same(exp, g[f"knn_join_m{method}"]) and it == int(g[f"knn_join_m{method}_iterations"])
    assert same(np.stack([oracle.exact_knn(x, ids, q, 6) for q in qs]), g["exact_knn"])
    gi, gg = oracle.grouping_pq(pq, x[[9, 199, 349]], g["grouping_input"])
    assert np.array_equal(gi, g["grouping_ids"]) and np.array_equal(gg, g["grouping_group"])
