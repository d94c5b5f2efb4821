"""ctypes binding of the CPU oracle (oracle/libfreddy_oracle.so).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  PARITY UNPINNED (see
freddy_oracle.h): the oracle restates the reference, it was never run against it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("FREDDY_ORACLE_SO") or os.path.join(_HERE, "libfreddy_oracle.so")   # (tests/test_sanitizers.py: the ASan + UBSan build)

ENTRY = np.dtype([("id", np.int32), ("dist", np.float32)])


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("freddy_oracle.c", "freddy_oracle.h", "Makefile")]
    if os.environ.get("FREDDY_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class PQTable(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32), ("N", C.c_int64),
                ("codebook", C.c_void_p), ("ids", C.c_void_p), ("codes", C.c_void_p)]


class IVFTable(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32), ("C", C.c_int32),
                ("N", C.c_int64), ("coarse", C.c_void_p), ("codebook", C.c_void_p),
                ("list_off", C.c_void_p), ("ids", C.c_void_p), ("codes", C.c_void_p)]


class IVPQTable(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32), ("cpos", C.c_int32),
                ("ccodes", C.c_int32), ("N", C.c_int64), ("codebook", C.c_void_p),
                ("coarse", C.c_void_p), ("ids", C.c_void_p), ("coarse_id", C.c_void_p),
                ("codes", C.c_void_p), ("vectors", C.c_void_p), ("stats", C.c_void_p)]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin, array-in/array-out access to every oracle entry point."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.fo_sqdist.restype = C.c_float
        L.fo_sqdist.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.fo_adc.restype = C.c_float
        L.fo_adc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.fo_confidence_hyp.restype = C.c_float
        L.fo_confidence_hyp.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int]
        L.fo_emit_roundtrip.restype = C.c_float
        L.fo_emit_roundtrip.argtypes = [C.c_float]
        L.fo_topk_insert.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int32]
        L.fo_offer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int32]
        L.fo_offer.restype = C.c_int
        L.fo_ivfadc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.fo_ivfadc_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_int, C.c_int, C.c_void_p]
        L.fo_ivpq_search_in.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                        C.c_void_p, C.c_void_p]
        L.fo_cosine_similarity_bytea.restype = C.c_float
        L.fo_cosine_similarity_bytea.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.fo_exact_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p]
        L.fo_vec_minus.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.fo_vec_plus.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.fo_vec_normalize.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.fo_grouping_pq.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.fo_encode_pq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.fo_assign_coarse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.fo_multi_index_select.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.c_void_p, C.c_void_p]

    # ---- primitives ------------------------------------------------------------------
    def sqdist(self, a, b):
        a, b = _f32(a), _f32(b)
        return np.float32(self.lib.fo_sqdist(_p(a), _p(b), a.size))

    def lut(self, q, codebook):
        cb = _f32(codebook)
        m, K, s = cb.shape
        q = _f32(q)
        out = np.empty(m * K, np.float32)
        self.lib.fo_lut(_p(out), m, K, s, _p(q), _p(cb))
        return out

    def lut_entries(self, q, K, pos, code, vectors):
        pos, code, vectors, q = _i32(pos), _i32(code), _f32(vectors), _f32(q)
        n, s = vectors.shape
        out = np.full(n, np.nan, np.float32)
        self.lib.fo_lut_entries(_p(out), n, K, s, _p(q), _p(pos), _p(code), _p(vectors))
        return out

    def lut_double(self, q, codebook):
        cb = _f32(codebook)
        m, K, s = cb.shape
        q = _f32(q)
        out = np.empty((m // 2) * K * K, np.float32)
        self.lib.fo_lut_double(_p(out), m, K, s, _p(q), _p(cb))
        return out

    def adc(self, lut, codes, K):
        lut, codes = _f32(lut), _i16(codes)
        return np.float32(self.lib.fo_adc(_p(lut), _p(codes), codes.size, K))

    def topk_stream(self, dists, ids, k, sentinel):
        """offer() a whole candidate stream (a5 + guard) and return the final list."""
        tk = np.empty(k, ENTRY)
        self.lib.fo_topk_init(_p(tk), k, C.c_float(sentinel))
        maxd = np.array([sentinel], np.float32)
        for dd, ii in zip(np.asarray(dists, np.float32), np.asarray(ids, np.int32)):
            self.lib.fo_offer(_p(tk), k, _p(maxd), C.c_float(float(dd)), int(ii))
        return tk

    def cosine_similarity_bytea(self, a, b):
        a, b = _f32(a), _f32(b)
        return np.float32(self.lib.fo_cosine_similarity_bytea(_p(a), _p(b), a.size))

    def exact_knn(self, vectors, ids, q, k, input_ids=None):
        """(entries[:n]) with .dist = similarity, ORDER BY similarity DESC, id ASC."""
        v, ids, q = _f32(vectors), _i32(ids), _f32(q)
        out = np.empty(k, ENTRY)
        sub = None if input_ids is None else _i32(input_ids)
        n = self.lib.fo_exact_knn(_p(v), _p(ids), ids.size, v.shape[1], _p(q), k, _p(sub),
                                  0 if sub is None else sub.size, _p(out))
        return out[:n]

    def vec_minus(self, a, b):
        a, b = _f32(a), _f32(b); out = np.empty_like(a)
        self.lib.fo_vec_minus(_p(a), _p(b), a.size, _p(out)); return out

    def vec_plus(self, a, b):
        a, b = _f32(a), _f32(b); out = np.empty_like(a)
        self.lib.fo_vec_plus(_p(a), _p(b), a.size, _p(out)); return out

    def vec_normalize(self, v):
        v = _f32(v); out = np.empty_like(v)
        self.lib.fo_vec_normalize(_p(v), v.size, _p(out)); return out

    def grouping_pq(self, table, group_vecs, input_ids):
        """(ids, group index) of every row with id IN input_ids; group_vecs in ascending group-id order."""
        gv, ids = _f32(group_vecs), _i32(input_ids)
        oi = np.empty(max(ids.size, 1), np.int32); og = np.empty(max(ids.size, 1), np.int32)
        n = self.lib.fo_grouping_pq(C.byref(table), _p(gv), gv.shape[0], _p(ids), ids.size, _p(oi), _p(og))
        assert n >= 0
        return oi[:n], og[:n]

    def encode_pq(self, codebook, vecs):
        cb, v = _f32(codebook), _f32(vecs)
        m, K, s_ = cb.shape
        out = np.empty((v.shape[0], m), np.int16)
        self.lib.fo_encode_pq(_p(cb), m, K, s_, _p(v), v.shape[0], _p(out))
        return out

    def assign_coarse(self, coarse, vecs):
        c, v = _f32(coarse), _f32(vecs)
        out = np.empty(v.shape[0], np.int32)
        self.lib.fo_assign_coarse(_p(c), c.shape[0], c.shape[1], _p(v), v.shape[0], _p(out))
        return out

    def kmeans(self, vecs, k, iters, init_rows=None):
        v = _f32(vecs)
        cent = np.empty((k, v.shape[1]), np.float32)
        a = np.empty(v.shape[0], np.int32)
        ir = None if init_rows is None else _i32(init_rows)
        rc = self.lib.fo_kmeans(_p(v), C.c_int64(v.shape[0]), v.shape[1], k, iters, _p(ir), _p(cent), _p(a))
        assert rc == 0, rc
        return cent, a

    # ---- insert_batch (freddy.c:1403-1658) ----------------------------------------------------
    def text_roundtrip(self, a):
        a = _f32(a)
        self.lib.fo_text_roundtrip.restype = C.c_float
        return np.array([self.lib.fo_text_roundtrip(C.c_float(float(v))) for v in a.ravel()], np.float32).reshape(a.shape)

    def update_codebook(self, codebook, counts, vecs, order=None):
        """-> (new codebook, new counts, codes[n, m], count_incs[m*K]); raises on the reference's undefined case.
        order: the slots (pos * K + code) in the order the codebook table's tuples are scanned (None: position-major)."""
        cb, cnt, v = _f32(codebook).copy(), _i32(counts).copy(), _f32(vecs)
        m, K, s_ = cb.shape
        codes = np.empty((v.shape[0], m), np.int16)
        incs = np.empty(m * K, np.int32)
        od = None if order is None else _i32(order)
        assert od is None or sorted(od.tolist()) == list(range(m * K))
        rc = self.lib.fo_update_codebook_ordered(_p(cb), _p(cnt), m, K, s_, _p(v), v.shape[0], _p(codes), _p(incs), _p(od))
        if rc:
            raise ValueError(f"fo_update_codebook: {rc}")
        return cb, cnt, codes, incs

    def insert_coarse(self, coarse, vecs):
        c, v = _f32(coarse), _f32(vecs)
        cq = np.empty(v.shape[0], np.int32)
        res = np.empty_like(v)
        rc = self.lib.fo_insert_coarse(_p(c), c.shape[0], c.shape[1], _p(v), v.shape[0], _p(cq), _p(res))
        if rc:
            raise ValueError(f"fo_insert_coarse: {rc}")
        return cq, res

    def insert_coarse_multi(self, cq_multi, vecs):
        c, v = _f32(cq_multi), _f32(vecs)
        P, Kc, sub = c.shape
        out = np.empty(v.shape[0], np.int32)
        rc = self.lib.fo_insert_coarse_multi(_p(c), P, Kc, P * sub, _p(v), v.shape[0], _p(out))
        assert rc == 0
        return out

    def confidence_hyp(self, expect, size, p, stat_size):
        return np.float32(self.lib.fo_confidence_hyp(int(expect), int(size), C.c_float(float(p)), int(stat_size)))

    def emit_roundtrip(self, dist):
        return np.float32(self.lib.fo_emit_roundtrip(C.c_float(float(dist))))

    # ---- tables ----------------------------------------------------------------------
    @staticmethod
    def pq_table(codebook, ids, codes):
        cb, ids, codes = _f32(codebook), _i32(ids), _i16(codes)
        m, K, s = cb.shape
        t = PQTable(m * s, m, K, ids.size, _p(cb), _p(ids), _p(codes))
        t._keep = (cb, ids, codes)
        return t

    @staticmethod
    def ivf_table(coarse, codebook, list_off, ids, codes):
        cq, cb, lo, ids, codes = _f32(coarse), _f32(codebook), _i32(list_off), _i32(ids), _i16(codes)
        m, K, s = cb.shape
        t = IVFTable(m * s, m, K, cq.shape[0], ids.size, _p(cq), _p(cb), _p(lo), _p(ids), _p(codes))
        t._keep = (cq, cb, lo, ids, codes)
        return t

    @staticmethod
    def ivpq_table(codebook, coarse, ids, coarse_id, codes, vectors, stats):
        cb, cq, ids, cid, codes, st = _f32(codebook), _f32(coarse), _i32(ids), _i32(coarse_id), _i16(codes), _f32(stats)
        vec = None if vectors is None else _f32(vectors)
        m, K, s = cb.shape
        cpos, ccodes, _ = cq.shape
        t = IVPQTable(m * s, m, K, cpos, ccodes, ids.size, _p(cb), _p(cq), _p(ids), _p(cid), _p(codes),
                      None if vec is None else _p(vec), _p(st))
        t._keep = (cb, cq, ids, cid, codes, vec, st)
        return t

    # ---- drivers ---------------------------------------------------------------------
    def pq_search(self, t, q, k):
        q = _f32(q)
        out = np.empty(k, ENTRY)
        rc = self.lib.fo_pq_search(C.byref(t), _p(q), k, _p(out))
        assert rc == 0, rc
        return out

    def pq_search_in(self, t, q, k, input_ids):
        q, iid = _f32(q), _i32(input_ids)
        out = np.empty(k, ENTRY)
        rc = self.lib.fo_pq_search_in(C.byref(t), _p(q), k, _p(iid), iid.size, _p(out))
        assert rc == 0, rc
        return out

    def pq_search_in_batch(self, t, queries, k, input_ids, use_target_lists=True):
        qs, iid = _f32(queries), _i32(input_ids)
        out = np.empty((qs.shape[0], k), ENTRY)
        rc = self.lib.fo_pq_search_in_batch(C.byref(t), _p(qs), qs.shape[0], k, _p(iid), iid.size,
                                            int(use_target_lists), _p(out))
        assert rc == 0, rc
        return out

    def ivfadc_search(self, t, q, k, W, sentinel=1000.0, found_rule=0):
        q = _f32(q)
        out = np.empty(k, ENTRY)
        rc = self.lib.fo_ivfadc_search(C.byref(t), _p(q), k, W, C.c_float(sentinel), found_rule, _p(out))
        assert rc == 0, rc
        return out

    def ivfadc_search_many(self, t, queries, k, W, sentinel=1000.0, found_rule=0, n_threads=1):
        qs = _f32(queries)
        out = np.empty((qs.shape[0], k), ENTRY)
        rc = self.lib.fo_ivfadc_search_many(C.byref(t), _p(qs), qs.shape[0], k, W, C.c_float(sentinel),
                                            found_rule, n_threads, _p(out))
        assert rc == 0, rc
        return out

    def ivfadc_batch_search(self, t, queries, k):
        qs = _f32(queries)
        out = np.empty((qs.shape[0], k), ENTRY)
        rc = self.lib.fo_ivfadc_batch_search(C.byref(t), _p(qs), qs.shape[0], k, _p(out))
        assert rc == 0, rc
        return out

    def multi_index_select(self, t, queries, active, n_targets, min_target_count, confidence):
        qs, act = _f32(queries), _i32(active)
        cells = t.ccodes * t.ccodes
        out = np.full((act.size, cells), -1, np.int32)
        cnt = np.zeros(act.size, np.int32)
        last = self.lib.fo_multi_index_select(C.byref(t), _p(qs), _p(act), act.size, n_targets,
                                              min_target_count, C.c_float(confidence), _p(out), _p(cnt))
        assert last >= 0, last
        return [out[i, :cnt[i]].copy() for i in range(act.size)], bool(last)

    def ivpq_search_in(self, t, queries, k, target_ids, alpha, pvf, method, use_target_lists=True,
                       confidence=0.8, double_threshold=10000000):
        qs, tid = _f32(queries), _i32(target_ids)
        out = np.empty((qs.shape[0], k), ENTRY)
        iters = C.c_int(0)
        rc = self.lib.fo_ivpq_search_in(C.byref(t), _p(qs), qs.shape[0], k, _p(tid), tid.size, alpha, pvf,
                                        method, int(use_target_lists), C.c_float(confidence),
                                        double_threshold, _p(out), C.byref(iters))
        assert rc == 0, rc
        return out, iters.value
