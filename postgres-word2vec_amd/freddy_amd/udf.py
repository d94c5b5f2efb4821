"""Python face of the host-side UDF mirror (libfreddy_host.so, include/freddy_udf.h).

Function names and arguments are the reference's SQL-visible C functions
(freddy--0.0.1.sql:334-424): `pq_search(query, k)`, `ivfadc_search(query, k)`,
`pq_search_in(query, k, ids)`, `pq_search_in_batch(queries, query_ids, k, ids, use_targetlist)`,
`ivfadc_batch_search(ids, k)`, `ivpq_search_in(...)`, plus `knn_join` with the parameters
taken from the `set_*()` config functions.  Rows come back as numpy structured arrays shaped
like the SRF records.  No CPU fallback: a missing library raises.
"""
import ctypes as C
import os

import numpy as np

from . import gpu as _gpu

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("FREDDY_HOST_SO") or os.path.join(_PKG, "libfreddy_host.so")   # (tests/test_sanitizers.py: the ASan + UBSan build)

ROW2 = np.dtype([("id", np.int32), ("distance", np.float32)])
ROW3 = np.dtype([("query_id", np.int32), ("id", np.int32), ("distance", np.float32)])
GROUP_ROW = np.dtype([("id", np.int32), ("group_id", np.int32)])

_lib = None


class FreddyError(RuntimeError):
    """What the reference raises with elog(ERROR, ...)."""


def load():
    global _lib
    if _lib is None:
        _gpu.load()   # same HIP runtime ordering rule as the device library
        if not os.path.exists(LIB_PATH):
            raise FreddyError(f"{LIB_PATH} is missing (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = C.CDLL(LIB_PATH)
        _lib.freddy_udf_last_error.restype = C.c_char_p
        _lib.freddy_get_confidence_value.restype = C.c_float
        _lib.freddy_set_confidence_value.argtypes = [C.c_void_p, C.c_float]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _i16(a):
    return np.ascontiguousarray(a, np.int16)


def _entries(codebook):
    """dense [m][K][s] -> the (pos, code, vector) rows of a codebook table."""
    cb = _f32(codebook)
    m, K, s = cb.shape
    pos, code = np.divmod(np.arange(m * K, dtype=np.int32), K)
    return _i32(pos), _i32(code), cb.reshape(m * K, s), m * K, s


class Session:
    """One "database": tables + config functions + the UDFs."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = C.c_void_p()
        self._check(self.lib.freddy_session_open(device, C.byref(self.h)))

    def _check(self, rc):
        if rc != 0:
            raise FreddyError(self.lib.freddy_udf_last_error().decode())

    def close(self):
        if self.h:
            self.lib.freddy_session_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables ------------------------------------------------------------------------
    def load_vecs_norm(self, ids, vectors):
        ids, v = _i32(ids), _f32(vectors)
        self._check(self.lib.freddy_load_vecs_norm(self.h, _p(ids), _p(v), C.c_int64(ids.size), v.shape[1]))

    def load_pq(self, codebook, ids, codes):
        pos, code, vec, n, s = _entries(codebook)
        ids, codes = _i32(ids), _i16(codes)
        self._check(self.lib.freddy_load_pq(self.h, _p(pos), _p(code), _p(vec), n, s, _p(ids), _p(codes),
                                            C.c_int64(ids.size)))

    def load_ivfadc(self, coarse, codebook, ids, coarse_id, codes):
        pos, code, vec, n, s = _entries(codebook)
        cq = _f32(coarse)
        cids = _i32(np.arange(cq.shape[0]))
        ids, cid, codes = _i32(ids), _i32(coarse_id), _i16(codes)
        self._check(self.lib.freddy_load_ivfadc(self.h, _p(cids), _p(cq), cq.shape[0], _p(pos), _p(code), _p(vec), n, s,
                                                _p(ids), _p(cid), _p(codes), C.c_int64(ids.size)))

    def load_ivpq(self, codebook, coarse, ids, coarse_id, codes, stats):
        pos, code, vec, n, s = _entries(codebook)
        cpos, ccode, cvec, cn, _ = _entries(coarse)
        ids, cid, codes = _i32(ids), _i32(coarse_id), _i16(codes)
        st = _f32(stats)
        sid = _i32(np.arange(st.size))
        self._check(self.lib.freddy_load_ivpq(self.h, _p(pos), _p(code), _p(vec), n, s, _p(cpos), _p(ccode), _p(cvec), cn,
                                              _p(ids), _p(cid), _p(codes), C.c_int64(ids.size), _p(sid), _p(st), st.size))

    def import_index(self, path):
        """Load every complete table group of an index file (include/freddy_udf.h, FRDYIDX1)."""
        self._check(self.lib.freddy_import_index(self.h, str(path).encode()))

    # ---- config functions ----------------------------------------------------------------
    def set_w(self, v): self._check(self.lib.freddy_set_w(self.h, int(v)))
    def set_pvf(self, v): self._check(self.lib.freddy_set_pvf(self.h, int(v)))
    def set_alpha(self, v): self._check(self.lib.freddy_set_alpha(self.h, int(v)))
    def set_confidence_value(self, v): self._check(self.lib.freddy_set_confidence_value(self.h, C.c_float(v)))
    def set_long_codes_threshold(self, v): self._check(self.lib.freddy_set_long_codes_threshold(self.h, int(v)))
    def set_method_flag(self, v): self._check(self.lib.freddy_set_method_flag(self.h, int(v)))
    def set_use_targetlist(self, v): self._check(self.lib.freddy_set_use_targetlist(self.h, 1 if v else 0))
    def get_w(self): return self.lib.freddy_get_w(self.h)
    def get_pvf(self): return self.lib.freddy_get_pvf(self.h)
    def get_alpha(self): return self.lib.freddy_get_alpha(self.h)
    def get_confidence_value(self): return self.lib.freddy_get_confidence_value(self.h)

    # ---- UDFs ------------------------------------------------------------------------------
    def pq_search(self, query, k):
        q = _f32(query)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(self.lib.pq_search(self.h, _p(q), q.size, k, _p(out), C.byref(n)))
        return out[:n.value]

    def ivfadc_search(self, query, k):
        q = _f32(query)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(self.lib.ivfadc_search(self.h, _p(q), q.size, k, _p(out), C.byref(n)))
        return out[:n.value]

    def pq_search_in(self, query, k, input_ids):
        q, ids = _f32(query), _i32(input_ids)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(self.lib.pq_search_in(self.h, _p(q), q.size, k, _p(ids), ids.size, _p(out), C.byref(n)))
        return out[:n.value]

    def k_nearest_neighbour(self, query, k):
        """freddy--0.0.1.sql:426-439, rows carry (id, similarity)."""
        q = _f32(query)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(self.lib.k_nearest_neighbour(self.h, _p(q), q.size, k, _p(out), C.byref(n)))
        return out[:n.value]

    def knn_in_exact(self, query, k, input_ids):
        """freddy--0.0.1.sql:1041-1054, rows carry (id, similarity)."""
        q, ids = _f32(query), _i32(input_ids)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(self.lib.knn_in_exact(self.h, _p(q), q.size, k, _p(ids), ids.size, _p(out), C.byref(n)))
        return out[:n.value]

    def _knn2(self, fn, query, k):
        q = _f32(query)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(fn(self.h, _p(q), q.size, k, _p(out), C.byref(n)))
        return out[:n.value]

    def k_nearest_neighbour_pq(self, query, k): return self._knn2(self.lib.k_nearest_neighbour_pq, query, k)
    def k_nearest_neighbour_ivfadc(self, query, k): return self._knn2(self.lib.k_nearest_neighbour_ivfadc, query, k)
    def k_nearest_neighbour_pq_pv(self, query, k): return self._knn2(self.lib.k_nearest_neighbour_pq_pv, query, k)
    def k_nearest_neighbour_ivfadc_pv(self, query, k): return self._knn2(self.lib.k_nearest_neighbour_ivfadc_pv, query, k)

    def knn_in_pq(self, query, k, input_ids):
        q, ids = _f32(query), _i32(input_ids)
        out = np.empty(k, ROW2)
        n = C.c_int32(0)
        self._check(self.lib.knn_in_pq(self.h, _p(q), q.size, k, _p(ids), ids.size, _p(out), C.byref(n)))
        return out[:n.value]

    def k_nearest_neighbour_ivfadc_batch(self, query_ids, k):
        qid = _i32(query_ids)
        out = np.empty(max(qid.size, 1) * k, ROW3)
        n = C.c_int32(0)
        self._check(self.lib.k_nearest_neighbour_ivfadc_batch(self.h, _p(qid), qid.size, k, _p(out), C.byref(n)))
        return out[:n.value]

    def grouping_pq(self, input_ids, group_ids):
        """freddy.c:1176-1401: rows (id, group_id)."""
        ids, groups = _i32(input_ids), _i32(group_ids)
        out = np.empty(max(ids.size, 1), GROUP_ROW)
        n = C.c_int32(0)
        self._check(self.lib.grouping_pq(self.h, _p(ids), ids.size, _p(groups), groups.size, _p(out), C.byref(n)))
        return out[:n.value]

    def analogy_3cosadd_pq(self, id1, id2, id3):
        r = C.c_int32(-1)
        self._check(self.lib.analogy_3cosadd_pq(self.h, int(id1), int(id2), int(id3), C.byref(r)))
        return r.value

    def analogy_3cosadd_ivfadc(self, id1, id2, id3):
        r = C.c_int32(-1)
        self._check(self.lib.analogy_3cosadd_ivfadc(self.h, int(id1), int(id2), int(id3), C.byref(r)))
        return r.value

    def analogy_3cosadd_in_pq(self, id1, id2, id3, input_ids):
        r, ids = C.c_int32(-1), _i32(input_ids)
        self._check(self.lib.analogy_3cosadd_in_pq(self.h, int(id1), int(id2), int(id3), _p(ids), ids.size, C.byref(r)))
        return r.value

    def analogy_3cosadd_in_ivpq(self, id1, id2, id3, input_ids):
        r, ids = C.c_int32(-1), _i32(input_ids)
        self._check(self.lib.analogy_3cosadd_in_ivpq(self.h, int(id1), int(id2), int(id3), _p(ids), ids.size, C.byref(r)))
        return r.value

    def _cluster(self, fn, token_ids, k, draws):
        ids = _i32(token_ids)
        dr = None if draws is None else np.ascontiguousarray(draws, dtype=np.float64)
        out = np.zeros(ids.size, np.int32)
        self._check(fn(self.h, _p(ids), ids.size, int(k), _p(dr), 0 if dr is None else dr.size, _p(out)))
        return out

    def cluster_exact(self, token_ids, k, draws=None): return self._cluster(self.lib.cluster_exact, token_ids, k, draws)
    def cluster_pq(self, token_ids, k, draws=None): return self._cluster(self.lib.cluster_pq, token_ids, k, draws)
    def cluster_ivpq(self, token_ids, k, draws=None): return self._cluster(self.lib.cluster_ivpq, token_ids, k, draws)

    def set_codebook_counts(self, table, counts):
        """count column of pq_codebook (0) / residual_codebook (1) / codebook_ivpq (2): counts[m, K]."""
        c = _i32(counts)
        m, K = c.shape
        pos, code = np.divmod(np.arange(m * K, dtype=np.int32), K)
        self._check(self.lib.freddy_set_codebook_counts(self.h, int(table), _p(_i32(pos)), _p(_i32(code)), _p(c.reshape(-1)), m * K))

    def insert_batch(self, norm_vectors):
        """freddy.c:1403-1658 for the normalised vectors of NEW terms; returns the ids given in google_vecs_norm."""
        v = _f32(norm_vectors)
        out = np.empty(v.shape[0], np.int32)
        self._check(self.lib.insert_batch(self.h, _p(v), v.shape[0], v.shape[1], _p(out)))
        return out

    def pq_search_in_batch(self, queries, query_ids, k, input_ids, use_targetlist=True):
        qs, qid, ids = _f32(queries), _i32(query_ids), _i32(input_ids)
        out = np.empty(max(qs.shape[0], 1) * k, ROW3)
        n = C.c_int32(0)
        self._check(self.lib.pq_search_in_batch(self.h, _p(qs), qs.shape[0], qs.shape[1], _p(qid), qid.size, k, _p(ids),
                                                ids.size, int(use_targetlist), _p(out), C.byref(n)))
        return out[:n.value]

    def ivfadc_batch_search(self, query_ids, k):
        qid = _i32(query_ids)
        out = np.empty(max(qid.size, 1) * k, ROW3)
        n = C.c_int32(0)
        self._check(self.lib.ivfadc_batch_search(self.h, _p(qid), qid.size, k, _p(out), C.byref(n)))
        return out[:n.value]

    def ivpq_search_in(self, queries, query_ids, k, input_ids, alpha, pvf, method, use_targetlist, confidence,
                       double_threshold):
        qs, qid, ids = _f32(queries), _i32(query_ids), _i32(input_ids)
        out = np.empty(max(qs.shape[0], 1) * k, ROW3)
        n = C.c_int32(0)
        self._check(self.lib.ivpq_search_in(self.h, _p(qs), qs.shape[0], qs.shape[1], _p(qid), qid.size, k, _p(ids),
                                            ids.size, alpha, pvf, method, int(use_targetlist), C.c_float(confidence),
                                            double_threshold, _p(out), C.byref(n)))
        return out[:n.value]

    def knn_join(self, queries, query_ids, k, input_ids):
        qs, qid, ids = _f32(queries), _i32(query_ids), _i32(input_ids)
        out = np.empty(max(qs.shape[0], 1) * k, ROW3)
        n = C.c_int32(0)
        self._check(self.lib.knn_join(self.h, _p(qs), qs.shape[0], qs.shape[1], _p(qid), k, _p(ids), ids.size, _p(out),
                                      C.byref(n)))
        return out[:n.value]

    def emit_row3(self, row):
        vals = ((C.c_char * 16) * 3)()
        r = np.array([row], ROW3)
        self.lib.freddy_emit_row3(_p(r), vals)
        return tuple(v.value.decode() for v in vals)


class FileArray(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("dims", C.c_int64 * 2), ("data", C.c_void_p)]


_DT = {np.dtype(np.float32): 0, np.dtype(np.int32): 1, np.dtype(np.int16): 2}


def write_index_file(path, arrays):
    """arrays: {"<table>.<column>": ndarray (float32 / int32 / int16, 1-d or 2-d)} -> FRDYIDX1 file."""
    lib = load()
    keep, recs = [], (FileArray * len(arrays))()
    for i, (name, a) in enumerate(arrays.items()):
        a = np.ascontiguousarray(a)
        keep.append(a)
        dims = (C.c_int64 * 2)(a.shape[0], a.shape[1] if a.ndim == 2 else 0)
        recs[i] = FileArray(name.encode(), _DT[a.dtype], a.ndim, dims, a.ctypes.data_as(C.c_void_p))
    if lib.freddy_index_file_write(str(path).encode(), recs, len(arrays)) != 0:
        raise FreddyError(lib.freddy_udf_last_error().decode())


def table_arrays(pq=None, ivfadc=None, ivpq=None, vecs_norm=None):
    """The PG-table view of the builders' outputs (index_build.py), named as in the index file format."""
    out = {}
    if vecs_norm is not None:
        ids, v = vecs_norm
        out["google_vecs_norm.id"], out["google_vecs_norm.vector"] = _i32(ids), _f32(v)
    if pq is not None:
        pos, code, vec, _, _ = _entries(pq["codebook"])
        out.update({"pq_codebook.pos": pos, "pq_codebook.code": code, "pq_codebook.vector": vec,
                    "pq_quantization.id": _i32(pq["ids"]), "pq_quantization.vector": _i16(pq["codes"])})
    if ivfadc is not None:
        pos, code, vec, _, _ = _entries(ivfadc["codebook"])
        cq = _f32(ivfadc["coarse"])
        cell_of = np.repeat(np.arange(cq.shape[0]), np.diff(ivfadc["list_off"])).astype(np.int32)
        out.update({"coarse_quantization.id": _i32(np.arange(cq.shape[0])), "coarse_quantization.vector": cq,
                    "residual_codebook.pos": pos, "residual_codebook.code": code, "residual_codebook.vector": vec,
                    "fine_quantization.id": _i32(ivfadc["ids"]), "fine_quantization.coarse_id": cell_of,
                    "fine_quantization.vector": _i16(ivfadc["codes"])})
    if ivpq is not None:
        pos, code, vec, _, _ = _entries(ivpq["codebook"])
        cpos, ccode, cvec, _, _ = _entries(ivpq["coarse"])
        st = _f32(ivpq["stats"])
        out.update({"codebook_ivpq.pos": pos, "codebook_ivpq.code": code, "codebook_ivpq.vector": vec,
                    "coarse_quantization_ivpq.pos": cpos, "coarse_quantization_ivpq.code": ccode,
                    "coarse_quantization_ivpq.vector": cvec, "fine_quantization_ivpq.id": _i32(ivpq["ids"]),
                    "fine_quantization_ivpq.coarse_id": _i32(ivpq["coarse_id"]),
                    "fine_quantization_ivpq.vector": _i16(ivpq["codes"]), "stat.coarse_id": _i32(np.arange(st.size)),
                    "stat.coarse_freq": st})
    return out
