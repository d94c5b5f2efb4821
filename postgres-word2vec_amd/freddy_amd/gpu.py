"""ctypes binding of the C ABI in include/freddy_gpu.h (libfreddy_gpu.so).

The library is the product; this module only marshals numpy arrays / device pointers
into it.  There is NO CPU fallback: if the shared object is missing or a call fails the
error is raised.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("FREDDY_GPU_SO") or os.path.join(_PKG, "libfreddy_gpu.so")   # (FREDDY_GPU_SO: the tools' -DFREDDY_LAB build)

FOUND_ROWS = 0
FOUND_ACCEPTED = 1
FOUND_BATCH_UDF = 2   # ivfadc_batch_search itself (W == 1): accepted-rows rule + its argmin cell limit of 1000
METHOD_PQ, METHOD_EXACT, METHOD_PQ_PV = 0, 1, 2

EXPORTS = [
    "freddy_gpu_pin_pq", "freddy_gpu_pin_ivf", "freddy_gpu_pin_ivpq", "freddy_gpu_unpin",
    "freddy_gpu_pq_search", "freddy_gpu_ivfadc_search", "freddy_gpu_knn_join",
    "freddy_gpu_ivfadc_search_dev", "freddy_gpu_pq_search_dev", "freddy_gpu_last_error",
    "freddy_gpu_profile_enable", "freddy_gpu_profile_read", "freddy_gpu_index_bytes",
    "freddy_gpu_last_scanned_rows", "freddy_gpu_filter_bound_violations", "freddy_gpu_filter_bound_checked", "freddy_gpu_pin_vectors", "freddy_gpu_exact_search", "freddy_gpu_grouping_pq",
    "freddy_gpu_encode", "freddy_gpu_set_option", "freddy_gpu_last_track", "freddy_gpu_last_probed_cells", "freddy_gpu_coarse_bound_checked",
    "freddy_gpu_insert_quantize", "freddy_gpu_append_rows", "freddy_gpu_update_codebook", "freddy_gpu_kmeans",
    "freddy_gpu_host_alloc", "freddy_gpu_host_free", "freddy_gpu_pin_ivf_multi", "freddy_gpu_replica_count",
    "freddy_gpu_last_track_sized", "freddy_gpu_abi_version",
]
ABI_VERSION = 4   # include/freddy_gpu.h FREDDY_GPU_ABI_VERSION this binding was written against


class FreddyGpuError(RuntimeError):
    pass


class PQDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32), ("N", C.c_int64),
                ("codebook", C.c_void_p), ("ids", C.c_void_p), ("codes", C.c_void_p)]


class IVFDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32), ("C", C.c_int32),
                ("N", C.c_int64), ("coarse", C.c_void_p), ("codebook", C.c_void_p),
                ("list_off", C.c_void_p), ("ids", C.c_void_p), ("codes", C.c_void_p)]


class VecDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("N", C.c_int64), ("ids", C.c_void_p), ("vectors", C.c_void_p)]


class IVPQDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32),
                ("coarse_positions", C.c_int32), ("coarse_codes", C.c_int32), ("N", C.c_int64),
                ("codebook", C.c_void_p), ("coarse", C.c_void_p), ("ids", C.c_void_p),
                ("coarse_id", C.c_void_p), ("codes", C.c_void_p), ("vectors", C.c_void_p),
                ("stats", C.c_void_p)]


_lib = None


def load():
    """dlopen the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
        # runtimes in one process fight over the device, so when torch is around let its copy
        # be the one the dynamic loader binds first.  (A PostgreSQL backend has no torch and
        # simply uses /opt/rocm's runtime.)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    if not os.path.exists(LIB_PATH):
        raise FreddyGpuError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    if lib.freddy_gpu_abi_version() != ABI_VERSION:
        raise FreddyGpuError(f"{LIB_PATH} has ABI version {lib.freddy_gpu_abi_version()}, this binding expects {ABI_VERSION}: rebuild")
    lib.freddy_gpu_last_track_sized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.freddy_gpu_last_error.restype = C.c_char_p
    lib.freddy_gpu_index_bytes.restype = C.c_int64
    lib.freddy_gpu_index_bytes.argtypes = [C.c_void_p]
    lib.freddy_gpu_last_scanned_rows.restype = C.c_int64
    lib.freddy_gpu_last_scanned_rows.argtypes = [C.c_void_p]
    lib.freddy_gpu_filter_bound_violations.restype = C.c_int64
    lib.freddy_gpu_filter_bound_violations.argtypes = [C.c_void_p]
    lib.freddy_gpu_filter_bound_checked.restype = C.c_int64
    lib.freddy_gpu_filter_bound_checked.argtypes = [C.c_void_p]
    lib.freddy_gpu_coarse_bound_checked.restype = C.c_int64
    lib.freddy_gpu_coarse_bound_checked.argtypes = [C.c_void_p]
    lib.freddy_gpu_pin_pq.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.freddy_gpu_pin_ivf.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.freddy_gpu_pin_ivpq.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.freddy_gpu_pin_ivf_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.freddy_gpu_replica_count.argtypes = [C.c_void_p]
    lib.freddy_gpu_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    lib.freddy_gpu_host_free.argtypes = [C.c_void_p]
    lib.freddy_gpu_unpin.argtypes = [C.c_void_p]
    lib.freddy_gpu_pin_vectors.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.freddy_gpu_exact_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p]
    lib.freddy_gpu_pq_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_ivfadc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_knn_join.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_ivfadc_search_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
    lib.freddy_gpu_pq_search_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_last_probed_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_insert_quantize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    lib.freddy_gpu_append_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_update_codebook.argtypes = [C.c_void_p, C.c_void_p]
    lib.freddy_gpu_kmeans.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.freddy_gpu_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.freddy_gpu_last_track.argtypes = [C.c_void_p, C.c_void_p]
    lib.freddy_gpu_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.freddy_gpu_profile_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise FreddyGpuError(f"freddy_gpu error {rc}: {load().freddy_gpu_last_error().decode()}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _Index:
    kind = "?"

    def __init__(self):
        self.h = C.c_void_p()
        self.lib = load()

    def close(self):
        if self.h:
            self.lib.freddy_gpu_unpin(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nbytes(self):
        return int(self.lib.freddy_gpu_index_bytes(self.h))

    def append_rows(self, ids, coarse_id=None, codes=None, vectors=None):
        """INSERT of new rows into the pinned tables (freddy_gpu_append_rows): ids ascending beyond the pinned ones."""
        ids = _i32(ids)
        cid = None if coarse_id is None else _i32(coarse_id)
        cd = None if codes is None else _i16(codes)
        v = None if vectors is None else _f32(vectors)
        _check(self.lib.freddy_gpu_append_rows(self.h, ids.size, _p(ids), _p(cid), _p(cd), _p(v)))

    def update_codebook(self, codebook):
        cb = _f32(codebook)
        _check(self.lib.freddy_gpu_update_codebook(self.h, _p(cb)))

    def bound_violations(self):
        """Rows of the filter + refine scan's exact stage whose distance left the proven bracket (must be 0)."""
        return int(self.lib.freddy_gpu_filter_bound_violations(self.h))

    def set_option(self, name, value):
        """Tuning / debug switch of this pinned index (include/freddy_gpu.h: freddy_gpu_set_option)."""
        _check(self.lib.freddy_gpu_set_option(self.h, name.encode(), int(value)))

    def profile_enable(self, on=True):
        _check(self.lib.freddy_gpu_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        cap = 32
        names = ((C.c_char * 64) * cap)()
        launches = (C.c_int64 * cap)()
        ms = (C.c_double * cap)()
        n = self.lib.freddy_gpu_profile_read(self.h, cap, names, launches, ms)
        if n < 0:
            _check(n)
        return {names[i].value.decode(): (int(launches[i]), float(ms[i])) for i in range(min(n, cap))}


class PQIndex(_Index):
    """pq_codebook + pq_quantization pinned in HBM."""
    kind = "pq"

    def __init__(self, codebook, ids, codes, device=0):
        super().__init__()
        cb, ids, codes = _f32(codebook), _i32(ids), _i16(codes)
        m, K, s = cb.shape
        assert codes.shape == (ids.size, m)
        self.d, self.m, self.K, self.N = m * s, m, K, ids.size
        desc = PQDesc(self.d, m, K, ids.size, _p(cb), _p(ids), _p(codes))
        _check(self.lib.freddy_gpu_pin_pq(C.byref(desc), device, C.byref(self.h)))

    def search(self, queries, k, sentinel=100.0, subset_ids=None):
        """(ids[Q,k], dist[Q,k]); subset_ids=None -> pq_search, else pq_search_in[_batch]."""
        qs = _f32(queries).reshape(-1, self.d)
        Q = qs.shape[0]
        out_i = np.empty((Q, k), np.int32)
        out_d = np.empty((Q, k), np.float32)
        sub = None if subset_ids is None else _i32(subset_ids)
        _check(self.lib.freddy_gpu_pq_search(self.h, _p(qs), Q, k, C.c_float(sentinel), _p(sub),
                                             0 if sub is None else sub.size, _p(out_i), _p(out_d)))
        return out_i, out_d

    def bind_search(self, queries, k, sentinel=100.0):
        """pq_search through the host-buffer call with every argument converted ONCE (the caller's loop then costs what a C
        caller's costs): returns (call, out_ids, out_dist); call() fills the two arrays."""
        qs = _f32(queries).reshape(-1, self.d)
        Q = qs.shape[0]
        out_i = np.empty((Q, k), np.int32)
        out_d = np.empty((Q, k), np.float32)
        fn = self.lib.freddy_gpu_pq_search
        args = (self.h, _p(qs), C.c_int32(Q), C.c_int32(k), C.c_float(sentinel), None, C.c_int64(0), _p(out_i), _p(out_d))

        def call(_keep=(qs,)):
            rc = fn(*args)
            if rc != 0:
                _check(rc)
        return call, out_i, out_d

    def grouping(self, group_vectors, subset_ids=None):
        """grouping_pq: (ids, group index) of every (requested) row; groups tried in the given order."""
        gv = _f32(group_vectors).reshape(-1, self.d)
        sub = None if subset_ids is None else _i32(subset_ids)
        cap = self.N if sub is None else max(sub.size, 1)
        oi = np.empty(cap, np.int32); og = np.empty(cap, np.int32)
        n = C.c_int64(0)
        _check(self.lib.freddy_gpu_grouping_pq(self.h, _p(gv), gv.shape[0], _p(sub), 0 if sub is None else sub.size,
                                               _p(oi), _p(og), C.byref(n)))
        return oi[:n.value], og[:n.value]

    def search_dev(self, d_queries_ptr, Q, k, sentinel, d_out_ids_ptr, d_out_dist_ptr, stream=None):
        _check(self.lib.freddy_gpu_pq_search_dev(self.h, C.c_void_p(d_queries_ptr), Q, k, C.c_float(sentinel),
                                                 C.c_void_p(d_out_ids_ptr), C.c_void_p(d_out_dist_ptr),
                                                 C.c_void_p(stream or 0)))


class VectorIndex(_Index):
    """google_vecs_norm pinned as raw vectors: exact brute-force kNN."""
    kind = "vec"

    def __init__(self, ids, vectors, device=0):
        super().__init__()
        ids, v = _i32(ids), _f32(vectors)
        self.d, self.N = v.shape[1], ids.size
        desc = VecDesc(self.d, ids.size, _p(ids), _p(v))
        _check(self.lib.freddy_gpu_pin_vectors(C.byref(desc), device, C.byref(self.h)))

    def search(self, queries, k, subset_ids=None):
        """(ids[Q,k], similarity[Q,k]) ORDER BY cosine_similarity_bytea DESC, id ASC."""
        qs = _f32(queries).reshape(-1, self.d)
        Q = qs.shape[0]
        out_i = np.empty((Q, k), np.int32)
        out_s = np.empty((Q, k), np.float32)
        sub = None if subset_ids is None else _i32(subset_ids)
        _check(self.lib.freddy_gpu_exact_search(self.h, _p(qs), Q, k, _p(sub), 0 if sub is None else sub.size,
                                                _p(out_i), _p(out_s)))
        return out_i, out_s


class IVFIndex(_Index):
    """coarse_quantization + residual_codebook + fine_quantization pinned in HBM."""
    kind = "ivf"

    def __init__(self, coarse, codebook, list_off, ids, codes, device=0, devices=None):
        """devices: a list of device ordinals -> the tables replicated on each of them behind one handle
        (freddy_gpu_pin_ivf_multi); host batches are then split contiguously over the devices."""
        super().__init__()
        cq, cb, lo, ids, codes = _f32(coarse), _f32(codebook), _i32(list_off), _i32(ids), _i16(codes)
        m, K, s = cb.shape
        assert cq.shape[1] == m * s and lo.size == cq.shape[0] + 1
        self.d, self.m, self.K, self.C, self.N = m * s, m, K, cq.shape[0], ids.size
        desc = IVFDesc(self.d, m, K, self.C, ids.size, _p(cq), _p(cb), _p(lo), _p(ids), _p(codes))
        if devices is None:
            _check(self.lib.freddy_gpu_pin_ivf(C.byref(desc), device, C.byref(self.h)))
        else:
            devs = (C.c_int * len(devices))(*devices)
            _check(self.lib.freddy_gpu_pin_ivf_multi(C.byref(desc), devs, len(devices), C.byref(self.h)))

    @property
    def replicas(self):
        return int(self.lib.freddy_gpu_replica_count(self.h))

    def search(self, queries, k, W, sentinel=1000.0, found_rule=FOUND_ROWS):
        qs = _f32(queries).reshape(-1, self.d)
        Q = qs.shape[0]
        out_i = np.empty((Q, k), np.int32)
        out_d = np.empty((Q, k), np.float32)
        _check(self.lib.freddy_gpu_ivfadc_search(self.h, _p(qs), Q, k, W, C.c_float(sentinel), found_rule,
                                                 _p(out_i), _p(out_d)))
        return out_i, out_d

    def search_dev(self, d_queries_ptr, Q, k, W, sentinel, found_rule, d_out_ids_ptr, d_out_dist_ptr,
                   d_status_ptr=0, stream=None):
        _check(self.lib.freddy_gpu_ivfadc_search_dev(self.h, C.c_void_p(d_queries_ptr), Q, k, W,
                                                     C.c_float(sentinel), found_rule,
                                                     C.c_void_p(d_out_ids_ptr), C.c_void_p(d_out_dist_ptr),
                                                     C.c_void_p(d_status_ptr or 0), C.c_void_p(stream or 0)))

    def bind_search_dev(self, d_queries_ptr, Q, k, W, sentinel, found_rule, d_out_ids_ptr, d_out_dist_ptr,
                        d_status_ptr=0, stream=None):
        """search_dev with every argument converted ONCE: returns a zero-argument callable that makes the C call and
        raises on a non-zero status.  (A loop that enqueues the same batch again and again -- bench.py -- spent a
        quarter of its host time per step converting the eleven arguments.)"""
        fn = self.lib.freddy_gpu_ivfadc_search_dev
        args = (self.h, C.c_void_p(d_queries_ptr), C.c_int32(Q), C.c_int32(k), C.c_int32(W), C.c_float(sentinel),
                C.c_int32(found_rule), C.c_void_p(d_out_ids_ptr), C.c_void_p(d_out_dist_ptr),
                C.c_void_p(d_status_ptr or 0), C.c_void_p(stream or 0))

        def call():
            rc = fn(*args)
            if rc != 0:
                _check(rc)
        return call

    def last_scanned_rows(self):
        return int(self.lib.freddy_gpu_last_scanned_rows(self.h))

    def last_probed_cells(self):
        """(distinct cells, rows of their lists) of the last probing round (cell-grouped scans)."""
        n, r = C.c_int64(0), C.c_int64(0)
        _check(self.lib.freddy_gpu_last_probed_cells(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def coarse_bound_checked(self):
        return int(self.lib.freddy_gpu_coarse_bound_checked(self.h))

    def bound_checked(self):
        return int(self.lib.freddy_gpu_filter_bound_checked(self.h))


class IVPQIndex(_Index):
    """codebook_ivpq + coarse multi-index + fine_quantization_ivpq (+vectors, stats) in HBM."""
    kind = "ivpq"

    def __init__(self, codebook, coarse, ids, coarse_id, codes, vectors, stats, device=0):
        super().__init__()
        cb, cq, ids, cid, codes, st = (_f32(codebook), _f32(coarse), _i32(ids), _i32(coarse_id),
                                       _i16(codes), _f32(stats))
        vec = None if vectors is None else _f32(vectors)
        m, K, s = cb.shape
        cpos, ccodes, _ = cq.shape
        self.d, self.m, self.K, self.N = m * s, m, K, ids.size
        desc = IVPQDesc(self.d, m, K, cpos, ccodes, ids.size, _p(cb), _p(cq), _p(ids), _p(cid), _p(codes),
                        _p(vec), _p(st))
        _check(self.lib.freddy_gpu_pin_ivpq(C.byref(desc), device, C.byref(self.h)))

    def knn_join(self, queries, k, target_ids, alpha, pvf, method, use_target_lists=True, confidence=0.8,
                 double_threshold=10000000):
        qs = _f32(queries).reshape(-1, self.d)
        tid = _i32(target_ids)
        Q = qs.shape[0]
        out_i = np.empty((Q, k), np.int32)
        out_d = np.empty((Q, k), np.float32)
        iters = C.c_int32(0)
        _check(self.lib.freddy_gpu_knn_join(self.h, _p(qs), Q, k, _p(tid), tid.size, alpha, pvf, method,
                                            1 if use_target_lists else 0, C.c_float(confidence),
                                            double_threshold, _p(out_i), _p(out_d), C.byref(iters)))
        return out_i, out_d, iters.value

    def last_track(self):
        """{stage name: seconds} of the most recent knn_join call (the reference's TRACK lines)."""
        t = Track()
        n = self.lib.freddy_gpu_last_track_sized(self.h, C.byref(t), C.sizeof(t))
        if n < 0:
            _check(n)
        return {n: getattr(t, n) for n, _ in Track._fields_ if n != "reserved"}


class Track(C.Structure):
    """freddy_track: stage timers under the reference's TRACK names (ivpq_search_in.c:234-697)."""
    _fields_ = [("precomputation_time", C.c_double), ("determine_coarse_quantization_time", C.c_double),
                ("query_construction_time", C.c_double), ("data_retrieval_time", C.c_double),
                ("computation_time", C.c_double), ("pv_computation_time", C.c_double),
                ("recalculate_query_indices_time", C.c_double), ("total_time", C.c_double),
                ("join_kernel_time", C.c_double), ("candidate_rows", C.c_int64),
                ("iterations", C.c_int32), ("reserved", C.c_int32), ("host_traversals", C.c_int64),
                ("libm_checks", C.c_int64)]


class EncodeDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("K", C.c_int32), ("codebook", C.c_void_p), ("C", C.c_int32),
                ("coarse", C.c_void_p)]


def encode(codebook, vectors, coarse=None, device=0):
    """Index build, encoding step: (cell[N] or None, codes[N, m]) of `vectors` for a trained
    codebook [m][K][s] and optional coarse quantizer [C][d] (codes of the residuals then)."""
    lib = load()
    cb, v = _f32(codebook), _f32(vectors)
    m, K, s_ = cb.shape
    d = m * s_
    v = v.reshape(-1, d)
    co = None if coarse is None else _f32(coarse)
    desc = EncodeDesc(d, m, K, _p(cb), 0 if co is None else co.shape[0], _p(co))
    codes = np.empty((v.shape[0], m), np.int16)
    cell = None if co is None else np.empty(v.shape[0], np.int32)
    _check(lib.freddy_gpu_encode(C.byref(desc), device, _p(v), C.c_int64(v.shape[0]), _p(cell), _p(codes)))
    return cell, codes


class InsertDesc(C.Structure):
    _fields_ = [("d", C.c_int32), ("pq_m", C.c_int32), ("pq_K", C.c_int32), ("pq_codebook", C.c_void_p),
                ("res_m", C.c_int32), ("res_K", C.c_int32), ("residual_codebook", C.c_void_p),
                ("C", C.c_int32), ("coarse", C.c_void_p),
                ("ivpq_m", C.c_int32), ("ivpq_K", C.c_int32), ("ivpq_codebook", C.c_void_p),
                ("multi_positions", C.c_int32), ("multi_codes", C.c_int32), ("coarse_multi", C.c_void_p)]


def insert_quantize(vectors, pq_codebook=None, residual_codebook=None, coarse=None, ivpq_codebook=None, coarse_multi=None,
                    device=0):
    """insert_batch's quantisation of new vectors (freddy.c:1557-1623) on the device.  Returns a dict with the
    arrays of the parts that were given: pq_codes, coarse_id, residual_codes, ivpq_codes, coarse_multi_codes."""
    lib = load()
    v = _f32(vectors)
    n, d = v.shape
    keep = []

    def cb(a):
        if a is None:
            return 0, 0, None
        a = _f32(a)
        keep.append(a)
        return a.shape[0], a.shape[1], _p(a)

    pm, pk, pp = cb(pq_codebook)
    rm, rk, rp = cb(residual_codebook)
    im, ik, ip = cb(ivpq_codebook)
    mp, mk, mpp = cb(coarse_multi)
    co = None if coarse is None else _f32(coarse)
    desc = InsertDesc(d, pm, pk, pp, rm, rk, rp, 0 if co is None else co.shape[0], _p(co), im, ik, ip, mp, mk, mpp)
    out = {}
    if pp: out["pq_codes"] = np.empty((n, pm), np.int16)
    if rp:
        out["coarse_id"] = np.empty(n, np.int32)
        out["residual_codes"] = np.empty((n, rm), np.int16)
    if ip: out["ivpq_codes"] = np.empty((n, im), np.int16)
    if mpp: out["coarse_multi_codes"] = np.empty((n, mp), np.int16)
    _check(lib.freddy_gpu_insert_quantize(C.byref(desc), device, _p(v), C.c_int64(n), _p(out.get("pq_codes")), _p(out.get("coarse_id")),
                                          _p(out.get("residual_codes")), _p(out.get("ivpq_codes")), _p(out.get("coarse_multi_codes"))))
    return out


def kmeans(vectors, k, iters=10, init_rows=None, device=0):
    """Quantizer training on the device (quantizer_creation.py:13-52): (centroids[k, d], assignment[n])."""
    lib = load()
    v = _f32(vectors)
    cent = np.empty((int(k), v.shape[1]), np.float32)
    assign = np.empty(v.shape[0], np.int32)
    ir = None if init_rows is None else _i32(init_rows)
    _check(lib.freddy_gpu_kmeans(device, _p(v), C.c_int64(v.shape[0]), v.shape[1], int(k), int(iters), _p(ir), _p(cent), _p(assign)))
    return cent, assign


def train_pq_codebook(vectors, m, K, iters=10, seed=0, device=0):
    """create_quantizer (quantizer_creation.py:13-30): one k-means per sub-vector position -> [m][K][d/m]."""
    v = _f32(vectors)
    s_ = v.shape[1] // m
    rng = np.random.default_rng(seed)
    out = np.empty((m, K, s_), np.float32)
    for p in range(m):
        init = rng.choice(v.shape[0], K, replace=v.shape[0] < K).astype(np.int32)
        out[p], _ = kmeans(np.ascontiguousarray(v[:, p * s_:(p + 1) * s_]), K, iters, init, device)
    return out


class PinnedBuffer:
    """Pinned host memory from freddy_gpu_host_alloc as a numpy array: query batches written into it skip the
    staging copy of the host-buffer calls (include/freddy_gpu.h)."""

    def __init__(self, shape, dtype=np.float32):
        self.lib = load()
        self.ptr = C.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _check(self.lib.freddy_gpu_host_alloc(C.byref(self.ptr), n))
        buf = (C.c_char * max(n, 1)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self.ptr:
            self.array = None
            self.lib.freddy_gpu_host_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
