"""pg/freddy_pure.h -- the PostgreSQL-free parts of the PostgreSQL hosts (the staleness decision, the bytea payload codecs,
updateCodebook's bookkeeping) -- compiled with gcc HERE and exercised: PostgreSQL itself is not in this image, so this is
the part of pg/ that can meet a compiler and a test."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def drv(tmp_path_factory):
    so = tmp_path_factory.mktemp("pgpure") / "libpgpure.so"
    flags = ["-O2"]
    if os.environ.get("FREDDY_SANITIZE") == "1":   # tests/test_sanitizers.py re-runs this file under ASan + UBSan
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call(["gcc"] + flags + ["-std=c11", "-Wall", "-Werror", "-ffp-contract=off", "-fPIC", "-shared", "-o", str(so),
                           os.path.join(ROOT, "tests", "c", "pg_pure_driver.c")])
    lib = C.CDLL(str(so))
    lib.drv_stamp_size.restype = C.c_size_t
    return lib


class Stamp(C.Structure):
    _fields_ = [("n_tabs", C.c_int), ("rel", C.c_uint32 * 5), ("filenode", C.c_uint32 * 5), ("appends", C.c_int64 * 5),
                ("rewrites", C.c_int64 * 5), ("weak", C.c_int64 * 5), ("max_id", C.c_int32), ("d", C.c_int), ("m", C.c_int)]


def mk(n, watched=True, **kw):
    s = Stamp()
    s.n_tabs = n
    for i in range(n):
        s.rel[i] = 1000 + i; s.filenode[i] = 2000 + i
        s.appends[i] = 5 if watched else -1; s.rewrites[i] = 7 if watched else -1; s.weak[i] = 0 if watched else 8192 * (i + 1)
    s.max_id = 100
    for k, v in kw.items():
        name, i = k.rsplit("_", 1)
        getattr(s, name)[int(i)] = v
    return s


CURRENT, CATCH_UP, STALE = 0, 1, 2


def compare(drv, old, now, row_max=-1, mask=0):
    a, c = C.c_int(-1), C.c_int(-1)
    r = drv.drv_compare(C.byref(old), C.byref(now), C.c_longlong(row_max), C.c_uint(mask), C.byref(a), C.byref(c))
    return r, a.value, c.value


def test_stamp_layout(drv):
    assert drv.drv_stamp_size() == C.sizeof(Stamp)


def test_staleness_decisions_with_the_watch_triggers(drv):
    old = mk(3)
    assert compare(drv, old, mk(3)) == (CURRENT, 0, 0)
    assert compare(drv, old, mk(3, appends_0=6)) == (CATCH_UP, 1, 0)                      # insert_batch's INSERTs
    assert compare(drv, old, mk(3, rewrites_1=8)) == (CATCH_UP, 0, 1)                     # ... and its codebook UPDATEs
    assert compare(drv, old, mk(3, appends_0=9, rewrites_1=30)) == (CATCH_UP, 1, 1)
    assert compare(drv, old, mk(3, rewrites_0=8))[0] == STALE                             # UPDATE / DELETE of quantization rows
    assert compare(drv, old, mk(3, appends_1=6))[0] == STALE                              # new codebook rows: another shape
    assert compare(drv, old, mk(3, appends_2=6))[0] == STALE                              # the coarse quantizer changed
    assert compare(drv, old, mk(3, rewrites_2=8))[0] == STALE
    assert compare(drv, old, mk(3, rel_0=4242))[0] == STALE                               # set_*() pointed the role at another table
    assert compare(drv, old, mk(3, filenode_1=1))[0] == STALE                             # TRUNCATE / VACUUM FULL
    assert compare(drv, old, mk(2))[0] == STALE
    assert compare(drv, old, mk(3, watched=False))[0] == STALE                            # the watch script went away
    # the ivpq handle: INSERTs into the vector table [3] accompany appended rows, anything else there is a change
    old5 = mk(5)
    assert compare(drv, old5, mk(5, appends_3=6), mask=1 << 3) == (CURRENT, 0, 0)
    assert compare(drv, old5, mk(5, appends_3=6))[0] == STALE
    assert compare(drv, old5, mk(5, rewrites_3=8), mask=1 << 3)[0] == STALE
    assert compare(drv, old5, mk(5, appends_0=6, appends_3=6, rewrites_1=9), mask=1 << 3) == (CATCH_UP, 1, 1)
    assert compare(drv, old5, mk(5, rewrites_4=9), mask=1 << 3)[0] == STALE               # the statistics table


def test_staleness_decisions_without_the_watch_triggers(drv):
    old = mk(3, watched=False)
    assert compare(drv, old, mk(3, watched=False), row_max=100) == (CURRENT, 0, 0)
    assert compare(drv, old, mk(3, watched=False), row_max=103) == (CATCH_UP, 1, 0)       # max(id) grew, the file did not
    assert compare(drv, old, mk(3, watched=False, weak_0=99999), row_max=100) == (CATCH_UP, 1, 0)   # the file grew: look for rows
    assert compare(drv, old, mk(3, watched=False, weak_1=12345), row_max=100) == (CATCH_UP, 0, 1)   # sum(count) of the codebook
    assert compare(drv, old, mk(3, watched=False), row_max=90)[0] == STALE                # rows vanished
    assert compare(drv, old, mk(3, watched=False, weak_2=1), row_max=100)[0] == STALE     # the coarse quantizer's file changed
    assert compare(drv, old, mk(3, watched=False, filenode_0=9), row_max=100)[0] == STALE


def test_rolled_back_insert_batch_makes_the_handle_stale(drv):
    """insert_batch refreshes the pinned handles INSIDE its transaction (appended rows, max_id and the stamp
    advance).  After ROLLBACK the generation counters are lower than the stamp: that must read as STALE, never as an
    append -- the catch-up fetch `id > max_id` would find nothing, the phantom rows would stay pinned and the ids they
    hold would be handed out again by the next insert_batch."""
    before = mk(3)                                   # appends 5, rewrites 7, max_id 100
    inside = mk(3, appends_0=6, rewrites_1=8)        # what the handle was brought up to date with inside the transaction
    inside.max_id = 103
    assert compare(drv, before, inside) == (CATCH_UP, 1, 1)
    after_rollback = mk(3)                           # the counters are back where they were
    assert compare(drv, inside, after_rollback)[0] == STALE
    assert compare(drv, inside, mk(3, appends_0=6, rewrites_1=7))[0] == STALE      # only the codebook UPDATEs rolled back (savepoint)
    assert compare(drv, inside, mk(3, appends_0=5, rewrites_1=8))[0] == STALE      # only the INSERTs rolled back
    # a later committed insert_batch reuses ids 101.. : the append counter is equal again -- the catch-up fetch decides
    ids = np.array([101, 102, 103], np.int32)
    ok = lambda mx, a: drv.drv_catch_up_ok(mx, C.c_longlong(len(a)), a.ctypes.data_as(C.c_void_p))
    assert ok(100, ids) == 1                                                        # the continuation of what is pinned
    assert ok(103, np.zeros(0, np.int32)) == 0                                      # append flagged, nothing above max_id: phantom rows
    assert ok(100, np.array([102, 103], np.int32)) == 0                             # a gap: a row committed out of id order elsewhere
    assert ok(100, np.array([101, 103], np.int32)) == 0
    assert ok(2**31 - 2, np.array([2**31 - 1], np.int32)) == 1                      # no overflow at the top of int32


def test_handles_mutated_in_an_aborted_transaction_are_dropped(drv):
    # level 0 = not touched inside an open transaction; 1 = top level; 2.. = savepoints
    assert drv.drv_survives_abort(0, 1) == 1
    assert drv.drv_survives_abort(1, 1) == 0         # ROLLBACK of the transaction that appended
    assert drv.drv_survives_abort(2, 1) == 0
    assert drv.drv_survives_abort(1, 2) == 1         # ROLLBACK TO SAVEPOINT: the append happened outside it
    assert drv.drv_survives_abort(2, 2) == 0
    assert drv.drv_survives_abort(3, 2) == 0
    assert drv.drv_level_after_commit(2, 2) == 1     # RELEASE SAVEPOINT: the change now belongs to the parent
    assert drv.drv_level_after_commit(1, 2) == 1
    assert drv.drv_level_after_commit(1, 1) == 0     # COMMIT: durable
    assert drv.drv_level_after_commit(3, 2) == 1


def test_bytea_payload_codecs(drv):
    v = np.arange(300, dtype=np.float32) * np.float32(0.25)
    out = np.zeros(300, np.float32)
    assert drv.drv_payload_f32(v.ctypes.data_as(C.c_void_p), C.c_size_t(1200), -1, out.ctypes.data_as(C.c_void_p)) == 300
    assert np.array_equal(out.view(np.uint32), v.view(np.uint32))
    assert drv.drv_payload_f32(v.ctypes.data_as(C.c_void_p), C.c_size_t(1200), 300, out.ctypes.data_as(C.c_void_p)) == 300
    assert drv.drv_payload_f32(v.ctypes.data_as(C.c_void_p), C.c_size_t(1200), 25, out.ctypes.data_as(C.c_void_p)) == -1    # another dimensionality
    assert drv.drv_payload_f32(v.ctypes.data_as(C.c_void_p), C.c_size_t(1199), -1, out.ctypes.data_as(C.c_void_p)) == -1    # not whole floats
    assert drv.drv_payload_f32(v.ctypes.data_as(C.c_void_p), C.c_size_t(0), -1, out.ctypes.data_as(C.c_void_p)) == 0
    c = np.array([3, 1023, 0, 517, 9, 1, 2, 3, 4, 5, 6, 7], np.int16)
    oc = np.zeros(12, np.int16)
    assert drv.drv_payload_i16(c.ctypes.data_as(C.c_void_p), C.c_size_t(24), 12, oc.ctypes.data_as(C.c_void_p)) == 12
    assert np.array_equal(oc, c)
    assert drv.drv_payload_i16(c.ctypes.data_as(C.c_void_p), C.c_size_t(23), 12, oc.ctypes.data_as(C.c_void_p)) == -1
    assert drv.drv_payload_i16(c.ctypes.data_as(C.c_void_p), C.c_size_t(24), 30, oc.ctypes.data_as(C.c_void_p)) == -1


@pytest.mark.parametrize("m,K,s,n", [(12, 64, 25, 9), (30, 32, 10, 40), (3, 5, 4, 1)])
def test_codebook_bookkeeping_equals_the_oracles_update_codebook(drv, oracle, m, K, s, n):
    """pg/freddy_insert.c runs the 1-NN search on the device and then freddy_update_codebook_known_codes: given the codes
    the oracle's literal updateCodebook finds, the bookkeeping must leave the same counts and -- after the "%f" text round
    trip updateCodebookRelation applies to the entries it writes -- the same vectors.  The table's row order matters
    (index_utils.c:925-939: ONE `nearestCentroidRaw` pointer, left at the entry that improved last in SPI order; insert_batch's
    own UPDATEs move tuples): position-major and shuffled entries, each against the oracle scanning in that same order."""
    rng = np.random.default_rng(m * 1000 + K)
    cb = (rng.standard_normal((m, K, s)) * 0.2).astype(np.float32)
    counts = rng.integers(1, 50, m * K).astype(np.int32)
    vecs = (rng.standard_normal((n, m * s)) * 0.2).astype(np.float32)
    base = oracle.update_codebook(cb, counts, vecs)
    differs = False
    for order in (np.arange(m * K), rng.permutation(m * K), rng.permutation(m * K)):
        exp_cb, exp_cnt, codes, incs = oracle.update_codebook(cb, counts, vecs, order=order)
        assert np.array_equal(codes, base[2])      # (no two entries equally near in these tables: the codes do not depend on the order)
        differs = differs or not np.array_equal(exp_cb.view(np.uint32), base[0].view(np.uint32))
        work, cnt = cb.copy(), counts.copy()
        got_incs = np.zeros(m * K, np.int32)
        order = order.astype(np.int32)
        drv.drv_update_codebook(work.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), m, K, s, codes.ctypes.data_as(C.c_void_p),
                                n, order.ctypes.data_as(C.c_void_p), got_incs.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got_incs, incs)
        written = incs > 0
        stored = cb.reshape(m * K, s).copy()
        stored[written] = oracle.text_roundtrip(work.reshape(m * K, s)[written])
        assert np.array_equal(stored.view(np.uint32), exp_cb.reshape(m * K, s).view(np.uint32))
        new_cnt = counts.copy()
        new_cnt[written] = cnt[written]
        assert np.array_equal(new_cnt, exp_cnt)
        assert np.array_equal(cnt[~written], counts[~written])
    assert differs or m == 1, "a shuffled table must change which vector a row adds (else this test does not see the order)"
