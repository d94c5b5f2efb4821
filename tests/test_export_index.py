"""tools/export_index.py: table dumps (psql \\copy ... CSV with encode(vector, 'hex')) -> FRDYIDX1 file.
CPU part: the file holds exactly the dumped arrays; GPU part (marked): a session imported from the
exported file answers like the oracle on the same tables."""
import csv
import os
import subprocess
import sys

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dump(path, columns):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        for row in zip(*columns):
            w.writerow([("\\x" + c.tobytes().hex()) if isinstance(c, np.ndarray) else c for c in row])


def _entries(cb):
    m, K, s_ = cb.shape
    pos, code = np.divmod(np.arange(m * K), K)
    return pos.tolist(), code.tolist(), list(cb.reshape(m * K, s_))


def _write_dumps(d, N=600):
    x = util.corpus(20000).numpy()[:N]
    ids = np.arange(1, N + 1)
    from freddy_amd import index_build as ib
    import torch
    xt = torch.from_numpy(x)
    pq = ib.build_pq_index(xt, m=12, K=16, train_size=N, iters=2, seed=1)
    ivf = ib.build_ivf_index(xt, C=6, m=12, K=16, train_size=N, iters=2, seed=2)
    iv = ib.build_ivpq_index(xt, m=30, K=8, k_coarse=4, train_size=N, iters=2, seed=3)
    _dump(os.path.join(d, "google_vecs_norm.csv"), [ids.tolist(), list(x)])
    p, c, v = _entries(pq["codebook"])
    _dump(os.path.join(d, "pq_codebook.csv"), [p, c, v, list(range(1, len(p) + 1))])
    _dump(os.path.join(d, "pq_quantization.csv"), [pq["ids"].tolist(), list(pq["codes"])])
    _dump(os.path.join(d, "coarse_quantization.csv"), [list(range(6)), list(ivf["coarse"])])
    p, c, v = _entries(ivf["codebook"])
    _dump(os.path.join(d, "residual_codebook.csv"), [p, c, v, [3] * len(p)])
    cell = np.repeat(np.arange(6), np.diff(ivf["list_off"]))
    _dump(os.path.join(d, "fine_quantization.csv"), [ivf["ids"].tolist(), cell.tolist(), list(ivf["codes"])])
    p, c, v = _entries(iv["codebook"])
    _dump(os.path.join(d, "codebook_ivpq.csv"), [p, c, v, [1] * len(p)])
    p, c, v = _entries(iv["coarse"])
    _dump(os.path.join(d, "coarse_quantization_ivpq.csv"), [p, c, v])
    _dump(os.path.join(d, "fine_quantization_ivpq.csv"), [iv["ids"].tolist(), iv["coarse_id"].tolist(), list(iv["codes"])])
    _dump(os.path.join(d, "stat.csv"), [list(range(17)), [repr(float(f)) for f in iv["stats"]]])
    return x, pq, ivf, iv


def _export(d, out):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_index.py"), "--csv-dir", str(d), "--out", str(out)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]


def _parse(path):
    import struct
    raw = open(path, "rb").read()
    assert raw[:8] == b"FRDYIDX1"
    (n,) = struct.unpack_from("<I", raw, 8)
    off, got = 12, {}
    for _ in range(n):
        (nl,) = struct.unpack_from("<H", raw, off); off += 2
        name = raw[off:off + nl].decode(); off += nl
        dt, nd = raw[off], raw[off + 1]; off += 2
        dims = struct.unpack_from("<%dQ" % nd, raw, off); off += 8 * nd
        off = (off + 7) // 8 * 8
        dtype = [np.float32, np.int32, np.int16][dt]
        count = int(np.prod(dims))
        got[name] = np.frombuffer(raw, dtype, count, off).reshape(dims)
        off = (off + count * np.dtype(dtype).itemsize + 7) // 8 * 8
    return got


def test_exported_file_holds_the_dumped_tables(tmp_path):
    x, pq, ivf, iv = _write_dumps(str(tmp_path))
    out = tmp_path / "all.fidx"
    _export(tmp_path, out)
    got = _parse(out)
    assert np.array_equal(got["google_vecs_norm.vector"].view(np.uint32), x.view(np.uint32))
    assert np.array_equal(got["pq_quantization.vector"], pq["codes"]) and got["pq_quantization.vector"].dtype == np.int16
    assert np.array_equal(got["pq_codebook.vector"].reshape(pq["codebook"].shape), pq["codebook"])
    assert got["pq_codebook.count"].tolist() == list(range(1, 12 * 16 + 1))
    assert np.array_equal(got["fine_quantization.coarse_id"], np.repeat(np.arange(6), np.diff(ivf["list_off"])))
    assert np.array_equal(got["stat.coarse_freq"], iv["stats"])
    assert np.array_equal(got["coarse_quantization_ivpq.vector"].reshape(iv["coarse"].shape), iv["coarse"])


@pytest.mark.gpu
def test_session_imported_from_an_exported_file(tmp_path, oracle):
    from freddy_amd import udf
    x, pq, ivf, iv = _write_dumps(str(tmp_path))
    out = tmp_path / "all.fidx"
    _export(tmp_path, out)
    s = udf.Session()
    s.import_index(out)
    q = x[17]
    r = s.pq_search(q, 5)
    e = oracle.pq_search(oracle.pq_table(pq["codebook"], pq["ids"], pq["codes"]), q, 5)
    assert np.array_equal(r["id"], e["id"]) and np.array_equal(r["distance"].view(np.uint32), e["dist"].view(np.uint32))
    r = s.ivfadc_search(q, 5)
    e = oracle.ivfadc_search(oracle.ivf_table(ivf["coarse"], ivf["codebook"], ivf["list_off"], ivf["ids"], ivf["codes"]), q, 5, 3)
    assert np.array_equal(r["id"], e["id"]) and np.array_equal(r["distance"].view(np.uint32), e["dist"].view(np.uint32))
    new_ids = s.insert_batch(x[:2] * np.float32(0.998))     # the count columns came with the file
    assert new_ids.tolist() == [601, 602]
    s.close()


@pytest.mark.gpu
def test_native_index_build(oracle):
    """ivfadc.py on the native ABI only: k-means training + encoding on the device; the index it produces is
    searched like any other (and identically by the oracle)."""
    from freddy_amd import gpu, index_build as ib
    x = util.corpus(20000).numpy()[:6000]
    t = ib.build_ivf_index_native(x, C=16, m=12, K=32, train_size=3000, iters=3, seed=5)
    assert t["list_off"][-1] == 6000 and (np.diff(t["list_off"]) > 0).sum() >= 8
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    qs = x[::97]
    gi, gd = idx.search(qs, 5, 3)
    util.assert_same_lists(gi, gd, oracle.ivfadc_search_many(ot, qs, 5, 3), "native index")
    assert (gi[:, 0] == np.arange(0, 6000, 97) + 1).mean() > 0.8     # a row usually finds itself first
    idx.close()
