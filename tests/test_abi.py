"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares
(no compute calls here -- those are the -m gpu tests)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(freddy_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from freddy_amd import gpu
    lib = gpu.load()
    names = declared_functions(os.path.join(ROOT, "include", "freddy_gpu.h"))
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/freddy_gpu.h but not exported"
    assert set(gpu.EXPORTS) <= set(names)


def test_argument_errors_are_reported_without_a_gpu():
    from freddy_amd import gpu
    lib = gpu.load()
    h = ctypes.c_void_p()
    assert lib.freddy_gpu_pin_pq(None, 0, ctypes.byref(h)) == -1
    assert b"NULL" in lib.freddy_gpu_last_error()
    assert lib.freddy_gpu_ivfadc_search(None, None, 1, 5, 3, ctypes.c_float(1000.0), 0, None, None) == -1
    assert lib.freddy_gpu_unpin(None) == 0


def test_product_never_references_the_oracle():
    """A product path that routes through oracle/ would void every parity claim."""
    pkg = os.path.join(ROOT, "postgres-word2vec_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "freddy_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    so = open(os.path.join(pkg, "libfreddy_gpu.so"), "rb").read()
    assert b"fo_sqdist" not in so and b"libfreddy_oracle" not in so


def test_host_mirror_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from freddy_amd import udf
    lib = udf.load()
    src = open(os.path.join(ROOT, "include", "freddy_udf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\(freddy_session_t\b|\b(freddy_[a-z0-9_]+)\s*\(", src)))
    flat = sorted({n for pair in names for n in pair if n})
    assert {"pq_search", "ivfadc_search", "pq_search_in", "pq_search_in_batch", "ivfadc_batch_search",
            "ivpq_search_in", "knn_join", "k_nearest_neighbour", "knn_in_exact", "grouping_pq",
            "analogy_3cosadd_pq", "analogy_3cosadd_ivfadc", "k_nearest_neighbour_pq", "k_nearest_neighbour_ivfadc",
            "k_nearest_neighbour_pq_pv", "k_nearest_neighbour_ivfadc_pv", "knn_in_pq",
            "k_nearest_neighbour_ivfadc_batch"} <= set(flat)
    for n in flat:
        assert hasattr(lib, n), f"{n} declared in include/freddy_udf.h but not exported"


def test_host_mirror_config_functions_and_errors():
    """set_*/get_* defaults (freddy--0.0.1.sql:188-194) and error texts, no GPU needed."""
    from freddy_amd import udf
    s = udf.Session()
    assert (s.get_w(), s.get_pvf(), s.get_alpha()) == (3, 20, 3)
    assert abs(s.get_confidence_value() - 0.8) < 1e-7
    s.set_w(7); s.set_pvf(5); s.set_alpha(11); s.set_confidence_value(0.5)
    assert (s.get_w(), s.get_pvf(), s.get_alpha()) == (7, 5, 11)
    import numpy as np
    import pytest
    with pytest.raises(udf.FreddyError, match="not loaded"):
        s.pq_search(np.zeros(300, np.float32), 5)
    with pytest.raises(udf.FreddyError, match="not loaded"):
        s.ivfadc_batch_search([1, 2, 3], 5)
    assert s.emit_row3((7, 42, 0.1234567)) == ("7", "42", "0.123457")
    s.close()


def test_index_file_format_roundtrip(tmp_path):
    """FRDYIDX1 (include/freddy_udf.h): what freddy_index_file_write emits, parsed here byte by byte."""
    import struct
    import numpy as np
    from freddy_amd import udf
    rng = np.random.default_rng(0)
    arrays = {"pq_codebook.pos": np.arange(7, dtype=np.int32),
              "pq_codebook.vector": rng.standard_normal((7, 5)).astype(np.float32),
              "pq_quantization.vector": rng.integers(0, 9, (3, 4)).astype(np.int16),
              "x.empty": np.zeros((0, 3), np.float32)}
    path = tmp_path / "t.fidx"
    udf.write_index_file(path, arrays)
    raw = path.read_bytes()
    assert raw[:8] == b"FRDYIDX1" and struct.unpack_from("<I", raw, 8)[0] == len(arrays)
    off, got = 12, {}
    for _ in range(len(arrays)):
        (nl,) = struct.unpack_from("<H", raw, off); off += 2
        name = raw[off:off + nl].decode(); off += nl
        dt, nd = raw[off], raw[off + 1]; off += 2
        dims = struct.unpack_from("<%dQ" % nd, raw, off); off += 8 * nd
        off = (off + 7) // 8 * 8
        dtype = [np.float32, np.int32, np.int16][dt]
        count = int(np.prod(dims))
        got[name] = np.frombuffer(raw, dtype, count, off).reshape(dims)
        off += count * np.dtype(dtype).itemsize
        off = (off + 7) // 8 * 8
    assert off >= len(raw) and set(got) == set(arrays)
    for k, v in arrays.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v)
    s = udf.Session()
    import pytest
    with pytest.raises(udf.FreddyError, match="no complete table group"):
        s.import_index(path)
    bad = tmp_path / "bad.fidx"
    bad.write_bytes(b"NOTANIDX" + raw[8:])
    with pytest.raises(udf.FreddyError, match="bad magic"):
        s.import_index(bad)
    s.close()


def test_pg_hosts_bind_the_declared_abi():
    """pg/*.c cannot be compiled in this image (no PostgreSQL headers): at least every freddy_gpu_* entry point
    they call must be declared in include/freddy_gpu.h, and the SRF symbols the SQL script binds must be there."""
    import re
    hdr = open(os.path.join(ROOT, "include", "freddy_gpu.h")).read()
    declared = set(re.findall(r"\b(freddy_gpu_\w+)\s*\(", hdr))
    srcs = {f: open(os.path.join(ROOT, "pg", f)).read() for f in ("freddy_srf.c", "ivpq_search_in.c", "freddy_gpu_glue.c", "freddy_insert.c")}
    called = set()
    for text in srcs.values():
        called |= set(re.findall(r"\b(freddy_gpu_\w+)\s*\(", text))
    assert called and called <= declared, called - declared
    v1 = set(re.findall(r"PG_FUNCTION_INFO_V1\((\w+)\)", srcs["freddy_srf.c"] + srcs["ivpq_search_in.c"] + srcs["freddy_insert.c"]))
    assert v1 == {"pq_search", "ivfadc_search", "pq_search_in", "pq_search_in_batch", "ivfadc_batch_search", "ivpq_search_in",
                  "insert_batch", "grouping_pq"}
    # every glue function a host calls is declared in the glue header and defined in the glue
    glue_h = open(os.path.join(ROOT, "pg", "freddy_gpu_glue.h")).read()
    glue_calls = set()
    for f in ("freddy_srf.c", "ivpq_search_in.c", "freddy_insert.c"):
        glue_calls |= set(re.findall(r"\b(freddy_glue_\w+)\s*\(", srcs[f]))
    for fn in glue_calls:
        assert re.search(r"\b%s\s*\(" % fn, glue_h), f"{fn} is not declared in pg/freddy_gpu_glue.h"
        assert re.search(r"^[\w \*]+\b%s\s*\(" % fn, srcs["freddy_gpu_glue.c"], re.M), f"{fn} is not defined in pg/freddy_gpu_glue.c"
    mk = open(os.path.join(ROOT, "pg", "Makefile")).read()
    for fn in ("pq_search", "ivfadc_search", "pq_search_in", "pq_search_in_batch", "ivfadc_batch_search", "grouping_pq", "insert_batch"):
        assert f"-D{fn}=freddy_cpu_{fn}" in mk      # the reference's own copies step aside
