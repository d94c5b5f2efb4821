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
