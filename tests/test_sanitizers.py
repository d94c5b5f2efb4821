"""SURVEY 5 "race detection / sanitizers": the CPU-side C / C++ of this repository under AddressSanitizer +
UndefinedBehaviorSanitizer -- the oracle (oracle/freddy_oracle.c), the PostgreSQL-free parts of the hosts (pg/freddy_pure.h)
and the GPU-free parts of the host mirror (postgres-word2vec_amd/host/freddy_udf.cpp: parsers, the index-file reader /
writer, row emission, configuration).  The sanitised shared objects are loaded by a CHILD python that runs with libasan
preloaded and re-runs the ordinary CPU tests against them; any report makes the child fail (halt_on_error, UBSan without
recovery).  CPU only: GPU sanitizer runs are not available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run_under_asan(test_args, extra_env):
    lib = _libasan()
    if lib is None:
        pytest.skip("gcc has no libasan in this image")
    env = dict(os.environ)
    env.update(extra_env)
    # libstdc++ before libasan's interceptors resolve __cxa_throw (C++ exceptions inside the host mirror)
    libstdcxx = subprocess.check_output(["gcc", "-print-file-name=libstdc++.so.6"], text=True).strip()
    env["LD_PRELOAD"] = lib + (":" + libstdcxx if os.path.isabs(libstdcxx) else "")
    # leaks: python itself never frees everything; the drivers free what they allocate (checked by the allocator's own
    # bookkeeping: double free / use after free / overflow are what ASan is here for)
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=1:allocator_may_return_null=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["OMP_NUM_THREADS"] = "4"
    cp = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + test_args, cwd=ROOT, env=env,
                        capture_output=True, text=True, timeout=1500)
    tail = (cp.stdout[-3000:] + "\n" + cp.stderr[-3000:])
    assert cp.returncode == 0, "sanitizer run failed:\n" + tail
    assert "AddressSanitizer" not in cp.stderr and "runtime error:" not in cp.stderr, tail
    return cp


def test_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "_asan", "libfreddy_oracle_asan.so")
    cp = _run_under_asan(["tests/test_oracle.py"], {"FREDDY_ORACLE_SO": so})
    assert " passed" in cp.stdout


def test_pg_pure_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "_asan", "libfreddy_oracle_asan.so")
    cp = _run_under_asan(["tests/test_pg_pure.py"], {"FREDDY_SANITIZE": "1", "FREDDY_ORACLE_SO": so})
    assert " passed" in cp.stdout


def test_host_mirror_gpu_free_parts_under_asan_and_ubsan():
    """freddy_udf.cpp: configuration functions and their errors, the FRDYIDX1 index-file writer / reader (bad magic, truncated
    groups), emit_row* -- everything tests/test_abi.py and tests/test_export_index.py exercise without a GPU."""
    gpu_so = os.path.join(ROOT, "postgres-word2vec_amd", "libfreddy_gpu.so")
    if not os.path.exists(gpu_so):
        pytest.skip("libfreddy_gpu.so not built (run __graft_entry__.build())")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "postgres-word2vec_amd", "host"), "-s", "asan"])
    so = os.path.join(ROOT, "postgres-word2vec_amd", "_asan", "libfreddy_host_asan.so")
    cp = _run_under_asan(["tests/test_abi.py", "tests/test_export_index.py", "-m", "not gpu"], {"FREDDY_HOST_SO": so})
    assert " passed" in cp.stdout
