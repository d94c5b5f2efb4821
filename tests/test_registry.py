"""The registry of live backends per physical GPU (postgres-word2vec_amd/csrc/registry.h; SURVEY 8b: one backend = one process).

registry.h is plain C++ (no HIP): this test compiles it on its own with g++ behind a three-function C wrapper and drives it from
several PROCESSES through a registry object of its own name -- what core.hip's backend_handles / backends_other / backend_busy do
in the product library.  Properties: a backend sees the others of ITS GPU only (bench.py --gpus N: one process per GPU are not
neighbours); *_VISIBLE_DEVICES lists map device indices to physical ordinals; what cannot be mapped is everybody's neighbour; a slot
is free again when its process has freed its handles or died."""
import ctypes
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "postgres-word2vec_amd", "csrc")

WRAPPER = """
#include "registry.h"
extern "C" {
void reg_handles(int delta, int device) { freddy::registry::handles(delta, device); }
int reg_others(int searching, int device) { return freddy::registry::others(searching != 0, device); }
void reg_busy(int delta) { freddy::registry::busy(delta); }
int reg_physical(int device) { return freddy::registry::physical_device(device); }
}
"""

CHILD = """
import ctypes, os, sys, time
lib = ctypes.CDLL(sys.argv[1])
device, busy = int(sys.argv[2]), int(sys.argv[3])
lib.reg_handles(1, device)
if busy:
    lib.reg_busy(1)
print("ready", lib.reg_others(0, device), flush=True)
sys.stdin.readline()          # the parent says when to go
if busy:
    lib.reg_busy(-1)
lib.reg_handles(-1, device)
print("others_after_release", lib.reg_others(0, -1), flush=True)
"""


@pytest.fixture(scope="module")
def lib_path(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not on PATH")
    d = tmp_path_factory.mktemp("registry")
    src = d / "wrap.cpp"
    src.write_text(WRAPPER)
    so = d / "libreg.so"
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I" + CSRC, str(src), "-o", str(so), "-lrt", "-pthread"], check=True)
    return str(so)


def _env(name, **extra):
    env = {k: v for k, v in os.environ.items() if not k.endswith("_VISIBLE_DEVICES") and not k.startswith("FREDDY_GPU_REGISTRY")}
    env["FREDDY_GPU_REGISTRY_NAME"] = name
    env.update(extra)
    return env


def _child(lib_path, name, device, busy=0, **extra):
    p = subprocess.Popen([sys.executable, "-c", CHILD, lib_path, str(device), str(busy)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                         text=True, env=_env(name, **extra))
    line = p.stdout.readline().split()
    assert line and line[0] == "ready", line
    return p, int(line[1])


def _probe(lib_path, name, code, **extra):
    """a fresh process (no handles of its own) evaluates `code` with lib = the wrapper"""
    out = subprocess.run([sys.executable, "-c", "import ctypes, sys\nlib = ctypes.CDLL(sys.argv[1])\nprint(" + code + ")", lib_path],
                         capture_output=True, text=True, env=_env(name, **extra), check=True)
    return eval(out.stdout.strip())


def _finish(procs):
    for p in procs:
        p.stdin.write("\n")
        p.stdin.flush()
    outs = [p.communicate(timeout=30)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    return outs


@pytest.fixture()
def name():
    n = f"/freddy_gpu_backends_test.{os.getpid()}"
    yield n
    try:
        os.unlink("/dev/shm" + n)
    except OSError:
        pass


def test_backends_of_other_gpus_are_not_neighbours(lib_path, name):
    a, seen_a = _child(lib_path, name, 0)
    b, seen_b = _child(lib_path, name, 1)
    c, seen_c = _child(lib_path, name, 0, busy=1)
    try:
        assert (seen_a, seen_b, seen_c) == (0, 0, 1)      # what each found on its own GPU when it registered
        got = _probe(lib_path, name, "[lib.reg_others(0, 0), lib.reg_others(0, 1), lib.reg_others(0, 2), lib.reg_others(0, -1),"
                                     " lib.reg_others(1, 0), lib.reg_others(1, 1), lib.reg_others(1, -1)]")
        assert got == [2, 1, 0, 3, 1, 0, 1]
    finally:
        outs = _finish([a, b, c])
    assert _probe(lib_path, name, "lib.reg_others(0, -1)") == 0   # every slot released
    assert all("others_after_release" in o for o in outs)


def test_visible_devices_lists_map_to_physical_ordinals(lib_path, name):
    assert _probe(lib_path, name, "[lib.reg_physical(0), lib.reg_physical(3), lib.reg_physical(-1)]") == [0, 3, -1]
    assert _probe(lib_path, name, "[lib.reg_physical(0), lib.reg_physical(1), lib.reg_physical(2)]", HIP_VISIBLE_DEVICES="5, 2") == [5, 2, -1]
    assert _probe(lib_path, name, "[lib.reg_physical(0), lib.reg_physical(1)]", CUDA_VISIBLE_DEVICES="7") == [7, -1]
    assert _probe(lib_path, name, "[lib.reg_physical(0), lib.reg_physical(1)]", ROCR_VISIBLE_DEVICES="4,6", HIP_VISIBLE_DEVICES="1") == [6, -1]
    assert _probe(lib_path, name, "lib.reg_physical(0)", HIP_VISIBLE_DEVICES="GPU-0123abcd") == -1
    # device 0 of a process that sees only physical GPU 3 is a neighbour of device 3 of a process that sees them all -- and of nobody on GPU 0
    a, _ = _child(lib_path, name, 0, HIP_VISIBLE_DEVICES="3")
    try:
        assert _probe(lib_path, name, "[lib.reg_others(0, 3), lib.reg_others(0, 0)]") == [1, 0]
        # a device that cannot be mapped is a neighbour of everybody
        assert _probe(lib_path, name, "lib.reg_others(0, 0)", HIP_VISIBLE_DEVICES="GPU-0123abcd") == 1
    finally:
        _finish([a])
    u, _ = _child(lib_path, name, 0, ROCR_VISIBLE_DEVICES="GPU-feedbeef")
    try:
        assert _probe(lib_path, name, "[lib.reg_others(0, 0), lib.reg_others(0, 5)]") == [1, 1]
    finally:
        _finish([u])


def test_a_dead_backends_slot_does_not_count_and_is_reused(lib_path, name):
    a, _ = _child(lib_path, name, 0, busy=1)
    assert _probe(lib_path, name, "[lib.reg_others(0, 0), lib.reg_others(1, 0)]") == [1, 1]
    a.kill()
    a.wait()
    assert _probe(lib_path, name, "[lib.reg_others(0, 0), lib.reg_others(1, 0)]") == [0, 0]
    procs = [_child(lib_path, name, 0)[0] for _ in range(3)]
    try:
        assert _probe(lib_path, name, "lib.reg_others(0, 0)") == 3
    finally:
        _finish(procs)


def test_registry_can_be_switched_off(lib_path, name):
    a, _ = _child(lib_path, name, 0)
    try:
        assert _probe(lib_path, name, "lib.reg_others(0, 0)", FREDDY_GPU_REGISTRY="0") == 0
        assert _probe(lib_path, name, "lib.reg_others(0, 0)") == 1
    finally:
        _finish([a])
