"""RCCL called directly on the caller's HIP stream (the gather of the per-shard top-k lists).

torch.distributed stays what starts the ranks, exchanges the communicator id, barriers and reduces the timings; the DATA-PATH
collective -- one all_gather of 40 KB per rank and step -- is enqueued by a single ncclAllGather call on the very stream the step's
search kernels are on.  Measured on one MI355X with a one-rank group and four batches in flight (profiles/r06_collective_1rank.txt):
through c10d (ProcessGroupNCCL: work objects, events, its watchdog) the same steps ran at 0.65 - 0.80 of the rate without a
collective, whichever stream the collective used; with a plain in-stream device copy in its place 0.97.  A searching stream's chain
is a sequence of latency-bound launches, and everything c10d puts between two of them shows.

The library is the librccl.so torch itself loads (torch/lib), so both communicators live in one RCCL instance."""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_INT32 = 2


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path if os.path.exists(path) else "librccl.so")
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib.ncclGetErrorString(rc).decode()} ({rc})")


class Communicator:
    """One RCCL communicator over the ranks of the default process group (which must be initialised: it carries the id)."""

    def __init__(self, device):
        lib = _load()
        self.lib = lib
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib, lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]   # (the raw 128 bytes: the id is not a C string)
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.addressof(uid), box[0], 128)
        self.comm = C.c_void_p()
        torch.cuda.set_device(device)
        _check(lib, lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_gather_i32(self, dst, src, stream):
        """dst [world * n] <- every rank's src [n] (int32 device tensors), enqueued on `stream` (a torch.cuda.Stream); returns at once."""
        _check(self.lib, self.lib.ncclAllGather(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(src.numel()), NCCL_INT32,
                                                self.comm, C.c_void_p(stream.cuda_stream)), "ncclAllGather")

    def bind_all_gather_i32(self, dst, src, stream):
        """The same call with its arguments converted once (a step's gather is then one foreign call)."""
        fn, args = self.lib.ncclAllGather, (C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(src.numel()), NCCL_INT32,
                                            self.comm, C.c_void_p(stream.cuda_stream))
        lib = self.lib

        def call():
            rc = fn(*args)
            if rc != 0:
                _check(lib, rc, "ncclAllGather")
        return call

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
