"""Direct RCCL all-reduce on the compute stream (one process per GPU, communicator over xGMI).

Why not torch.distributed for the data path: a training step issues ~660 tiny SyncBN all-reduces that sit on the critical
path.  Through ProcessGroupNCCL each costs ~27 us of host time (work objects, events, stream hand-offs) and, because its
watchdog thread polls those events, the step cannot be captured into a hipGraph.  Calling `ncclAllReduce` ourselves on
torch's current stream has neither problem: the collective is just another node of the captured step.

torch.distributed is still used for the rendezvous (it carries the ncclUniqueId from rank 0) and as the fallback when
librccl cannot be loaded.  The library is the one PyTorch ships (torch/lib/librccl.so), so both communicators share the
same RCCL build.
"""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32, NCCL_SUM = 7, 0
_lib = None


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path)
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
            f.restype = ctypes.c_int
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _load().ncclGetErrorString(rc).decode()))


class Communicator:
    """One RCCL communicator spanning the ranks of the default torch.distributed group."""

    def __init__(self):
        lib = _load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        box = [bytes(uid.internal) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)                 # the 128-byte id travels over the existing rendezvous
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        _check(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._lib = lib

    def all_reduce_(self, t):
        """In-place fp32 sum over all ranks, enqueued on torch's current stream (graph-capturable)."""
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("rccl.all_reduce_: contiguous fp32 device tensor expected")
        st = torch._C._cuda_getCurrentRawStream(t.device.index)
        p = t.data_ptr()
        _check(self._lib.ncclAllReduce(p, p, t.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, st), "ncclAllReduce")
        return t

    def destroy(self):
        if self.comm:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


_COMM = None


def get():
    return _COMM


def init():
    """Create the communicator (all ranks must call this together).  Returns None - and leaves the torch.distributed
    data path in place - if RCCL cannot be used directly (RSSF_DP_BACKEND=torch, no GPU, load failure)."""
    global _COMM
    if _COMM is not None:
        return _COMM
    if (os.environ.get("RSSF_DP_BACKEND", "rccl") != "rccl" or not torch.cuda.is_available() or not dist.is_initialized()
            or dist.get_backend() != "nccl"):
        return None
    ok = torch.ones(1, device="cuda")
    try:
        _load()
    except OSError:
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                  # every rank takes the same branch
    if float(ok) < 1:
        return None
    comm = None
    try:
        comm = Communicator()
    except Exception as e:                                     # noqa: BLE001 - any failure means "use torch.distributed"
        print("[rssf] direct RCCL communicator unavailable (%s: %s); using torch.distributed collectives" % (type(e).__name__, e),
              flush=True)
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                  # all ranks or none
    if float(ok) < 1:
        if comm is not None:
            comm.destroy()
        return None
    _COMM = comm
    return _COMM


def shutdown():
    global _COMM
    if _COMM is not None:
        _COMM.destroy()
        _COMM = None
