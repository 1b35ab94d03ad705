"""Communicators of the data-parallel step (one process per GPU, RCCL over xGMI) on top of librssf's `rssf_comm_*` entry
points (include/rssf.h, csrc/comm.cpp), plus the torch.distributed stand-in used when RCCL cannot be driven directly.

Why not torch.distributed for the data path: a training step issues ~660 tiny SyncBN all-reduces that sit on the critical
path.  Through ProcessGroupNCCL each costs ~27 us of host time (work objects, events, stream hand-offs) and, because its
watchdog thread polls those events, the step cannot be captured into a hipGraph.  `rssf_syncbn_exchange` /
`rssf_allreduce_bucket` enqueue `ncclAllReduce` on the stream they are given: the collective is just another node of the
captured step.  torch.distributed is still used for the rendezvous (it carries the 128-byte id from rank 0).  librssf binds
the RCCL build PyTorch ships (torch/lib/librccl.so) at run time, so both share one RCCL runtime.

A communicator object offers `world`, `rank`, `syncbn_exchange_(stats)` and `allreduce_bucket_(flat_slice)`; both enqueue on
torch's CURRENT stream.  Two RCCL communicators are created per trainer: one for the SyncBN exchanges on the compute stream
and one for the gradient buckets, which run on a side stream overlapped with the rest of backward (independent
communicators may be in flight at the same time; one communicator must not be)."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib as L


_SKIP_1RANK = os.environ.get("RSSF_SKIP_1RANK_COLLECTIVES") == "1"      # debugging aid: a 1-rank all-reduce is the identity


def rccl_library_path():
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


class Communicator:
    """One RCCL communicator spanning the ranks of the default torch.distributed group (direct, graph-capturable)."""

    direct = True

    def __init__(self):
        lib = L.load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        path = rccl_library_path().encode()
        uid = ctypes.create_string_buffer(128)
        err = None
        if self.rank == 0 and lib.rssf_comm_unique_id(uid, path) != 0:
            err = lib.rssf_last_error().decode()
        box = [uid.raw if (self.rank == 0 and err is None) else None]
        dist.broadcast_object_list(box, src=0)                 # the 128-byte id travels over the existing rendezvous
        if box[0] is None:                                     # rank 0 could not reach RCCL: every rank raises together
            raise RuntimeError("rssf_comm_unique_id failed on rank 0" + (": " + err if err else ""))
        self._h = ctypes.c_void_p()
        L.check(lib.rssf_comm_init(ctypes.byref(self._h), self.rank, self.world, ctypes.create_string_buffer(box[0], 128), path),
                "rssf_comm_init")
        self._lib = lib
        self.n_syncbn = 0           # exchanges issued so far (reported by bench.py / checked by the DP tests)

    @staticmethod
    def _chk(t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("rccl: contiguous fp32 device tensor expected")

    def syncbn_exchange_(self, stats):
        """In-place sum over ranks of a BatchNorm statistics buffer, on torch's current stream."""
        self._chk(stats)
        self.n_syncbn += 1
        if _SKIP_1RANK and self.world == 1:
            return stats
        L.check(self._lib.rssf_syncbn_exchange(L.ptr(stats), stats.numel(), self._h, L.stream()), "rssf_syncbn_exchange")
        return stats

    def allreduce_bucket_(self, t):
        """In-place sum over ranks of a flat fp32 gradient bucket, on torch's current stream."""
        self._chk(t)
        if _SKIP_1RANK and self.world == 1:
            return t
        L.check(self._lib.rssf_allreduce_bucket(L.ptr(t), t.numel(), L.RSSF_F32, self._h, L.stream()), "rssf_allreduce_bucket")
        return t

    def destroy(self):
        if self._h:
            self._lib.rssf_comm_destroy(self._h)
            self._h = ctypes.c_void_p()


class TorchComm:
    """The same interface through torch.distributed (any backend: gloo on CPU hosts / for the 2-ranks-on-one-GPU parity
    test, ProcessGroupNCCL when RSSF_DP_BACKEND=torch).  Eager launches only."""

    direct = False

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_syncbn = 0

    def syncbn_exchange_(self, stats):
        self.n_syncbn += 1
        dist.all_reduce(stats, group=self.group)
        return stats

    def allreduce_bucket_(self, t):
        dist.all_reduce(t, group=self.group)
        return t

    def allreduce_bucket_async(self, t):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def destroy(self):
        pass


def _agree(flag):
    """Every rank takes the same branch: MIN over ranks of a local success flag."""
    t = torch.tensor([1.0 if flag else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t) >= 1


def create(n=1):
    """`n` communicators over the default group (all ranks call this together): direct RCCL ones when possible, else None
    (RSSF_DP_BACKEND=torch, non-NCCL backend, no GPU, load failure - decided collectively)."""
    if (os.environ.get("RSSF_DP_BACKEND", "rccl") != "rccl" or not torch.cuda.is_available() or not dist.is_initialized()
            or dist.get_backend() != "nccl"):
        return None
    comms, ok = [], True
    try:
        for _ in range(n):
            comms.append(Communicator())
    except Exception as e:                                     # noqa: BLE001 - any failure means "use torch.distributed"
        print("[rssf] direct RCCL communicator unavailable (%s: %s); using torch.distributed collectives" % (type(e).__name__, e),
              flush=True)
        ok = False
    if not _agree(ok):                                         # all ranks or none
        for c in comms:
            c.destroy()
        return None
    return comms
