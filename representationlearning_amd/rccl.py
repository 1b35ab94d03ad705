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

    def nranks(self):
        """The rank count RCCL itself reports for this communicator (ncclCommCount)."""
        return int(self._lib.rssf_comm_nranks(self._h))

    @staticmethod
    def _chk(t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("rccl: contiguous fp32 device tensor expected")

    def syncbn_exchange_(self, stats, layout=None):
        """In-place sum over ranks of a BatchNorm statistics buffer, on torch's current stream (`layout`: see P2PChannel; an
        all-reduce sums the slots of all ranks element-wise, which the consumers' slot fold turns into the same totals)."""
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


class P2PExchange:
    """The peer-to-peer SyncBN exchange of librssf (`rssf_p2p_*`, csrc/p2p.hip): one window per rank, mapped by every peer through
    hipIpc; `channel(k)` hands out the communicator-like object of exchange sequence k (one per stream)."""

    def __init__(self, channels, rank=None, world=None):
        """rank / world: given explicitly, the object is one of `world` ranks of THIS process (see `local_group`) and no rendezvous
        takes place; otherwise rank and world are torch.distributed's and the window handles travel over its rendezvous."""
        lib = L.load()
        local = rank is not None
        self.rank, self.world, self.channels = (rank, world, channels) if local else (dist.get_rank(), dist.get_world_size(), channels)
        self._lib, self._h = lib, ctypes.c_void_p()
        mine = ctypes.create_string_buffer(64)
        err = None
        if lib.rssf_p2p_create(ctypes.byref(self._h), self.rank, self.world, channels, mine) != 0:
            err = lib.rssf_last_error().decode()
        if local:
            self.error = err
            return
        handles = [None] * self.world
        dist.all_gather_object(handles, None if err else mine.raw)              # every rank joins this, failed or not
        if err is None and any(h is None for h in handles):
            err = "a peer could not create its window"
        if err is None:
            for r, hb in enumerate(handles):
                if r != self.rank and lib.rssf_p2p_connect(self._h, r, ctypes.create_string_buffer(hb, 64)) != 0:
                    err = lib.rssf_last_error().decode()
                    break
        self.error = err

    @classmethod
    def local_group(cls, world, channels):
        """`world` ranks in this process, connected window to window (rssf_p2p_connect_local): one rank per stream - the form the
        exchange kernel is tested in at the node's world size without another process competing for the GPU's hardware queues."""
        group = [cls(channels, rank=r, world=world) for r in range(world)]
        for a in group:
            if a.error:
                raise RuntimeError("P2PExchange.local_group: %s" % a.error)
            for b in group:
                if a is not b:
                    L.check(a._lib.rssf_p2p_connect_local(a._h, b.rank, b._h), "rssf_p2p_connect_local")
        return group

    def channel(self, k):
        return P2PChannel(self, k)

    def set_timeout_ms(self, ms):
        L.check(self._lib.rssf_p2p_set_timeout_ms(self._h, int(ms)), "rssf_p2p_set_timeout_ms")

    def timed_out(self):
        v = ctypes.c_int(0)
        L.check(self._lib.rssf_p2p_status(self._h, ctypes.byref(v)), "rssf_p2p_status")
        return v.value

    def wait_us(self, channel, reset=False):
        """(microseconds the exchanges of `channel` waited for their peers, exchanges) since creation / the last reset - counted on the
        device by the exchange kernel (rssf_p2p_wait_us; synchronises)."""
        us, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        L.check(self._lib.rssf_p2p_wait_us(self._h, int(channel), ctypes.byref(us), ctypes.byref(n), int(bool(reset))), "rssf_p2p_wait_us")
        return us.value, n.value

    def destroy(self):
        if self._h:
            self._lib.rssf_p2p_destroy(self._h)
            self._h = ctypes.c_void_p()


class P2PChannel:
    """Communicator interface (`world`, `rank`, `syncbn_exchange_`) over one channel of a P2PExchange."""

    direct = True

    def __init__(self, ex, k):
        self.ex, self.k, self.rank, self.world = ex, k, ex.rank, ex.world
        self.n_syncbn = 0

    def syncbn_exchange_(self, stats, layout=None):
        """In-place: slot 0 of every layer's block <- sum over slots and ranks, the other slots <- 0 (the consumers fold all slots).
        layout = (nslots, [(element offset, 2C), ...]); without one the buffer is taken as ONE unslotted item."""
        Communicator._chk(stats)
        self.n_syncbn += 1
        nslots, items = layout if layout is not None else (1, [(0, stats.numel())])
        lib = self.ex._lib
        # one kernel per run of items: at most P2P_MAX_ITEMS layers AND at most P2P_MAX_FLOATS values (rssf_p2p_exchange rejects
        # more: a wider backbone or a longer lock-step group would otherwise raise in the middle of a step - ADVICE r3); a single
        # layer above the window is named in the error
        part, floats = [], 0
        runs = []
        for o, n in items:
            if n > L.P2P_MAX_FLOATS:
                raise RuntimeError("P2PChannel: a layer of %d statistics values exceeds the exchange window (%d floats); "
                                   "use RSSF_SYNCBN=rccl for this model" % (n, L.P2P_MAX_FLOATS))
            if part and (len(part) == L.P2P_MAX_ITEMS or floats + n > L.P2P_MAX_FLOATS):
                runs.append(part)
                part, floats = [], 0
            part.append((o, n))
            floats += n
        if part:
            runs.append(part)
        for part in runs:
            offs = (ctypes.c_int * len(part))(*[o for o, _ in part])
            ns = (ctypes.c_int * len(part))(*[n for _, n in part])
            L.check(lib.rssf_p2p_exchange(self.ex._h, self.k, L.ptr(stats), offs, ns, len(part), nslots, L.stream()), "rssf_p2p_exchange")
        return stats

    def destroy(self):
        pass


def create_p2p(channels, reference_comm):
    """A validated P2PExchange with `channels` channels, or None (all ranks decide together).  The self-test runs one exchange per
    channel on the hardware it is going to be used on - every rank contributes rank-dependent values in a slotted two-layer layout -
    and holds the result against an all-reduce of the same numbers through `reference_comm` (RCCL); a time-out, a mismatch or any
    setup error on any rank sends every rank back to the RCCL exchanges."""
    if os.environ.get("RSSF_SYNCBN", "p2p") != "p2p":
        return None
    ex, why = None, None
    try:
        ex = P2PExchange(channels)
        why = ex.error
    except Exception as e:                                     # noqa: BLE001
        why = "%s: %s" % (type(e).__name__, e)
    if _agree(why is None):
        try:
            ex.set_timeout_ms(int(os.environ.get("RSSF_P2P_TIMEOUT_MS", "3000")))      # the self-test must not hang on a broken fabric
            rank, world = ex.rank, ex.world
            nslots, c1, c2 = 4, 24, 40
            base = torch.arange(nslots * (c1 + c2), device="cuda", dtype=torch.float32) * 0.25 + (rank + 1) * 3.0
            layout = (nslots, [(0, c1), (nslots * c1, c2)])
            want = torch.cat([base[:nslots * c1].view(nslots, c1).sum(0), base[nslots * c1:].view(nslots, c2).sum(0)]).contiguous()
            reference_comm.allreduce_bucket_(want)
            for k in range(channels):
                for _ in range(3):                              # both window parities, and the epoch carried between launches
                    got = base.clone()
                    ex.channel(k).syncbn_exchange_(got, layout)
                    tot = torch.cat([got[:c1], got[nslots * c1:nslots * c1 + c2]])
                    rest = torch.cat([got[c1:nslots * c1], got[nslots * c1 + c2:]])
                    if not (torch.equal(tot, want) and float(rest.abs().sum()) == 0.0):
                        why = why or "self-test mismatch on channel %d" % k
            if ex.timed_out():
                why = "self-test timed out waiting for a peer"
        except Exception as e:                                 # noqa: BLE001
            why = "%s: %s" % (type(e).__name__, e)
        if _agree(why is None):
            # steady state: wait for a late rank as long as it takes (it may be evaluating or writing a checkpoint), unless the
            # environment asks for a bound (tests)
            ex.set_timeout_ms(int(os.environ.get("RSSF_P2P_TIMEOUT_MS", "0")))
            return ex
    if why:
        print("[rssf] peer-to-peer SyncBN exchange unavailable (%s); using RCCL all-reduces" % why, flush=True)
    if ex is not None:
        ex.destroy()
    return None


class TorchComm:
    """The same interface through torch.distributed (any backend: gloo on CPU hosts / for the 2-ranks-on-one-GPU parity
    test, ProcessGroupNCCL when RSSF_DP_BACKEND=torch).  Eager launches only."""

    direct = False

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_syncbn = 0

    def syncbn_exchange_(self, stats, layout=None):
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
    (RSSF_DP_BACKEND=torch, non-NCCL backend, no GPU, load failure - decided collectively).
    Every step that can fail on ONE rank is followed by an agreement (MIN over ranks of a success flag) before the next collective
    step starts, so that a failing rank never leaves the others inside a broadcast or an ncclCommInitRank it does not join:
    library load first, then each communicator after its construction."""
    if (os.environ.get("RSSF_DP_BACKEND", "rccl") != "rccl" or not torch.cuda.is_available() or not dist.is_initialized()
            or dist.get_backend() != "nccl"):
        return None
    why = None
    try:
        L.load()
        if not os.path.exists(rccl_library_path()):
            why = "no RCCL library at %s" % rccl_library_path()
    except Exception as e:                                     # noqa: BLE001
        why = "%s: %s" % (type(e).__name__, e)
    if not _agree(why is None):
        if why:
            print("[rssf] direct RCCL communicator unavailable (%s); using torch.distributed collectives" % why, flush=True)
        return None
    comms = []
    for _ in range(n):
        c, err = None, None
        try:
            c = Communicator()          # rank 0's id failure reaches every rank through the broadcast: all raise together
        except Exception as e:                                 # noqa: BLE001 - any failure means "use torch.distributed"
            err = "%s: %s" % (type(e).__name__, e)
        if c is not None:
            comms.append(c)
        if not _agree(err is None):                            # all ranks or none, before the next communicator is built
            if err:
                print("[rssf] direct RCCL communicator unavailable (%s); using torch.distributed collectives" % err, flush=True)
            for k in comms:
                k.destroy()
            return None
    return comms
