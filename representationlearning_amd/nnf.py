"""Fused functional layers of the RSSFormer path on top of the librssf C ABI:

    conv_bn_act : Conv2d (one conv, or the MLP's three summed convs as ONE 19-tap launch) -> BatchNorm2d ->
                  {none, ReLU, GELU} with an optional residual before and/or after the activation
    conv_bias   : Conv2d with bias and nothing else (the segmentation head)

Activations are logical-NCHW tensors in channels_last memory (physically NHWC); parameters stay in the nn.Conv2d /
nn.BatchNorm2d modules of the mirror model (same state_dict as the reference).  Each op is ONE autograd node whose
forward/backward are short sequences of C-ABI launches (no arithmetic in Python).  SyncBN = an all-reduce of the
[2][C] statistics between two launches.
"""
import ctypes
import os
import threading

import torch
import torch.nn as nn

from . import _lib as L

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ACT_POST_RELU = 16                        # RSSF_ACT_POST_RELU: y = relu(act(z) + res_post)
BN_SLOTS, BN_BWD_SLOTS = 16, 8            # RSSF_BN_SLOTS / RSSF_BN_BWD_SLOTS of include/rssf.h


# ---- per-step scratch: one zero-fill and one weight re-pack per training step ----------------------------------------
# A training step needs ~700 small zeroed fp32 buffers (BatchNorm statistics, bias-gradient sums) and re-packs the
# weights of ~330 convolutions twice (forward and data-gradient layouts).  Done per call that is ~1400 tiny launches
# (4.7 ms of GPU time and as much host time per step on MI355X).  The trainer brackets a step with step_begin():
#   * ZeroPool: one flat fp32 buffer, zeroed by ONE memset, handed out in 16-byte-aligned slices (sized from the
#     previous step's demand; a request that does not fit falls back to torch.zeros),
#   * PackPlan: every (ConvSpec, layout) packed during the first step is recorded; from then on ONE
#     rssf_conv_pack_batch launch per step refreshes all of them (parameters live in the trainer's flat buffer, so
#     their addresses are stable) and _pack() returns views of the plan's buffer.
class ZeroPool:
    def __init__(self):
        self.buf, self.off, self.need, self.last_need, self.active = None, 0, 0, 0, False

    def begin(self, device):
        if self.last_need and (self.buf is None or self.buf.numel() < self.last_need or self.buf.device != device):
            self.buf = torch.zeros(self.last_need, device=device, dtype=torch.float32)
        elif self.buf is not None and self.off:
            from . import ops
            ops.zero_(self.buf[:self.off])                # only what the previous step handed out (a kernel, never a memset node)
        self.off, self.need, self.active = 0, 0, True

    def end(self):
        self.last_need = max(self.last_need, self.need)
        self.active = False

    def zeros(self, n, device):
        n4 = (n + 3) // 4 * 4
        self.need += n4
        if not self.active or self.buf is None or self.off + n4 > self.buf.numel() or self.buf.device != device:
            return torch.zeros(n, device=device, dtype=torch.float32)
        out = self.buf[self.off:self.off + n]
        self.off += n4
        return out


class Runtime:
    """Everything a step of the path needs besides the model: the data-parallel communicator and SyncBN policy, where
    parameter gradients go, the per-step scratch pools, stream policy.  One per Trainer (two trainers in one process do not
    share state); `current()` is the active one of the calling thread, the module-level default otherwise.  Autograd nodes
    capture the runtime in forward and use THAT in backward (backward runs on the autograd engine's own thread)."""

    def __init__(self):
        self.comm = None                 # rccl.Communicator / rccl.TorchComm used for the SyncBN exchanges
        self.sync_all_bn = False         # configs/base/loveda.py:107 train.sync_bn (MlpDWBN's nn.SyncBatchNorm always sync)
        self.force_collectives = False   # issue the exchanges even on a 1-rank group (plumbing tests)
        # direct gradient accumulation: with the trainer's flat gradient buffer every parameter already owns a zeroed fp32
        # `.grad` view, and every librssf backward kernel ACCUMULATES (+=) its parameter gradients.  So the kernels write
        # straight into `.grad` (no per-tensor zero-fill, no AccumulateGrad add: ~2 k tiny launches per step) and the autograd
        # node returns None for them; the trainer's bucket hook is invoked by hand instead.
        self.direct = False
        self.param_ready = None
        # HRNet branches of a HighResolutionModule on side streams (parallel branches of a captured hipGraph); off in
        # data-parallel runs, where the branches' SyncBN exchanges must reach the communicator in one fixed order
        self.branch_streams = False
        self.side_streams = {}
        # data parallel: the items of a lock-step group (nnf.conv_bn_act_group) on parallel streams around the group's ONE SyncBN
        # exchange, which stays on the step's stream - the branch concurrency of `branch_streams` with a fixed collective order
        self.group_streams = False
        self.in_side = 0                 # > 0 while a side-stream function runs (no nested forks: DESIGN lesson 16)
        # data parallel: one SyncBN communicator PER SIDE STREAM (stream_comms[k] belongs to side_streams[dev][k]); None = the side
        # streams may not issue exchanges (single communicator: branches and fuse paths then stay on the step's stream, in lock-step)
        self.stream_comms = None
        self.comm_active = None          # the communicator exchanges issued NOW go through (set by parallel_map / fork_side)
        self.zero_pool = ZeroPool()
        self.pack_plan = None
        self.wgrad_plan = None
        # fixed-order reductions for the BatchNorm statistics (bit-identical runs): RSSF_DETERMINISTIC=1 or set by the caller
        self.deterministic = os.environ.get("RSSF_DETERMINISTIC", "0") == "1"
        # transposed zero-bordered activation copies ([C][B][H + 2 pad][W + 2 pad]) for rssf_conv_wgrad_planes: one per producing
        # layer, allocated and zeroed ONCE (the kernels write the interior only, the border stays zero), so a captured step
        # replays against fixed addresses.  planes_gen counts the writes: a backward pass that finds a newer copy than the one
        # its forward pass wrote (the layer ran twice before its backward) takes the ordinary weight-gradient path
        self.planes = {}
        self.planes_gen = {}

    def planes_buffer(self, key, shape, dtype, device):
        k = (key, tuple(shape), dtype, str(device))
        buf = self.planes.get(k)
        if buf is None:
            # born in the eager warm-up steps (Trainer runs three before it captures): inside a capture the zero-fill would become a
            # node of the graph - a 94 MB fill replayed every step - and the buffer would live in the capture's private pool
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("planes_buffer: first use of %r inside a stream capture (run an eager step of this shape first)" % (k[1],))
            # Memory: C * B * (H + 24) * (W + 24) bf16 per MlpDWBN block and input shape (94 MB at 16 x 128 x 128 x 128), kept for the
            # life of the Runtime - a captured step replays against these addresses, so a set is never freed behind a graph's back;
            # a run that trains on many crop shapes pays one set per shape (RSSF_WGRAD_PLANES=0 turns the path off)
            buf = self.planes[k] = torch.zeros(shape, device=device, dtype=dtype)
        self.planes_gen[k] = gen = self.planes_gen.get(k, 0) + 1
        return buf, k, gen

    def planes_current(self, k, gen):
        return self.planes_gen.get(k) == gen

    @property
    def world(self):
        return self.comm.world if self.comm is not None else 1

    def exchanging(self):
        return self.comm is not None and (self.comm.world > 1 or self.force_collectives)

    def exchange_comm(self):
        """Communicator of the SyncBN exchanges issued at this point of the forward pass: every side stream has its own
        (collectives of one communicator must be issued in ONE order on every rank; two streams give no such order)."""
        return self.comm_active if self.comm_active is not None else self.comm


_DEFAULT = Runtime()
_TLS = threading.local()


def current():
    return getattr(_TLS, "rt", None) or _DEFAULT


class use:
    """`with nnf.use(rt):` - make rt the calling thread's runtime."""

    def __init__(self, rt):
        self.rt = rt

    def __enter__(self):
        self.prev = getattr(_TLS, "rt", None)
        _TLS.rt = self.rt
        return self.rt

    def __exit__(self, *exc):
        _TLS.rt = self.prev
        return False


def set_sync_bn(flag, force=False):
    rt = current()
    rt.sync_all_bn, rt.force_collectives = bool(flag), bool(force)


def set_direct_grad(flag, on_ready=None):
    rt = current()
    rt.direct, rt.param_ready = bool(flag), on_ready


def set_branch_streams(flag):
    current().branch_streams = bool(flag)


def grad_target(p, rt=None):
    """(buffer the kernels accumulate into, True if that buffer IS p.grad)."""
    rt = rt or current()
    if rt.direct and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous():
        return p.grad, True
    return torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format), False


def grad_result(p, buf, direct, rt=None):
    """What the autograd node returns for parameter p."""
    if direct:
        rt = rt or current()
        if rt.param_ready is not None:
            rt.param_ready(p)
        return None
    return buf


def parallel_map(fns, args):
    """[f(a) for f, a in zip(fns, args)], items 1.. on side streams when branch streams are enabled (item 0 stays on the
    current stream); joined before returning.  A small (32x32 or 16x16) map fills a fraction of the 256 CUs and its chain of
    kernels is latency-bound; autograd replays the backward of every node on its forward stream, and a captured hipGraph keeps
    the fork/join as parallel graph branches."""
    n = len(fns)
    rt = current()
    dp = rt.exchanging()
    if not (rt.branch_streams and n > 1 and args[0].is_cuda) or rt.in_side or (dp and (rt.stream_comms is None or len(rt.stream_comms) < n - 1)):
        if dp and rt.branch_streams and n > 1 and not rt.in_side and not getattr(rt, "_warned_serial_branches", False):
            # a module whose branches cannot walk in lock-step, under data parallelism, with fewer side communicators than branches:
            # correct, but the step's layout is not the single-GPU one any more - say so once
            rt._warned_serial_branches = True
            print("[rssf] %d parallel branches run one after the other: %d side communicator(s) for SyncBN exchanges (RSSF_LOCKSTEP=0 "
                  "gives every branch stream its own)" % (n, 0 if rt.stream_comms is None else len(rt.stream_comms)), flush=True)
        return [f(a) for f, a in zip(fns, args)]
    dev = args[0].device
    cur = torch.cuda.current_stream(dev)
    pool = rt.side_streams.setdefault(dev, [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(dev))
    outs = [None] * n
    for i in range(1, n):
        s = pool[i - 1]
        s.wait_stream(cur)
        args[i].record_stream(s)
        prev = rt.comm_active
        rt.in_side += 1
        rt.comm_active = rt.stream_comms[i - 1] if dp else None      # the SyncBN exchanges of this branch: its stream's communicator
        try:
            with torch.cuda.stream(s):
                outs[i] = fns[i](args[i])
        finally:
            rt.in_side -= 1
            rt.comm_active = prev
    outs[0] = fns[0](args[0])
    for i in range(1, n):
        cur.wait_stream(pool[i - 1])
        outs[i].record_stream(cur)
    return outs


def can_fork_side(t):
    """True when fork_side() would really run its side function on a side stream (branch streams on, not already inside a side
    function, and - under data parallelism - a communicator of its own for that stream)."""
    rt = current()
    return bool(rt.branch_streams and _FORK_FUSE and t is not None and t.is_cuda and not rt.in_side and not (rt.exchanging() and not rt.stream_comms))


def side_stream(t):
    """The stream fork_side() would run its side function on, or None when it would not fork (see can_fork_side)."""
    if not can_fork_side(t):
        return None
    pool = current().side_streams.setdefault(t.device, [])
    if not pool:
        pool.append(torch.cuda.Stream(t.device))
    return pool[0]


def fork_side(fn_side, fn_main, tensors):
    """(fn_side(), fn_main()): fn_side on a side stream when branch streams are enabled, concurrently with fn_main on the current
    stream; joined before returning.  `tensors`: what fn_side reads (they were produced on the current stream).
    (Measured and dropped, round 4: a data-driven form - no fork-time wait for inputs the side stream produced itself, the join of
    the fuse fork deferred to the first reader on another stream - leaves the step where it is, 29.84 against 29.80-29.87 ms:
    the streams share one GPU's CUs and fabric, the extra freedom buys nothing.)"""
    rt = current()
    ts = [t for t in tensors if t is not None]
    if not (rt.branch_streams and _FORK_FUSE and ts and ts[0].is_cuda) or (rt.exchanging() and not rt.stream_comms) or rt.in_side:
        return fn_side(), fn_main()
    dev = ts[0].device
    cur = torch.cuda.current_stream(dev)
    pool = rt.side_streams.setdefault(dev, [])
    if not pool:
        pool.append(torch.cuda.Stream(dev))
    s = pool[0]
    s.wait_stream(cur)
    for t in ts:
        t.record_stream(s)
    prev = rt.comm_active
    rt.in_side += 1
    rt.comm_active = rt.stream_comms[0] if rt.exchanging() else None       # pool[0]'s communicator
    try:
        with torch.cuda.stream(s):
            side = fn_side()
    finally:
        rt.in_side -= 1
        rt.comm_active = prev
    main = fn_main()
    cur.wait_stream(s)
    for t in (side if isinstance(side, (list, tuple)) else [side]):
        if t is not None:
            t.record_stream(cur)
    return side, main


class _Phases:
    """Runs the per-item launches of a lock-step group on parallel streams (item 0 on the current one) - `run(i, fn, inputs)`
    between `join()`s.  Disabled (everything on the current stream) unless the runtime asks for it, inside fork_side's side
    function (no nested forks) and for groups whose items share a gradient accumulator."""

    def __init__(self, rt, n, device, enabled=True, forked=None):
        self.streams = None
        on = (rt.group_streams and enabled and n > 1 and device.type == "cuda" and not rt.in_side) if forked is None else forked
        if on:
            self.cur = torch.cuda.current_stream(device)
            pool = rt.side_streams.setdefault(device, [])
            while len(pool) < n:
                pool.append(torch.cuda.Stream(device))
            self.streams = [self.cur] + pool[1:n]          # pool[0] is fork_side's stream (busy with the fuse outputs 1..)
            self.used = set()
            self.made = []

    @property
    def forked(self):
        return self.streams is not None

    def run(self, i, fn, inputs=()):
        if self.streams is None or i == 0:
            return fn()
        s = self.streams[i]
        if i not in self.used:
            s.wait_stream(self.cur)
            self.used.add(i)
        for t in inputs:
            if t is not None:
                t.record_stream(s)
        with torch.cuda.stream(s):
            out = fn()
        for t in (out if isinstance(out, (list, tuple)) else [out]):
            if torch.is_tensor(t):
                self.made.append(t)
        return out

    def join(self):
        if self.streams is None:
            return
        for i in sorted(self.used):
            self.cur.wait_stream(self.streams[i])
        for t in self.made:                    # allocated on a side stream, used (and possibly freed) on the current one from here on
            t.record_stream(self.cur)
        self.used, self.made = set(), []


def _ia(vals):
    return (ctypes.c_int * len(vals))(*vals)


class ConvSpec:
    """Tap list of a convolution (or of a sum of convolutions sharing input/output, stride 1)."""

    def __init__(self, convs, tap_range=None):
        c0 = convs[0]
        self.cin, self.cout = c0.in_channels, c0.out_channels
        self.stride = c0.stride[0]
        self.ksizes, self.src, self.kpos, self.dy, self.dx = [], [], [], [], []
        self.alias = []                                  # per tap: further (src, kpos) x 2 sampling the same pixel, -1 = none
        slot = {}
        for s, c in enumerate(convs):
            k, d, p = c.kernel_size[0], c.dilation[0], c.padding[0]
            if (c.in_channels, c.out_channels, c.stride[0]) != (self.cin, self.cout, self.stride) or c.groups != 1 \
                    or c.kernel_size[0] != c.kernel_size[1] or c.padding[0] != c.padding[1] or c.dilation[0] != c.dilation[1]:
                raise NotImplementedError("librssf conv: square, ungrouped convolutions sharing in/out/stride only")
            self.ksizes.append(k)
            for ky in range(k):
                for kx in range(k):
                    off = (ky * d - p, kx * d - p)
                    u = slot.get(off)
                    if u is not None and -1 in self.alias[u]:          # same source pixel as tap u: one tap, summed weights
                        e = self.alias[u].index(-1)
                        self.alias[u][e:e + 2] = [s, ky * k + kx]
                        continue
                    slot.setdefault(off, len(self.src))
                    self.src.append(s)
                    self.kpos.append(ky * k + kx)
                    self.dy.append(off[0])
                    self.dx.append(off[1])
                    self.alias.append([-1, -1, -1, -1])
        # more than RSSF_MAX_TAPS positions (the 7 x 7 stem of the ResNet-50 CAM path: 49): the convolution runs as a chain of
        # <= 19-tap launches, each adding its taps' contribution to the previous partial sum (rssf_conv_gather_add's addend)
        self.parts = None
        if tap_range is None and len(self.src) > 19:
            if len(convs) != 1:
                raise NotImplementedError("librssf conv: tap splitting needs a single convolution")
            n = len(self.src)
            self.parts = [ConvSpec(convs, (a, min(a + 19, n))) for a in range(0, n, 19)]
        if tap_range is not None:
            a_, b_ = tap_range
            self.src, self.kpos, self.dy, self.dx, self.alias = (self.src[a_:b_], self.kpos[a_:b_], self.dy[a_:b_], self.dx[a_:b_],
                                                                 self.alias[a_:b_])
        self.ntaps = len(self.src)
        self.c_alias = _ia([v for a4 in self.alias for v in a4])
        # ctypes views of the tap tables, built once (they are passed to every launch of this convolution)
        self.c_ksizes, self.c_src, self.c_kpos = _ia(self.ksizes), _ia(self.src), _ia(self.kpos)
        self.c_dy, self.c_dx = _ia(self.dy), _ia(self.dx)
        self.c_ndy, self.c_ndx = _ia([-v for v in self.dy]), _ia([-v for v in self.dx])
        if (self.ntaps > 19 and self.parts is None) or len(convs) > 3:
            raise NotImplementedError("librssf conv: at most 19 taps per launch / 3 fused convolutions")
        c = c0
        self._out_hw = lambda h, w: ((h + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // self.stride + 1,
                                     (w + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // self.stride + 1)

    def out_hw(self, h, w):
        return self._out_hw(h, w)


def spec_of(convs):
    """ConvSpec of a (tuple of) nn.Conv2d, cached ON the first module (a global dict keyed by id() would hand a stale
    spec to a new module that happens to reuse the address of a freed one)."""
    c0 = convs[0]
    key = tuple(id(c) for c in convs[1:])
    cache = c0.__dict__.setdefault("_rssf_specs", {})
    sp = cache.get(key)
    if sp is None:
        sp = cache[key] = ConvSpec(convs)
    return sp


def _nhwc(x):
    """logical NCHW -> [B,H,W,C] view; copies only if x is not channels-last."""
    t = x.permute(0, 2, 3, 1)
    return t if t.is_contiguous() else t.contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2)


class PackPlan:
    def __init__(self):
        self.jobs, self.views, self.fresh, self.built = {}, {}, False, False
        self.recording = False

    def record(self, key, spec, weights, transpose, dtype):
        if (self.recording and key not in self.jobs and all(w.is_contiguous() and w.dtype == torch.float32 for w in weights)
                and 32 * sum(k * k for k in spec.ksizes[:len(weights)]) <= 2560):      # (rssf_conv_pack_job_blocks: what the batched form holds)
            self.jobs[key] = (spec, list(weights), transpose, dtype)

    def build(self):
        """Allocate the packed buffers and the device-side job table for everything recorded so far."""
        lib = L.load()
        if not self.jobs:
            return
        self.by_dtype = {}
        for dtype in {j[3] for j in self.jobs.values()}:
            code = L.RSSF_BF16 if dtype == torch.bfloat16 else L.RSSF_F32
            keys = [k for k, j in self.jobs.items() if j[3] == dtype]
            dev = self.jobs[keys[0]][1][0].device
            sizes = []
            for k in keys:
                spec, ws, tr, _ = self.jobs[k]
                rows, cols = (spec.cin, spec.cout) if tr else (spec.cout, spec.cin)
                sizes.append(lib.rssf_conv_packed_elems(spec.ntaps, rows, cols, code))
            offs, tot = [], 0
            for n in sizes:
                offs.append(tot)
                tot += (n + 7) // 8 * 8
            flat = torch.empty(tot, device=dev, dtype=dtype)
            arr = (L.PackJob * len(keys))()
            bmap = []
            for i, k in enumerate(keys):
                spec, ws, tr, _ = self.jobs[k]
                rows, cols = (spec.cin, spec.cout) if tr else (spec.cout, spec.cin)
                j = arr[i]
                for s_, w in enumerate(ws):
                    j.w[s_] = w.data_ptr()
                    j.ks[s_] = spec.ksizes[s_]
                for s_ in range(len(ws), 3):
                    j.ks[s_] = 1
                j.out = flat.data_ptr() + offs[i] * flat.element_size()
                j.nsrc, j.ntaps, j.cout, j.cin, j.transpose = len(ws), spec.ntaps, spec.cout, spec.cin, int(tr)
                j.rows_p, j.cols_p = lib.rssf_conv_packed_rows(rows), lib.rssf_conv_packed_cols(cols, code)
                for t in range(spec.ntaps):
                    j.src_of_tap[t], j.kpos_of_tap[t] = spec.src[t], spec.kpos[t]
                    for e in range(4):
                        j.alias_of_tap[t][e] = spec.alias[t][e]
                self.views[k] = flat[offs[i]:offs[i] + sizes[i]]
                bmap += [(i, c) for c in range(lib.rssf_conv_pack_job_blocks(j.rows_p, j.cols_p, int(tr), sum(k * k for k in spec.ksizes[:len(ws)])))]
            jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            bmap_dev = torch.tensor(bmap, dtype=torch.int32).to(dev).contiguous()
            ptrs = [w.data_ptr() for k in keys for w in self.jobs[k][1]]
            self.by_dtype[dtype] = (code, flat, jobs_dev, bmap_dev, len(bmap), keys, ptrs)
        self.built = True

    def refresh(self):
        """Re-pack everything from the current parameter values (one launch per dtype)."""
        lib = L.load()
        for dtype, (code, flat, jobs_dev, bmap_dev, nb, keys, ptrs) in self.by_dtype.items():
            if ptrs != [w.data_ptr() for k in keys for w in self.jobs[k][1]]:
                raise RuntimeError("PackPlan: a recorded convolution weight moved in memory (parameters must stay in the flat buffer)")
            L.check(lib.rssf_conv_pack_batch(L.ptr(jobs_dev), L.ptr(bmap_dev), nb, code, L.stream()), "rssf_conv_pack_batch")
        self.fresh = True

    def lookup(self, key):
        return self.views.get(key) if (self.built and self.fresh) else None


class WgradPlan:
    """Deferred second stage of the split-K weight gradients: the ~330 `wgrad_reduce` launches of a step (5.6 us each, a
    latency-bound walk over ~9 MB of partials) become a few `rssf_conv_wgrad_reduce_batch` launches.
    Step 1 records the sequence of weight-gradient calls and their workspace sizes; then one arena holds every layer's partials
    at a fixed address (the captured hipGraph needs that anyway), and from step 2 on `_conv_wgrad` launches the first stage
    only.  `flush()` reduces what has been deferred since the previous flush - ONE launch at the end of backward without data
    parallelism, one per gradient bucket with it (the trainer flushes before a bucket's all-reduce starts, so that bucket's
    gradients are final).  Each flush is a SEGMENT [first job, last job) of the step's call sequence with its own device-side job
    table, built the first time that segment runs and reused while the step keeps its shape."""

    def __init__(self):
        self.sizes, self.keys = [], []          # recorded call sequence
        self.recording, self.built = False, False
        self.cursor = 0                         # next call of the running step
        self.flushed = 0                        # first job not yet reduced
        self.seg_i = 0                          # next segment of the running step
        self.jobs_host, self.segments = [], []
        self.seen_dw = set()

    def begin(self):
        self.cursor = self.flushed = self.seg_i = 0
        self.seen_dw = set()

    @property
    def pending(self):
        return self.flushed < self.cursor

    def duplicate_target(self, ptrs):
        """True when one of the gradient buffers `ptrs` is already the target of a deferred job of this step: the batched reduction
        adds into `dw` without atomics, so two jobs of one launch must not share a target (a weight used twice in a step keeps the
        immediate, stream-ordered reduction).  Deterministic per step shape: the call sequence stays consistent."""
        dup = any(q in self.seen_dw for q in ptrs)
        self.seen_dw.update(ptrs)
        return dup

    def record(self, key, elems):
        self.keys.append(key)
        self.sizes.append(int(elems))

    def build(self, device):
        offs, tot = [], 0
        for n in self.sizes:
            offs.append(tot)
            tot += (n + 63) // 64 * 64
        self.arena = torch.empty(max(tot, 1), device=device, dtype=torch.float32)
        self.offs = offs
        self.built = True

    def slot(self, key):
        """Workspace slice + job struct for the next weight-gradient call of the step, or None if the call sequence changed."""
        i = self.cursor
        if not self.built or i >= len(self.keys) or self.keys[i] != key:
            self.built = False                       # a different step shape: immediate reductions from here on, for good
            return None
        self.cursor += 1
        ws = self.arena[self.offs[i]:self.offs[i] + self.sizes[i]]
        if len(self.jobs_host) <= i:
            self.jobs_host.append(L.WgradReduceJob())
            return ws, self.jobs_host[i], None
        return ws, L.WgradReduceJob(), self.jobs_host[i]

    def _table(self, a, b):
        lib = L.load()
        arr = (L.WgradReduceJob * (b - a))(*self.jobs_host[a:b])
        bmap = []
        for j in range(b - a):
            bmap += [(j, blk) for blk in range(lib.rssf_conv_wgrad_reduce_blocks(ctypes.byref(arr[j])))]
        dev = self.arena.device
        jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        bmap_dev = torch.tensor(bmap, dtype=torch.int32).reshape(-1, 2).to(dev).contiguous()
        return (a, b, jobs_dev, bmap_dev, len(bmap))

    def flush(self):
        """Reduce every job deferred since the last flush: jobs [flushed, cursor) of the running step, one launch."""
        a, b = self.flushed, self.cursor
        if a >= b:
            return
        seg = self.segments[self.seg_i] if self.seg_i < len(self.segments) else None
        if seg is None or seg[0] != a or seg[1] != b:
            seg = self._table(a, b)                  # first run of this segment (or the flush points moved: rebuilt, not reused)
            if self.seg_i < len(self.segments):
                self.segments[self.seg_i] = seg
            else:
                self.segments.append(seg)
        self.seg_i += 1
        self.flushed = b
        if seg[4]:
            L.check(L.load().rssf_conv_wgrad_reduce_batch(L.ptr(seg[2]), L.ptr(seg[3]), seg[4], L.stream()), "rssf_conv_wgrad_reduce_batch")

    def end(self):
        """End of the step: reduce the rest; a step that issued fewer calls than recorded has changed shape."""
        self.flush()
        if self.built and self.cursor != len(self.keys):
            self.built = False


def step_begin(device, plan=None, rt=None, wgrad_plan=None):
    """Trainer hook: start of a training step (zero pool reset + batched weight packing)."""
    rt = rt or current()
    rt.zero_pool.begin(device)
    rt.pack_plan = plan
    rt.wgrad_plan = wgrad_plan
    if wgrad_plan is not None:
        wgrad_plan.begin()
        if not wgrad_plan.built and not wgrad_plan.keys:
            wgrad_plan.recording = True
    if plan is not None:
        if plan.built:
            plan.refresh()
        else:
            plan.recording = True


def step_end(rt=None):
    """Trainer hook: end of a training step (the parameters are about to change / have changed)."""
    rt = rt or current()
    rt.zero_pool.end()
    if rt.pack_plan is not None:
        plan = rt.pack_plan
        if plan.recording and not plan.built:
            plan.recording = False
            plan.build()
        plan.fresh = False
    rt.pack_plan = None
    wp = rt.wgrad_plan
    if wp is not None:
        wp.end()
        if wp.recording:
            wp.recording = False
            if wp.keys:
                wp.build(rt.zero_pool.buf.device if rt.zero_pool.buf is not None else torch.device("cuda"))
    rt.wgrad_plan = None


def _zeros(n, device, rt=None):
    return (rt or current()).zero_pool.zeros(n, device)


def _pack(spec, weights, transpose, dtype, device, rt=None, cache=False):
    """cache: keep the packed copy ON the spec until a weight's version counter or address changes.  Only for paths whose
    parameters are never written behind autograd's back (the inference-only CAM networks): the trainer's fused SGD kernel updates
    the flat parameter buffer through raw pointers, which no version counter sees - training / evaluation go through PackPlan."""
    key = (id(spec), bool(transpose), dtype)
    plan = (rt or current()).pack_plan
    if plan is not None:
        v = plan.lookup(key)
        if v is not None:
            return v
        plan.record(key, spec, weights, transpose, dtype)
    elif cache:
        tag = tuple((w.data_ptr(), w._version) for w in weights)
        hit = spec.__dict__.setdefault("_packed", {}).get((bool(transpose), dtype, device))
        if hit is not None and hit[0] == tag:
            return hit[1]
    lib = L.load()
    code = L.RSSF_BF16 if dtype == torch.bfloat16 else L.RSSF_F32
    rows, cols = (spec.cin, spec.cout) if transpose else (spec.cout, spec.cin)
    n = lib.rssf_conv_packed_elems(spec.ntaps, rows, cols, code)
    out = torch.empty(n, device=device, dtype=dtype)
    w = [wt if wt.is_contiguous() else wt.contiguous() for wt in weights] + [None, None]
    L.check(lib.rssf_conv_pack(L.ptr(w[0]), L.ptr(w[1]), L.ptr(w[2]), _ia(spec.ksizes), len(weights), _ia(spec.src), _ia(spec.kpos),
                               spec.c_alias, spec.ntaps, spec.cout, spec.cin, int(transpose), L.ptr(out), code, L.stream()), "rssf_conv_pack")
    if cache and plan is None and not torch.cuda.is_current_stream_capturing():      # (a buffer born inside a capture belongs to the graph)
        spec._packed[(bool(transpose), dtype, device)] = (tag, out)
    return out


def _pad_channels(t):
    """[B,H,W,C] -> channels zero-padded to the 16-byte vector width (C=3 stem input, C=6 head gradient, 18-channel Small):
    the kernels then take their vector paths; the packed weight slabs are zero there anyway."""
    v = 8 if t.dtype == torch.bfloat16 else 4
    c = t.shape[-1]
    if c % v == 0:
        return t
    if t.is_cuda and t.dtype in (torch.bfloat16, torch.float32) and t.is_contiguous():
        cp = (c + v - 1) // v * v
        out = torch.empty(*t.shape[:-1], cp, device=t.device, dtype=t.dtype)
        L.check(L.load().rssf_pad_channels(L.ptr(t), L.ptr(out), t.numel() // c, c, cp, L.dtype_code(t), L.stream()), "rssf_pad_channels")
        return out
    return torch.nn.functional.pad(t, (0, v - c % v))


def _conv_forward(spec, xh, weights, bias, stats, rt=None, addend=None, preact=None, cache_pack=False, generic=False):
    """generic: run the generic gather / halo kernels only (rssf.h RSSF_CONV_GENERIC) - what the parity tests hold the
    shape-specialised kernels (conv_pw.hip, conv_taps128.hip) against."""
    if spec.parts is not None:                      # > 19 taps: partial sums chained through the epilogue's addend (inference)
        if stats is not None:
            raise NotImplementedError("librssf conv: fused statistics are not available for tap-split convolutions")
        out = None
        for k, part in enumerate(spec.parts):
            out = _conv_forward(part, xh, weights, bias if k == 0 else None, None, rt, addend=out, cache_pack=cache_pack, generic=generic)
        return out
    xh = _pad_channels(xh)
    B, H, W, C = xh.shape
    OH, OW = spec.out_hw(H, W)
    wpk = _pack(spec, weights, False, xh.dtype, xh.device, rt, cache=cache_pack)
    out = torch.empty(B, OH, OW, spec.cout, device=xh.device, dtype=xh.dtype)
    lib = L.load()
    ws = None
    if stats is not None and (rt or current()).deterministic:       # fixed-order statistics: per-tile partials + ordered fold
        ws = torch.empty(lib.rssf_conv_stats_workspace_elems(B, OH, OW, spec.cout), device=xh.device, dtype=torch.float32)
    if preact is not None:          # xh is the RAW output of the producing convolution: its BatchNorm + activation are applied on load
        if addend is not None:
            raise RuntimeError("conv forward: a pre-activation input cannot be combined with an addend")
        (pst, pga, pbe, prm, prv, pmi, pss, pn, pmom, peps, ptr), pact = preact
        L.check(lib.rssf_conv_gather_preact(L.ptr(xh), L.ptr(pst), L.ptr(pga), L.ptr(pbe), L.ptr(prm), L.ptr(prv), L.ptr(pmi), L.ptr(pss), pn, pmom,
                                            peps, int(ptr), pact, L.ptr(wpk), L.ptr(out), L.ptr(bias), L.ptr(stats), L.ptr(ws), B, H, W, C, OH, OW,
                                            spec.cout, spec.stride, 1, spec.ntaps, spec.c_dy, spec.c_dx, L.dtype_code(xh) | (L.CONV_GENERIC if generic else 0),
                                            L.stream()), "rssf_conv_gather_preact")
        return out
    L.check(lib.rssf_conv_gather_add(L.ptr(xh), L.ptr(wpk), L.ptr(out), L.ptr(bias), L.ptr(stats), L.ptr(addend), L.ptr(ws), B, H, W, C, OH, OW,
                                     spec.cout, spec.stride, 1, spec.ntaps, spec.c_dy, spec.c_dx, L.dtype_code(xh) | (L.CONV_GENERIC if generic else 0),
                                     L.stream()), "rssf_conv_gather")
    return out


_DEFER_CACHE = {}


def can_defer_apply(x, conv_a, conv_b):
    """True when `act(bn(conv_a(x)))` - consumed by conv_b and nothing else - need not be materialised: conv_b's forward and
    weight-gradient kernels can apply the producer's BatchNorm + activation to the raw convolution output while they stage it
    (rssf_conv_gather_preact / rssf_conv_wgrad_bnapply(in_scale_shift) / rssf_conv_wgrad_preact; bf16; 3x3 / stride 1 with channels
    multiples of 8, or MlpDWBN's fc2 behind its tap sum - conv_a may be the list of summed convolutions)."""
    if not (_DEFER_BN_APPLY and _FUSED_BN_APPLY and _FUSED_BN_STATS and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16):
        return False
    if current().deterministic:
        return False
    sa, sb = spec_of(list(conv_a) if isinstance(conv_a, (list, tuple)) else [conv_a]), spec_of([conv_b])
    B, _, H, W = x.shape
    OH, OW = sa.out_hw(H, W)
    key = (id(sb), B, OH, OW)
    if key not in _DEFER_CACHE:
        lib = L.load()
        OH2, OW2 = sb.out_hw(OH, OW)
        code = L.RSSF_BF16
        ok = sb.parts is None and sa.parts is None and sb.cin % 8 == 0 and sb.cout % 8 == 0
        ok = ok and lib.rssf_conv_gather_preact_supported(B, OH, OW, sb.cin, OH2, OW2, sb.cout, sb.stride, 1, sb.ntaps, sb.c_dy, sb.c_dx, code) == 1
        # 1: pre-activation operand + fused apply in one launch (bias-free 3x3); 2: pre-activation operand only (MlpDWBN's fc2, biased)
        ok = ok and lib.rssf_conv_wgrad_preact_supported(B, OH, OW, sb.cin, OH2, OW2, sb.cout, sb.stride, sb.ntaps, 1, sb.c_dy, sb.c_dx,
                                                         int(conv_b.bias is not None), code) in (1, 2)
        _DEFER_CACHE[key] = bool(ok)
    return _DEFER_CACHE[key]


class GradLink:
    """Fuses the skip-path gradient of a residual block into the data-gradient launch of the block's first convolution.

    `y = act(bn(conv_b(act(bn(conv_a(x))))) + x)`: autograd would add the two gradients of x (through conv_a, and the skip)
    with a separate elementwise kernel - 108 of them per step.  The node that owns the skip (`deposit=link`) hands its
    residual gradient to the link and returns None for it; the node of conv_a (`sink=link`), which by construction runs
    later in backward, adds it in the epilogue of its data-gradient convolution (`rssf_conv_gather_add`)."""

    def __init__(self):
        self.value = None


class BnBwdLink:
    """BatchNorm-backward statistics of layer L produced by the data-gradient launch of its ONLY consumer, layer L + 1
    (`rssf_conv_gather_bnbwd`): conv_bn_act(.., stats_out=link) fills in what the statistics need (raw, scale/shift, the residual
    added before the activation, the activation) and, in backward, takes `link.sums` instead of launching rssf_bn_bwd_reduce;
    conv_bn_act(y, .., stats_in=link) - y being that layer's output, consumed by nothing else but a residual skip whose gradient
    rides on the same launch (GradLink) - produces them.  The caller vouches for the single consumer."""

    __slots__ = ("raw", "ss", "rp", "act", "C", "sums", "deferred", "pre", "want_planes", "planes")

    def __init__(self):
        self.raw = self.ss = self.rp = self.sums = None
        # want_planes: the consumer's weight gradient takes its input TRANSPOSED (rssf_conv_wgrad_planes) - the producer's BatchNorm
        # apply then writes that copy beside its output (rssf_bn_finalize_apply_planes) and leaves (buffer, pad) in `planes`
        self.want_planes, self.planes = False, None
        self.act, self.C = ACT_NONE, 0
        self.deferred = False      # the producer returned its RAW convolution output: the consumer applies BatchNorm + activation on load
        self.pre = None            # (stats, gamma, beta, running mean / var, mi, ss, n, momentum, eps, training) of a deferred producer


class GroupBwdLink:
    """The BnBwdLinks of the items of one _ConvBNActGroup plus ONE contiguous sums buffer for them, so that the producer group still
    needs a single SyncBN exchange.  Item i of the consumer group must consume the output of item i of the producer group."""

    __slots__ = ("items", "sums_all", "offs", "filled")

    def __init__(self, n):
        self.items = [BnBwdLink() for _ in range(n)]
        self.sums_all, self.offs, self.filled = None, None, 0

    def slot(self, i, device, rt):
        if self.sums_all is None:
            sizes = [BN_BWD_SLOTS * 2 * it.C for it in self.items]
            self.offs = [sum(sizes[:k]) for k in range(len(sizes) + 1)]
            self.sums_all = _zeros(self.offs[-1], device, rt)
        self.filled += 1
        return self.sums_all[self.offs[i]:self.offs[i + 1]]


def group_stats_link(n):
    return GroupBwdLink(n) if (_FUSED_BN_STATS and torch.is_grad_enabled() and not current().deterministic) else None


_GROUP_LAUNCH = os.environ.get("RSSF_GROUP_LAUNCH", "1") != "0"      # A/B switch: the phases of a lock-step group as grouped launches
_PLAN_BIAS = os.environ.get("RSSF_WGRAD_PLAN_BIAS", "1") != "0"      # A/B switch: deferred split-K reduction also for convolutions with a bias
_FORK_FUSE = os.environ.get("RSSF_FORK_FUSE", "1") != "0"      # A/B switch: fuse outputs 1.. beside the transformer block (fork_side)
_DEFER_BN_APPLY = os.environ.get("RSSF_DEFER_BN_APPLY", "1") != "0"      # A/B switch: forward BatchNorm apply inside the consumer's staging
_WGRAD_PLANES = os.environ.get("RSSF_WGRAD_PLANES", "1") != "0"          # A/B switch: transposed-input weight gradient of the MLP's tap sum
PLANES_PAD = 12                      # zero border of the transposed copy: the largest tap offset of the MLP's dilated convolutions
_FUSED_PW_DGRAD = os.environ.get("RSSF_FUSED_PW_DGRAD", "1") != "0"      # A/B switch: a point-wise layer's data gradient inside its weight-gradient launch
_FUSED_BN_APPLY = os.environ.get("RSSF_FUSED_BN_APPLY", "1") != "0"      # A/B switch: BatchNorm-backward apply inside the weight-gradient launch
_FUSED_BN_STATS = os.environ.get("RSSF_FUSED_BN_STATS", "1") != "0"      # A/B switch (tools)


def wgrad_planes_ok(x, C, convs):
    """True when the weight gradient of `convs` (summed, stride 1, C -> C channels) on an input of x's batch / size takes the
    transposed-planes operand (rssf_conv_wgrad_planes_supported with the pad the apply pass writes)."""
    if not (_WGRAD_PLANES and x.is_cuda and x.dtype == torch.bfloat16):
        return False
    sp = spec_of(list(convs))
    B, _, H, W = x.shape
    return bool(sp.parts is None and sp.stride == 1 and sp.cin == C and
                L.load().rssf_conv_wgrad_planes_supported(B, H, W, C, sp.cout, 1, sp.ntaps, sp.c_dy, sp.c_dx, PLANES_PAD, L.RSSF_BF16) == 1)


def bwd_stats_link():
    """A BnBwdLink when gradients are being recorded (and the run is not in deterministic-statistics mode), else None."""
    return BnBwdLink() if (_FUSED_BN_STATS and torch.is_grad_enabled() and not current().deterministic) else None


class GradAccum:
    """Gradient of a tensor that feeds SEVERAL convolutions (a branch output of a HighResolutionModule feeds every fuse path;
    layer1's output feeds both transition convolutions).  Autograd would sum the per-consumer gradients with one elementwise
    kernel each (87 bf16 adds per step, up to 134 MB apiece).  Here the tensor passes through `fanout(x, acc)` once, every
    convolution consumer is given `grad_accum=acc` and returns None for its input gradient: the first to run in backward writes its
    data gradient into `acc.buf`, the others accumulate IN PLACE through the data-gradient epilogue's addend (out == addend; each
    element is read and written by the same thread).  The fan-out node - which autograd runs after all consumers, in whatever
    order those came - returns the buffer, plus the ordinary gradients of consumers that are not convolutions.  All consumers
    must run on ONE stream (the fuse layers / transitions do)."""

    __slots__ = ("buf", "stream")

    def __init__(self):
        self.buf = None
        self.stream = None          # stream the consumers accumulate on (the fan-out node may run on another one)


def _dense(t):
    """every element of the storage range exactly once (any dimension order): an element-wise kernel may walk it linearly"""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) or t.permute(
        *sorted(range(t.dim()), key=lambda d: -t.stride(d))).is_contiguous()


def _add_into(t, g):
    """t += g in place with the library's own launch (same dtype, both dense in the same memory order), else torch's."""
    if (t.is_cuda and g.dtype == t.dtype and t.dtype in (torch.float32, torch.bfloat16) and g.shape == t.shape and g.stride() == t.stride()
            and _dense(t) and t.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0):
        L.check(L.load().rssf_add(L.ptr(t), L.ptr(g), L.ptr(t), t.numel(), L.dtype_code(t), L.stream()), "rssf_add")
        return t
    return t.add_(g.to(t.dtype))


class _Add(torch.autograd.Function):
    """a + b of two activations of one shape / dtype / memory order (the running sums of a HighResolutionModule's fuse outputs,
    _hrnet_rssformer.py:424-435); the gradient goes to both unchanged."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        L.check(L.load().rssf_add(L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.dtype_code(a), L.stream()), "rssf_add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    """a + b; the library's launch where both operands are dense GPU tensors of one shape, dtype and memory order."""
    if (a.is_cuda and a.dtype == b.dtype and a.dtype in (torch.float32, torch.bfloat16) and a.shape == b.shape and a.stride() == b.stride()
            and _dense(a) and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        return _Add.apply(a, b)
    return a + b


def _sum_into(t, gs):
    """t += sum(gs) in place (t: a buffer this node owns); two addends meet in ONE launch (rssf_add3)."""
    gs = [g for g in gs if g is not None]
    ok = lambda g: (g.dtype == t.dtype and g.shape == t.shape and g.stride() == t.stride() and g.data_ptr() % 16 == 0)
    if (len(gs) == 2 and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and _dense(t) and t.data_ptr() % 16 == 0 and ok(gs[0]) and ok(gs[1])):
        L.check(L.load().rssf_add3(L.ptr(t), L.ptr(gs[0]), L.ptr(gs[1]), L.ptr(t), t.numel(), L.dtype_code(t), L.stream()), "rssf_add3")
        return t
    for g in gs:
        t = _add_into(t, g)
    return t


class _Fanout(torch.autograd.Function):
    """x -> n aliases of x (n >= 1).  Backward: the accumulated convolution gradients (GradAccum) plus the gradients of the aliases,
    summed by the library's own launches - handing ONE tensor to several autograd consumers instead leaves the sum to the
    engine's accumulation (an element-wise framework launch per extra consumer)."""

    @staticmethod
    def forward(ctx, x, acc, n):
        ctx.acc = acc
        ctx.set_materialize_grads(False)
        if n == 1:
            return x.view_as(x)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        acc = ctx.acc
        buf = None
        if acc is not None:
            buf, acc.buf = acc.buf, None
        gs = [g for g in gs if g is not None]
        if buf is None:
            if len(gs) <= 1:
                return (gs[0] if gs else None), None, None
            first = gs[0]
            if len(gs) == 2 and first.dtype == gs[1].dtype and first.shape == gs[1].shape and first.stride() == gs[1].stride():
                return add(first, gs[1]), None, None
            t = first.clone()
            return _sum_into(t, gs[1:]), None, None
        if acc.stream is not None:      # the consumers ran on a side stream (fork_side): order this node after them
            cur = torch.cuda.current_stream(buf.device)
            if acc.stream != cur:
                cur.wait_stream(acc.stream)
                buf.record_stream(cur)
        t = _nchw(buf)
        return (t if not gs else _sum_into(t, gs)), None, None


def fanout(x, n_conv_consumers, n_alias=1):
    """(x', GradAccum) for a tensor with convolution consumers that take `grad_accum=`; (x, None) when there are none.  Also with a
    SINGLE such consumer: when that consumer runs on a side stream (fork_side), its gradient must not meet the other consumers'
    in autograd's own cross-stream accumulation - under hipGraph capture that ended in a crash of the capture (ROCm 7.0) - but in
    the fan-out node, which orders itself after the side stream explicitly.
    n_alias > 1: x' is a tuple of n_alias aliases, one per further autograd consumer - their gradients meet the accumulated ones
    inside the node (rssf_add3 / rssf_add) instead of in the engine's accumulation."""
    if n_alias > 1 and os.environ.get("RSSF_FANOUT_ALIAS", "1") == "0":       # A/B switch: the consumers share x, autograd sums their gradients
        t, acc = fanout(x, n_conv_consumers)
        return tuple([t] * n_alias), acc
    if not (torch.is_grad_enabled() and x.requires_grad and x.is_cuda) or (n_conv_consumers < 1 and n_alias < 2):
        return (x if n_alias == 1 else tuple([x] * n_alias)), None
    acc = GradAccum() if n_conv_consumers >= 1 else None
    return _Fanout.apply(x, acc, n_alias), acc


def _accumulate_dgrad(accum, spec, dout, weights, in_shape, rt):
    st = torch.cuda.current_stream(dout.device)
    if accum.stream is not None and accum.buf is not None and accum.stream != st:
        raise RuntimeError("GradAccum: the consumers of one accumulator must run on ONE stream")
    accum.stream = st
    if accum.buf is None:
        accum.buf = _conv_dgrad(spec, dout, weights, in_shape, None, rt)
    else:
        _conv_dgrad(spec, dout, weights, in_shape, accum.buf, rt, out=accum.buf)


def _conv_dgrad(spec, dout, weights, in_shape, addend=None, rt=None, out=None, bn=None, generic=False):
    """bn: (BnBwdLink, zeroed sums buffer) - the launch also accumulates the BatchNorm-backward statistics of the layer that
    produced the convolution's input (see BnBwdLink)."""
    B, H, W, C = in_shape
    dout = _pad_channels(dout)
    _, OH, OW, cout_p = dout.shape
    wpk = _pack(spec, weights, True, dout.dtype, dout.device, rt)
    dx = torch.empty(B, H, W, C, device=dout.device, dtype=dout.dtype) if out is None else out
    if addend is not None and (addend.shape != dx.shape or addend.dtype != dx.dtype or not addend.is_contiguous()):
        raise RuntimeError("conv dgrad: fused skip gradient has shape/dtype %s %s, expected %s %s"
                           % (tuple(addend.shape), addend.dtype, tuple(dx.shape), dx.dtype))
    if bn is not None:
        link, sums = bn
        if link.raw.shape != dx.shape or link.raw.dtype != dx.dtype or link.C != C:
            raise RuntimeError("BnBwdLink: the producer's output %s %s is not this convolution's input %s %s"
                               % (tuple(link.raw.shape), link.raw.dtype, tuple(dx.shape), dx.dtype))
        L.check(L.load().rssf_conv_gather_bnbwd(L.ptr(dout), L.ptr(wpk), L.ptr(dx), L.ptr(addend), L.ptr(link.raw), L.ptr(link.rp), L.ptr(link.ss),
                                                link.act, L.ptr(sums), B, OH, OW, cout_p, H, W, C, 1, spec.stride, spec.ntaps, spec.c_ndy,
                                                spec.c_ndx, L.dtype_code(dout) | (L.CONV_GENERIC if generic else 0), L.stream()), "rssf_conv_gather_bnbwd")
        return dx
    L.check(L.load().rssf_conv_gather_add(L.ptr(dout), L.ptr(wpk), L.ptr(dx), None, None, L.ptr(addend), None, B, OH, OW, cout_p, H, W, C, 1, spec.stride,
                                      spec.ntaps, spec.c_ndy, spec.c_ndx, L.dtype_code(dout) | (L.CONV_GENERIC if generic else 0),
                                      L.stream()), "rssf_conv_gather(dgrad)")
    return dx


def _conv_wgrad(spec, dout, xh, dws, db, rt=None, bn=None, xpre=None, planes=None, generic=False, dgrad=None, no_draw=False):
    """Accumulates (+=) into the fp32 buffers dws (one per source conv) and db (optional).  Under a WgradPlan the split-K
    reduction is deferred to the plan's one batched launch (the gradients are complete after WgradPlan.flush()).
    bn: (dy, raw, ss, mi, sums, res_pre, dres, dgamma, dbeta, act, n, training, pscale) - `dout` is then an OUTPUT: the launch
    performs the layer's BatchNorm-backward apply on the way (rssf_conv_wgrad_bnapply) and leaves draw in `dout` (dz in dres)."""
    xh, dout = _pad_channels(xh), _pad_channels(dout)
    B, H, W, C = xh.shape
    _, OH, OW, CO = dout.shape
    padded = C != spec.cin or CO != spec.cout
    tgt, tdb = list(dws), db
    if padded:            # gradients of the padded problem (slices of the step's pre-zeroed pool), added back row by row afterwards
        rt0 = rt or current()
        tgt = [_zeros(CO * C * k * k, xh.device, rt0).view(CO, C, k, k) for k in spec.ksizes]
        tdb = _zeros(CO, xh.device, rt0) if db is not None else None
    d = tgt + [None, None]
    lib = L.load()
    nws = lib.rssf_conv_wgrad_workspace_elems(B, OH, OW, C, CO, spec.ntaps)
    plan = (rt or current()).wgrad_plan
    ws = job = ref = None
    # (a bias gradient does not stand in the way: the first stage adds it straight into its buffer, only the weight partials wait)
    if plan is not None and not padded and (tdb is None or _PLAN_BIAS) and not plan.duplicate_target([t.data_ptr() for t in dws]):
        key = (id(spec), B, H, W, C, OH, OW, CO, xh.dtype, tuple(t.data_ptr() for t in dws))
        if plan.recording:
            plan.record(key, nws)
        else:
            got = plan.slot(key)
            if got is not None:
                ws, job, ref = got
    if ws is None:
        ws = torch.empty(nws, device=xh.device, dtype=torch.float32)
    tail = (L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), spec.c_ksizes, len(dws), spec.c_src, spec.c_kpos, spec.c_alias, L.ptr(tdb),
            L.ptr(ws), B, H, W, C, OH, OW, CO, spec.stride, spec.ntaps, spec.c_dy, spec.c_dx, None if job is None else ctypes.byref(job),
            L.dtype_code(xh) | (L.CONV_GENERIC if generic else 0) | (L.WGRAD_NO_DRAW if no_draw else 0), L.stream())
    use_planes = (planes is not None and (rt or current()).planes_current(planes[2], planes[3]) and not padded and xpre is None and
                  spec.stride == 1 and OH == H and OW == W and tuple(planes[0].shape) == (C, B, H + 2 * planes[1], W + 2 * planes[1]) and
                  lib.rssf_conv_wgrad_planes_supported(B, H, W, C, CO, 1, spec.ntaps, spec.c_dy, spec.c_dx, planes[1], L.dtype_code(xh)) == 1)
    if use_planes:
        # the input operand comes TRANSPOSED (written by the producer's BatchNorm apply): csrc/conv_wgrad_planes.hip.  A fused
        # BatchNorm-backward apply is what the library itself falls back to for this shape: the apply pass, then the weight gradient
        if bn is not None:
            bdy, braw, bss, bmi, bsums, brp, bdres, bdg, bdb, bact, bn_n, btr, bps = bn
            L.check(lib.rssf_bn_bwd_apply(L.ptr(bdy), L.ptr(braw), L.ptr(bss), L.ptr(bmi), L.ptr(bsums), L.ptr(brp), L.ptr(dout), L.ptr(bdres),
                                          L.ptr(bdg), L.ptr(bdb), braw.numel() // CO, CO, bact, bn_n, int(btr), bps, L.dtype_code(braw), L.stream()),
                    "rssf_bn_bwd_apply")
        L.check(lib.rssf_conv_wgrad_planes(L.ptr(dout), L.ptr(planes[0]), planes[1], L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), spec.c_ksizes, len(dws),
                                           spec.c_src, spec.c_kpos, spec.c_alias, L.ptr(tdb), L.ptr(ws), B, H, W, C, spec.ntaps, spec.c_dy, spec.c_dx,
                                           None if job is None else ctypes.byref(job), L.dtype_code(xh), L.stream()), "rssf_conv_wgrad_planes")
    elif dgrad is not None and len(dgrad) == 3:
        # a pre-activation input (xpre) and no apply of its own in the call: dgrad = (fp32 weights, dx to fill, the producer's statistics)
        L.check(lib.rssf_conv_wgrad_preact_dgrad(L.ptr(dout), L.ptr(xh), L.ptr(xpre[0]), xpre[1], L.ptr(dgrad[0]), L.ptr(dgrad[1]), L.ptr(dgrad[2]),
                                                 L.ptr(d[0]), L.ptr(tdb), L.ptr(ws), B, H, W, C, CO, None if job is None else ctypes.byref(job),
                                                 L.dtype_code(xh), L.stream()), "rssf_conv_wgrad_preact_dgrad")
    elif dgrad is not None:
        # the layer's whole backward in one launch (rssf_conv_wgrad_bnapply_dgrad): dgrad = (fp32 weights, dx to fill)
        bdy, braw, bss, bmi, bsums, brp, bdres, bdg, bdb, bact, bn_n, btr, bps = bn
        L.check(lib.rssf_conv_wgrad_bnapply_dgrad(L.ptr(bdy), L.ptr(braw), L.ptr(bss), L.ptr(bmi), L.ptr(bsums), L.ptr(dout), L.ptr(bdg), L.ptr(bdb),
                                                  bact, bn_n, int(btr), bps, L.ptr(xh), L.ptr(dgrad[0]), L.ptr(dgrad[1]), L.ptr(d[0]), L.ptr(tdb),
                                                  L.ptr(ws), B, H, W, C, CO, None if job is None else ctypes.byref(job), L.dtype_code(xh), L.stream()),
                "rssf_conv_wgrad_bnapply_dgrad")
    elif bn is not None:
        if CO != spec.cout:
            raise RuntimeError("conv_wgrad: the fused BatchNorm-backward apply needs an output channel count the kernels take unpadded")
        bdy, braw, bss, bmi, bsums, brp, bdres, bdg, bdb, bact, bn_n, btr, bps = bn
        xss, xact = xpre if xpre is not None else (None, 0)
        L.check(lib.rssf_conv_wgrad_bnapply(L.ptr(bdy), L.ptr(braw), L.ptr(bss), L.ptr(bmi), L.ptr(bsums), L.ptr(brp), L.ptr(dout), L.ptr(bdres),
                                            L.ptr(bdg), L.ptr(bdb), bact, bn_n, int(btr), bps, L.ptr(xh), L.ptr(xss), xact, *tail),
                "rssf_conv_wgrad_bnapply")
    else:
        if xpre is not None:           # xh is the producer's RAW output; this layer's own apply was a separate pass (e.g. the _post form)
            L.check(lib.rssf_conv_wgrad_preact(L.ptr(dout), L.ptr(xh), L.ptr(xpre[0]), xpre[1], *tail), "rssf_conv_wgrad_preact")
        else:
            L.check(lib.rssf_conv_wgrad(L.ptr(dout), L.ptr(xh), *tail), "rssf_conv_wgrad")
    if job is not None:
        if ref is not None and bytes(job) != bytes(ref):
            raise RuntimeError("WgradPlan: a deferred weight-gradient reduction changed between steps")
    if padded:
        for g, t, k in zip(dws, tgt, spec.ksizes):
            if g.is_contiguous() and g.dtype == torch.float32:
                # rows = output channels; [C, k, k] is channel-major, so the unpadded input channels are the first cin * k * k values of a row
                L.check(lib.rssf_add_rows(L.ptr(g), L.ptr(t), spec.cout, spec.cin * k * k, spec.cin * k * k, C * k * k, L.stream()), "rssf_add_rows")
            else:
                g += t[:spec.cout, :spec.cin]
        if db is not None:
            if db.is_contiguous() and db.dtype == torch.float32:
                L.check(lib.rssf_add_rows(L.ptr(db), L.ptr(tdb), 1, spec.cout, spec.cout, CO, L.stream()), "rssf_add_rows")
            else:
                db += tdb[:spec.cout]


class _ConvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res_pre, res_post, gamma, beta, rmean, rvar, spec, act, training, momentum, eps, sync, nbias, links, *wb):
        weights, biases = wb[:len(wb) - nbias], wb[len(wb) - nbias:]
        L.require_gpu(x)
        xh = _nhwc(x)
        dev = x.device
        bias = None
        if nbias == 1:
            bias = biases[0].float().contiguous()
        elif nbias:                                    # summed convs: their biases add up to ONE epilogue bias (one launch)
            bs = [b if (b.dtype == torch.float32 and b.is_contiguous()) else b.float().contiguous() for b in biases]
            bias = torch.empty_like(bs[0])
            L.check(L.load().rssf_vec_sum3(L.ptr(bs[0]), L.ptr(bs[1]), L.ptr(bs[2]) if nbias > 2 else None, L.ptr(bias), bias.numel(),
                                           L.stream()), "rssf_vec_sum3")
        C = spec.cout
        rt = current()
        stats = _zeros(BN_SLOTS * 2 * C, dev, rt) if training else None
        si_ = links[4] if len(links) > 4 else None
        preact = (si_.pre, si_.act) if (si_ is not None and si_.deferred) else None     # x is the producer's RAW output (BnBwdLink)
        raw = _conv_forward(spec, xh, weights, bias, stats, rt, preact=preact)
        rows = raw.numel() // C
        n = float(rows)
        exchanged = training and sync and rt.exchanging()
        ctx.comm = rt.exchange_comm() if exchanged else None
        if exchanged:                  # SyncBN: pooled {sum, sumsq} and sample count over the data-parallel ranks
            ctx.comm.syncbn_exchange_(stats, (BN_SLOTS, [(0, 2 * C)]))
            n *= rt.world
        mi = torch.empty(2, C, device=dev, dtype=torch.float32)
        ss = torch.empty(2, C, device=dev, dtype=torch.float32)
        lib = L.load()
        rp = None if res_pre is None else _nhwc(res_pre)
        rq = None if res_post is None else _nhwc(res_post)
        for r in (rp, rq):
            if r is not None and (r.dtype != raw.dtype or r.shape != raw.shape):
                raise RuntimeError("conv_bn_act: residual dtype/shape mismatch %s%s vs %s%s" % (r.dtype, tuple(r.shape), raw.dtype, tuple(raw.shape)))
        defer = len(links) > 5 and links[5] and links[3] is not None and rp is None and rq is None
        if defer:
            # the ONLY consumer finalizes and applies this BatchNorm + activation while it stages `raw` (can_defer_apply): its launch
            # fills mi / ss (saved below for the backward pass) and updates the running statistics
            y = raw
        else:
            y = torch.empty_like(raw)
            so_ = links[3] if len(links) > 3 else None
            planes = None
            if so_ is not None and so_.want_planes and rp is None and rq is None and not (act & ACT_POST_RELU) and _WGRAD_PLANES:
                Bp, Hp, Wp, _ = raw.shape
                if lib.rssf_bn_finalize_apply_planes_supported(Bp, Hp, Wp, C, PLANES_PAD, L.dtype_code(raw)) == 1:
                    planes = rt.planes_buffer(gamma.data_ptr(), (C, Bp, Hp + 2 * PLANES_PAD, Wp + 2 * PLANES_PAD), raw.dtype, dev)
                    L.check(lib.rssf_bn_finalize_apply_planes(L.ptr(raw), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(rmean), L.ptr(rvar), L.ptr(mi),
                                                              L.ptr(ss), L.ptr(y), L.ptr(planes[0]), Bp, Hp, Wp, C, PLANES_PAD, act, n, momentum, eps,
                                                              int(training), L.dtype_code(raw), L.stream()), "rssf_bn_finalize_apply_planes")
            if so_ is not None:
                so_.planes = None if planes is None else (planes[0], PLANES_PAD, planes[1], planes[2])
            if planes is None:
                L.check(lib.rssf_bn_finalize_apply(L.ptr(raw), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(rmean), L.ptr(rvar), L.ptr(mi), L.ptr(ss),
                                                   L.ptr(rp), L.ptr(rq), L.ptr(y), rows, C, act, n, momentum, eps, int(training), L.dtype_code(raw),
                                                   L.stream()), "rssf_bn_finalize_apply")
        if act & ACT_POST_RELU and (defer or (len(links) > 3 and links[3] is not None)):
            raise RuntimeError("conv_bn_act: a ReLU behind res_post belongs to a layer without a BatchNorm-statistics link")
        ctx.rq = rq if (act & ACT_POST_RELU) else None          # (held by the node: the residual stream is alive until the block's backward anyway)
        ctx.save_for_backward(xh, raw, ss, mi, rp, *weights)
        ctx.meta = (spec, act, training, n, exchanged, nbias, len(weights), res_pre is not None, res_post is not None, x.requires_grad)
        ctx.rt = rt
        ctx.params = (gamma, beta, weights, biases)
        ctx.links = links[:2]             # (sink, deposit) GradLinks or (None, None)
        ctx.accum = links[2] if len(links) > 2 else None      # GradAccum of a multi-consumer input
        ctx.stats_out, ctx.stats_in = (links[3], links[4]) if len(links) > 4 else (None, None)      # BnBwdLinks
        if ctx.stats_out is not None:
            so = ctx.stats_out
            so.raw, so.ss, so.rp, so.act, so.C, so.sums = raw, ss, rp, act, C, None
            so.deferred = bool(defer)
            so.pre = (stats, gamma, beta, rmean, rvar, mi, ss, n, momentum, eps, training) if defer else None
        ctx.xpre = (si_.ss, si_.act) if preact is not None else None
        ctx.xplanes = si_.planes if (si_ is not None and preact is None) else None      # (buffer, pad): xh transposed, for the weight gradient
        return _nchw(y)

    @staticmethod
    def backward(ctx, dy):
        spec, act, training, n, exchanged, nbias, nw, has_pre, has_post, x_req = ctx.meta
        if spec.parts is not None:
            raise NotImplementedError("librssf conv: convolutions of more than 19 taps are forward-only (the CAM path's frozen stem)")
        rt = ctx.rt
        xh, raw, ss, mi, rp = ctx.saved_tensors[:5]
        weights = ctx.saved_tensors[5:]
        dyh = _nhwc(dy)
        if dyh.dtype != raw.dtype:
            dyh = dyh.to(raw.dtype)
        C = spec.cout
        rows = raw.numel() // C
        lib = L.load()
        so = ctx.stats_out
        sums = None
        if so is not None:               # the consumer's data-gradient launch already summed {dz, dz*raw} (BnBwdLink)
            sums, so.raw, so.ss, so.rp, so.sums = so.sums, None, None, None, None
        post = bool(act & ACT_POST_RELU)       # y = relu(act(z) + res_post): the kernels mask dy where y <= 0 and hand back d(res_post)
        rq = ctx.rq
        if sums is None and post:
            sums = _zeros(BN_BWD_SLOTS * 2 * C, raw.device, rt)
            dws = None
            if rt.deterministic:
                dws = torch.empty(lib.rssf_bn_bwd_reduce_workspace_elems(rows, C), device=raw.device, dtype=torch.float32)
            L.check(lib.rssf_bn_bwd_reduce_post(L.ptr(dyh), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(rq), L.ptr(sums), rows, C, act, L.ptr(dws),
                                                L.dtype_code(raw), L.stream()), "rssf_bn_bwd_reduce_post")
        elif sums is None:
            sums = _zeros(BN_BWD_SLOTS * 2 * C, raw.device, rt)
            dws = None
            if rt.deterministic:
                dws = torch.empty(lib.rssf_bn_bwd_reduce_workspace_elems(rows, C), device=raw.device, dtype=torch.float32)
            L.check(lib.rssf_bn_bwd_reduce(L.ptr(dyh), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(sums), rows, C, act, L.ptr(dws), L.dtype_code(raw),
                                           L.stream()), "rssf_bn_bwd_reduce")
        pscale = 1.0
        if exchanged:
            # the input gradient needs the GLOBAL {sum dz, sum dz*raw}; gamma/beta take the data-parallel MEAN of the local
            # sums (torch SyncBatchNorm + DDP), which is the global total / world
            ctx.comm.syncbn_exchange_(sums, (BN_BWD_SLOTS, [(0, 2 * C)]))
            pscale = 1.0 / rt.world
        draw = torch.empty_like(raw)
        dres = torch.empty_like(raw) if has_pre else None
        p_gamma, p_beta, p_weights, p_biases = ctx.params
        dgamma, dg_direct = grad_target(p_gamma, rt)
        dbeta, db_direct = grad_target(p_beta, rt)
        wt = [grad_target(w, rt) for w in p_weights]

        xpre = ctx.xpre             # xh is the producer's RAW output: the weight gradient applies its BatchNorm + activation on load

        bn_no_draw_ok = ctx.accum is None

        def weight_grads(bn, dgrad=None):
            gbs = []
            if nbias == 1:          # the weight-gradient kernel accumulates straight into the bias gradient: no staging buffer, no add
                tb, direct = grad_target(p_biases[0], rt)
                _conv_wgrad(spec, draw, xh, [t[0] for t in wt], tb, rt, bn=bn, xpre=xpre, planes=ctx.xplanes, dgrad=dgrad, no_draw=no_draw)
                gbs.append(grad_result(p_biases[0], tb, direct, rt))
            else:
                db = _zeros(C, raw.device, rt) if nbias else None
                _conv_wgrad(spec, draw, xh, [t[0] for t in wt], db, rt, bn=bn, xpre=xpre, planes=ctx.xplanes, no_draw=no_draw)
                tbs = [grad_target(b, rt) for b in p_biases]      # every summed conv's bias sees the same gradient: one launch
                if tbs:
                    d = [t[0] for t in tbs] + [None, None]
                    L.check(lib.rssf_vec_add_to3(L.ptr(db), L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), C, L.stream()), "rssf_vec_add_to3")
                for b, (tb, direct) in zip(p_biases, tbs):
                    gbs.append(grad_result(b, tb, direct, rt))
            return gbs

        # the BatchNorm-backward apply rides in the weight-gradient launch where a kernel for that exists (rssf_conv_wgrad_bnapply:
        # the entry point falls back to the two launches itself); channel counts the kernels would see padded keep the two calls
        vch = 8 if raw.dtype == torch.bfloat16 else 4
        # (an input whose channels were padded - the 3-channel image - is fine: the weight gradient of the padded problem is folded back
        # by _conv_wgrad; `draw` has no reader when the layer needs no data gradient: the stem's launch then never writes it)
        fuse_apply = _FUSED_BN_APPLY and (spec.cin % vch == 0 or xpre is None) and spec.cout % vch == 0 and not post
        no_draw = bool(bn_no_draw_ok and not x_req)
        dpost = None
        # a point-wise layer whose data gradient has no rider (no skip gradient to add, no producer statistics to collect, no shared
        # accumulator): the weight-gradient launch forms dx too (rssf_conv_wgrad_bnapply_dgrad: MlpDWBN's fc1)
        dx_fused = None
        wg_done = False
        si0 = ctx.stats_in
        if (fuse_apply and _FUSED_PW_DGRAD and x_req and nbias == 1 and nw == 1 and rp is None and xpre is None and ctx.xplanes is None and
                ctx.accum is None and ctx.links[0] is None and (si0 is None or si0.raw is None or rt.deterministic) and spec.parts is None and
                raw.dtype == torch.bfloat16 and weights[0].dtype == torch.float32 and weights[0].is_contiguous() and
                xh.shape[3] == spec.cin and raw.shape[3] == spec.cout):
            Bx, Hx, Wx, _ = xh.shape
            if lib.rssf_conv_wgrad_bnapply_dgrad_supported(Bx, Hx, Wx, spec.cin, raw.shape[1], raw.shape[2], spec.cout, spec.stride, spec.ntaps,
                                                           spec.c_dy, spec.c_dx, 0, L.dtype_code(raw)) == 1:
                dx_fused = torch.empty_like(xh)
        if fuse_apply and dx_fused is not None:
            gbs = weight_grads((dyh, raw, ss, mi, sums, rp, dres, dgamma, dbeta, act, n, training, pscale), dgrad=(weights[0], dx_fused))
        elif fuse_apply:
            gbs = weight_grads((dyh, raw, ss, mi, sums, rp, dres, dgamma, dbeta, act, n, training, pscale))
        elif post:
            dpost = torch.empty_like(raw) if has_post else None
            L.check(lib.rssf_bn_bwd_apply_post(L.ptr(dyh), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), L.ptr(rp), L.ptr(rq), L.ptr(draw), L.ptr(dres),
                                               L.ptr(dpost), L.ptr(dgamma), L.ptr(dbeta), rows, C, act, n, int(training), pscale, L.dtype_code(raw),
                                               L.stream()), "rssf_bn_bwd_apply_post")
        else:
            L.check(lib.rssf_bn_bwd_apply(L.ptr(dyh), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), L.ptr(rp), L.ptr(draw), L.ptr(dres),
                                          L.ptr(dgamma), L.ptr(dbeta), rows, C, act, n, int(training), pscale, L.dtype_code(raw), L.stream()),
                    "rssf_bn_bwd_apply")
        sink, deposit = ctx.links
        if deposit is not None and dres is not None:
            deposit.value, dres = dres, None              # the first conv of the block folds it into its data gradient
        addend = None
        if sink is not None:
            addend, sink.value = sink.value, None
        accum = ctx.accum
        if x_req and accum is not None:
            if addend is not None:
                raise RuntimeError("GradAccum: a convolution cannot be both the sink of a residual link and an accumulating consumer")
            _accumulate_dgrad(accum, spec, draw, weights, xh.shape, rt)
            dx = None
        elif x_req and dx_fused is not None:
            dx = _nchw(dx_fused)
        elif x_req:
            si, bn = ctx.stats_in, None
            if si is not None and si.raw is not None and not rt.deterministic:
                si.sums = _zeros(BN_BWD_SLOTS * 2 * si.C, raw.device, rt)
                bn = (si, si.sums)
            # a point-wise layer behind a deferred producer (xpre) whose own apply was a separate pass: weight gradient, data gradient and
            # the producer's statistics in ONE pass over the producer's raw output (rssf_conv_wgrad_preact_dgrad: MlpDWBN's fc2)
            if (bn is not None and xpre is not None and not fuse_apply and _FUSED_PW_DGRAD and addend is None and nbias == 1 and nw == 1 and
                    si.rp is None and si.raw.data_ptr() == xh.data_ptr() and si.act == xpre[1] and spec.parts is None and
                    weights[0].dtype == torch.float32 and weights[0].is_contiguous() and xh.shape[3] == spec.cin and
                    lib.rssf_conv_wgrad_preact_dgrad_supported(xh.shape[0], xh.shape[1], xh.shape[2], spec.cin, spec.cout,
                                                               L.dtype_code(xh)) == 1):
                dxh = torch.empty_like(xh)
                gbs = weight_grads(None, dgrad=(weights[0], dxh, si.sums))
                wg_done = True
                dx = _nchw(dxh)
            else:
                dx = _nchw(_conv_dgrad(spec, draw, weights, xh.shape, addend, rt, bn=bn))
        else:
            if addend is not None:
                raise RuntimeError("GradLink: a skip gradient was deposited but this node computes no input gradient")
            dx = None
        if not fuse_apply and not wg_done:
            gbs = weight_grads(None)
        gws = [grad_result(w, t[0], t[1], rt) for w, t in zip(p_weights, wt)]
        return (dx, None if dres is None else _nchw(dres), (dy if dpost is None else _nchw(dpost)) if has_post else None,
                grad_result(p_gamma, dgamma, dg_direct, rt),
                grad_result(p_beta, dbeta, db_direct, rt), None, None, None, None, None, None, None, None, None, None, *gws, *gbs)


def _is_plain3x3(spec):
    """3x3 / stride 1 / padding 1 / single bias-free source in the standard tap order: what the grouped 3x3 entry points take."""
    return (spec.parts is None and spec.ntaps == 9 and spec.stride == 1 and len(spec.ksizes) == 1 and spec.ksizes[0] == 3
            and spec.dy == [t // 3 - 1 for t in range(9)] and spec.dx == [t % 3 - 1 for t in range(9)])


def _chunks(idx):
    """Runs of at most RSSF_GROUP_MAX item indices (one grouped launch each)."""
    return [idx[k:k + L.GROUP_MAX] for k in range(0, len(idx), L.GROUP_MAX)]


def _group_conv3x3(entries, mirrored, code):
    """entries: dicts of rssf_conv3x3_item fields (tensors or None, scalars) -> grouped launches of <= RSSF_GROUP_MAX problems."""
    lib = L.load()
    for ch in _chunks(list(range(len(entries)))):
        arr = (L.Conv3x3Item * len(ch))()
        for k, i in enumerate(ch):
            it, e = arr[k], entries[i]
            for name, v in e.items():
                setattr(it, name, v.data_ptr() if torch.is_tensor(v) else (0 if v is None else v))
        L.check(lib.rssf_conv3x3_group(ctypes.cast(arr, ctypes.c_void_p), len(ch), int(mirrored), code, L.stream()), "rssf_conv3x3_group")


class _ConvBNActGroup(torch.autograd.Function):
    """N INDEPENDENT Conv2d -> BatchNorm2d -> act (+ residual before the activation) layers as one autograd node: the parallel
    branches of a HighResolutionModule run the same BasicBlock step at the same moment (_hrnet_rssformer.py:216-246, 410-423),
    so do the fuse paths of one depth (:361-405).  Every phase of the node is ONE grouped launch where a grouped kernel exists
    (rssf.h "Grouped launches": 3x3 / stride-1 convolutions forward, data gradient and weight gradient with the fused
    BatchNorm-backward apply, the BatchNorm finalize+apply and backward-statistics passes) - the launches of the branches are
    latency-bound one by one and fill each other's bubbles in one grid - and there is ONE SyncBN exchange per direction: the
    statistics buffers are carved out of one contiguous allocation and summed over the ranks by a single collective.
    Per item (7 tensor slots): x, res_pre, gamma, beta, running_mean, running_var, weight.  Single bias-free convolutions only
    (every HRNet convolution); meta = (spec, act, training, momentum, eps, sync, sink link, deposit link, GradAccum, defer)."""

    SLOTS = 7

    @staticmethod
    def forward(ctx, metas, glinks, *flat):
        rt = current()
        lib = L.load()
        n_items = len(metas)
        it = [flat[i * 7:(i + 1) * 7] for i in range(n_items)]
        L.require_gpu(*[t[0] for t in it])
        dev = it[0][0].device
        gout, gin = glinks
        training = [m[2] for m in metas]
        sizes = [BN_SLOTS * 2 * m[0].cout if tr else 0 for m, tr in zip(metas, training)]
        stats_all = _zeros(sum(sizes), dev, rt) if sum(sizes) else None
        xs, stats, pres, o = [], [], [], 0
        for i, ((x, rp, gamma, beta, rm, rv, w), m, sz) in enumerate(zip(it, metas, sizes)):
            xs.append(_nhwc(x))
            stats.append(stats_all[o:o + sz] if sz else None)
            o += sz
            si = gin.items[i] if gin is not None else None
            pres.append((si.pre, si.act, si.ss) if (si is not None and si.deferred) else None)     # x_i is the producer's RAW output
        # ---- convolutions: one grouped launch for the plain 3x3 items, the others one by one
        bf16 = all(xh.dtype == torch.bfloat16 for xh in xs)
        g3 = [i for i in range(n_items) if bf16 and _GROUP_LAUNCH and not rt.deterministic and _is_plain3x3(metas[i][0])
              and xs[i].shape[3] == metas[i][0].cin and metas[i][0].cin % 8 == 0 and metas[i][0].cout % 8 == 0]
        raws = [None] * n_items
        for kind in (False, True):                          # plain inputs, then pre-activation inputs (two kernels)
            sel = [i for i in g3 if (pres[i] is not None) == kind]
            if len(sel) < 2:
                continue
            entries = []
            for i in sel:
                spec, xh = metas[i][0], xs[i]
                B, H, W, C = xh.shape
                raws[i] = torch.empty(B, H, W, spec.cout, device=dev, dtype=xh.dtype)
                e = dict(in_=xh, wpk=_pack(spec, [it[i][6]], False, xh.dtype, dev, rt), out=raws[i], stats=stats[i], B=B, H=H, W=W, Cin=C, Cout=spec.cout)
                if kind:
                    (pst, pga, pbe, prm, prv, pmi, pss, pn, pmom, peps, ptr), pact, _ = pres[i]
                    e.update(pre_stats=pst, pre_gamma=pga, pre_beta=pbe, pre_running_mean=prm, pre_running_var=prv, pre_mean_invstd=pmi, pre_ss=pss,
                             pre_n=pn, pre_momentum=pmom, pre_eps=peps, pre_training=int(ptr), pre_act=pact)
                entries.append(e)
            _group_conv3x3(entries, False, L.RSSF_BF16)
        for i in range(n_items):
            if raws[i] is None:
                raws[i] = _conv_forward(metas[i][0], xs[i], [it[i][6]], None, stats[i], rt,
                                        preact=None if pres[i] is None else (pres[i][0], pres[i][1]))
        # ---- ONE exchange
        exchanged = stats_all is not None and rt.exchanging() and all(m[5] for m in metas)
        comm = rt.exchange_comm() if exchanged else None
        if exchanged:
            comm.syncbn_exchange_(stats_all, (BN_SLOTS, [(sum(sizes[:k]), 2 * m[0].cout) for k, m in enumerate(metas) if sizes[k]]))
        # ---- finalize + apply: one grouped launch (an item whose only consumer applies on load keeps its raw output)
        outs, saved, ns, defers, todo = [], [], [], [], []
        for i, ((x, rp, gamma, beta, rm, rv, w), m, xh, raw, st) in enumerate(zip(it, metas, xs, raws, stats)):
            spec, act, tr, mom, eps = m[:5]
            C = spec.cout
            rows = raw.numel() // C
            n = float(rows) * (rt.world if (exchanged and tr) else 1)
            rph = None if rp is None else _nhwc(rp)
            if rph is not None and (rph.dtype != raw.dtype or rph.shape != raw.shape):
                raise RuntimeError("conv_bn_act_group: residual dtype/shape mismatch")
            mi = torch.empty(2, C, device=dev, dtype=torch.float32)
            ss = torch.empty(2, C, device=dev, dtype=torch.float32)
            defer = bool(len(m) > 9 and m[9] and gout is not None and rph is None)
            if defer:
                y = raw
            else:
                y = torch.empty_like(raw)
                todo.append(dict(raw=raw, stats=st, gamma=gamma, beta=beta, running_mean=rm, running_var=rv, mean_invstd=mi, scale_shift=ss,
                                 res_pre=rph, res_post=None, y=y, rows=rows, n=n, momentum=mom, eps=eps, C=C, act=act, training=int(tr)))
            outs.append(_nchw(y))
            saved += [xh, raw, ss, mi, rph, w]
            ns.append(n)
            defers.append(defer)
        code = L.dtype_code(raws[0])
        same = all(r.dtype == raws[0].dtype for r in raws)
        for ch in (_chunks(list(range(len(todo)))) if (same and _GROUP_LAUNCH) else [[k] for k in range(len(todo))]):
            arr = (L.BnApplyItem * len(ch))()
            for k, j in enumerate(ch):
                for name, v in todo[j].items():
                    setattr(arr[k], name, v.data_ptr() if torch.is_tensor(v) else (0 if v is None else v))
            L.check(lib.rssf_bn_finalize_apply_group(ctypes.cast(arr, ctypes.c_void_p), len(ch), L.dtype_code(todo[ch[0]]["raw"]), L.stream()),
                    "rssf_bn_finalize_apply_group")
        ctx.save_for_backward(*saved)
        ctx.metas, ctx.ns, ctx.exchanged, ctx.rt = metas, ns, exchanged, rt
        ctx.comm = comm
        ctx.gout, ctx.gin = gout, gin
        ctx.xpres = [None if p is None else (p[2], p[1]) for p in pres]      # (scale/shift, act) of a pre-activation input operand
        if gout is not None:
            for i, (lk, m) in enumerate(zip(gout.items, metas)):
                xh_, raw_, ss_, mi_, rph_, w_ = saved[i * 6:(i + 1) * 6]
                lk.raw, lk.ss, lk.rp, lk.act, lk.C, lk.sums = raw_, ss_, rph_, m[1], m[0].cout, None
                lk.deferred = defers[i]
                tr = m[2]
                lk.pre = (stats[i], it[i][2], it[i][3], it[i][4], it[i][5], mi_, ss_, ns[i], m[3], m[4], tr) if defers[i] else None
        ctx.params = [(t[2], t[3], t[6]) for t in it]
        ctx.x_req = [t[0].requires_grad for t in it]
        ctx.has_pre = [t[1] is not None for t in it]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        rt, metas, lib = ctx.rt, ctx.metas, L.load()
        sv = ctx.saved_tensors
        n_items = len(metas)
        dev = sv[1].device
        sizes = [BN_BWD_SLOTS * 2 * m[0].cout for m in metas]
        gout = ctx.gout
        fused = gout is not None and gout.sums_all is not None and gout.filled == n_items     # the consumer group's data gradients summed them
        sums_all = gout.sums_all if fused else _zeros(sum(sizes), dev, rt)
        if gout is not None:
            gout.sums_all, gout.filled = None, 0
            for lk in gout.items:
                lk.raw = lk.ss = lk.rp = lk.pre = None
        dyhs, sums, o = [], [], 0
        for i, (m, sz) in enumerate(zip(metas, sizes)):
            raw = sv[i * 6 + 1]
            dyh = _nhwc(dys[i])
            if dyh.dtype != raw.dtype:
                dyh = dyh.to(raw.dtype)
            dyhs.append(dyh)
            sums.append(sums_all[o:o + sz])
            o += sz
        same = all(sv[i * 6 + 1].dtype == sv[1].dtype for i in range(n_items))
        if not fused:
            if rt.deterministic or not (same and _GROUP_LAUNCH):
                for i, m in enumerate(metas):
                    xh, raw, ss, mi, rph, w = sv[i * 6:(i + 1) * 6]
                    C = m[0].cout
                    rows = raw.numel() // C
                    dws = None
                    if rt.deterministic:
                        dws = torch.empty(lib.rssf_bn_bwd_reduce_workspace_elems(rows, C), device=dev, dtype=torch.float32)
                    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dyhs[i]), L.ptr(raw), L.ptr(ss), L.ptr(rph), L.ptr(sums[i]), rows, C, m[1], L.ptr(dws),
                                                   L.dtype_code(raw), L.stream()), "rssf_bn_bwd_reduce")
            else:
                for ch in _chunks(list(range(n_items))):
                    arr = (L.BnReduceItem * len(ch))()
                    for k, i in enumerate(ch):
                        xh, raw, ss, mi, rph, w = sv[i * 6:(i + 1) * 6]
                        C = metas[i][0].cout
                        q = arr[k]
                        q.dy, q.raw, q.scale_shift, q.res_pre, q.sums = dyhs[i].data_ptr(), raw.data_ptr(), ss.data_ptr(), (0 if rph is None else rph.data_ptr()), sums[i].data_ptr()
                        q.rows, q.C, q.act = raw.numel() // C, C, metas[i][1]
                    L.check(lib.rssf_bn_bwd_reduce_group(ctypes.cast(arr, ctypes.c_void_p), len(ch), L.dtype_code(sv[1]), L.stream()), "rssf_bn_bwd_reduce_group")
        pscale = 1.0
        if ctx.exchanged:
            ctx.comm.syncbn_exchange_(sums_all, (BN_BWD_SLOTS, [(sum(sizes[:k]), 2 * m[0].cout) for k, m in enumerate(metas)]))
            pscale = 1.0 / rt.world
        # ---- per item: gradient buffers; then the weight gradients (with the fused BatchNorm-backward apply) and the data gradients,
        #      each phase ONE grouped launch for the plain 3x3 items
        st = []
        for i, m in enumerate(metas):
            spec, act, tr = m[0], m[1], m[2]
            xh, raw, ss, mi, rph, w = sv[i * 6:(i + 1) * 6]
            p_gamma, p_beta, p_w = ctx.params[i]
            d = dict(spec=spec, act=act, tr=tr, xh=xh, raw=raw, ss=ss, mi=mi, rph=rph, w=w, C=spec.cout, rows=raw.numel() // spec.cout)
            d["draw"] = torch.empty_like(raw)
            d["dres"] = torch.empty_like(raw) if ctx.has_pre[i] else None
            d["dgamma"], d["dg_direct"] = grad_target(p_gamma, rt)
            d["dbeta"], d["db_direct"] = grad_target(p_beta, rt)
            d["tw"], d["wd"] = grad_target(p_w, rt)
            vch = 8 if raw.dtype == torch.bfloat16 else 4
            d["fuse_apply"] = _FUSED_BN_APPLY and spec.cin % vch == 0 and spec.cout % vch == 0      # see _ConvBNAct.backward
            d["xpre"] = ctx.xpres[i]
            if d["xpre"] is not None and not d["fuse_apply"]:
                raise RuntimeError("conv_bn_act_group: a pre-activation input needs the fused weight-gradient path")
            d["g3"] = (_GROUP_LAUNCH and raw.dtype == torch.bfloat16 and not rt.deterministic and _is_plain3x3(spec) and d["fuse_apply"]
                       and xh.shape[3] == spec.cin)
            st.append(d)
        # weight gradients
        grouped_w = set()
        for res_kind in (False, True):
            for pre_kind in (False, True):
                sel = [i for i in range(n_items) if st[i]["g3"] and (st[i]["rph"] is not None) == res_kind and (st[i]["xpre"] is not None) == pre_kind]
                if len(sel) < 2:
                    continue
                for ch in _chunks(sel):
                    if len(ch) < 2:
                        continue
                    if not _group_wgrad3x3([st[i] for i in ch], [dyhs[i] for i in ch], [sums[i] for i in ch], [ctx.ns[i] for i in ch], pscale, rt):
                        continue
                    grouped_w.update(ch)
        # the layers whose weight-gradient kernel cannot carry the BatchNorm-backward apply (1x1 and strided convolutions: the fuse
        # paths of one depth) get it as ONE grouped launch; their weight gradients then read the finished draw
        applied = set()
        sel = [i for i in range(n_items) if i not in grouped_w and st[i]["xpre"] is None and not _is_plain3x3(st[i]["spec"])
               and _GROUP_LAUNCH and not rt.deterministic]
        if len(sel) >= 2 and all(st[i]["raw"].dtype == st[sel[0]]["raw"].dtype for i in sel):
            for ch in _chunks(sel):
                if len(ch) < 2:
                    continue
                arr = (L.BnBwdApplyItem * len(ch))()
                for k, i in enumerate(ch):
                    d, q = st[i], arr[k]
                    q.dy, q.raw, q.scale_shift, q.mean_invstd, q.sums = dyhs[i].data_ptr(), d["raw"].data_ptr(), d["ss"].data_ptr(), d["mi"].data_ptr(), sums[i].data_ptr()
                    q.res_pre = 0 if d["rph"] is None else d["rph"].data_ptr()
                    q.draw, q.dres = d["draw"].data_ptr(), (0 if d["dres"] is None else d["dres"].data_ptr())
                    q.dgamma, q.dbeta = d["dgamma"].data_ptr(), d["dbeta"].data_ptr()
                    q.rows, q.n, q.C, q.act, q.training = d["rows"], ctx.ns[i], d["C"], d["act"], int(d["tr"])
                    q.param_grad_scale = pscale if d["tr"] else 1.0
                L.check(lib.rssf_bn_bwd_apply_group(ctypes.cast(arr, ctypes.c_void_p), len(ch), L.dtype_code(st[ch[0]]["raw"]), L.stream()),
                        "rssf_bn_bwd_apply_group")
                applied.update(ch)
        for i in range(n_items):
            d = st[i]
            if i in grouped_w:
                continue
            if i in applied:
                d["fuse_apply"] = False                # draw is final: the plain weight gradient below reads it
                continue
            if d["fuse_apply"]:
                _conv_wgrad(d["spec"], d["draw"], d["xh"], [d["tw"]], None, rt,
                            bn=(dyhs[i], d["raw"], d["ss"], d["mi"], sums[i], d["rph"], d["dres"], d["dgamma"], d["dbeta"], d["act"], ctx.ns[i], d["tr"],
                                pscale if d["tr"] else 1.0), xpre=d["xpre"])
            else:
                L.check(lib.rssf_bn_bwd_apply(L.ptr(dyhs[i]), L.ptr(d["raw"]), L.ptr(d["ss"]), L.ptr(d["mi"]), L.ptr(sums[i]), L.ptr(d["rph"]), L.ptr(d["draw"]),
                                              L.ptr(d["dres"]), L.ptr(d["dgamma"]), L.ptr(d["dbeta"]), d["rows"], d["C"], d["act"], ctx.ns[i], int(d["tr"]),
                                              pscale if d["tr"] else 1.0, L.dtype_code(d["raw"]), L.stream()), "rssf_bn_bwd_apply")
        # data gradients
        dxs = [None] * n_items
        entries, owners = [], []
        for i, m in enumerate(metas):
            d = st[i]
            spec, sink, deposit = d["spec"], m[6], m[7]
            if deposit is not None and d["dres"] is not None:
                deposit.value, d["dres_out"] = d["dres"], None
            else:
                d["dres_out"] = d["dres"]
            addend = None
            if sink is not None:
                addend, sink.value = sink.value, None
            accum = m[8] if len(m) > 8 else None
            if ctx.x_req[i] and accum is not None:
                if addend is not None:
                    raise RuntimeError("GradAccum: a convolution cannot be both the sink of a residual link and an accumulating consumer")
                _accumulate_dgrad(accum, spec, d["draw"], [d["w"]], d["xh"].shape, rt)
            elif ctx.x_req[i]:
                gin, bn = ctx.gin, None
                if gin is not None and gin.items[i].raw is not None and not rt.deterministic:
                    bn = (gin.items[i], gin.slot(i, dev, rt))
                B, H, W, C = d["xh"].shape
                if (_GROUP_LAUNCH and d["raw"].dtype == torch.bfloat16 and _is_plain3x3(spec) and C == spec.cin and C % 8 == 0 and spec.cout % 8 == 0
                        and (addend is None or (addend.shape == d["xh"].shape and addend.dtype == d["xh"].dtype and addend.is_contiguous()))):
                    dx = torch.empty(B, H, W, C, device=dev, dtype=d["raw"].dtype)
                    e = dict(in_=d["draw"], wpk=_pack(spec, [d["w"]], True, d["raw"].dtype, dev, rt), out=dx, addend=addend, B=B, H=H, W=W, Cin=spec.cout, Cout=C)
                    if bn is not None:
                        link, sm = bn
                        if link.raw.shape != dx.shape or link.raw.dtype != dx.dtype or link.C != C:
                            raise RuntimeError("GroupBwdLink: the producer's output is not this convolution's input")
                        e.update(bn_raw=link.raw, bn_res=link.rp, bn_ss=link.ss, bn_sums=sm, bn_act=link.act)
                    entries.append(e)
                    owners.append(i)
                    dxs[i] = dx
                else:
                    dxs[i] = _conv_dgrad(spec, d["draw"], [d["w"]], d["xh"].shape, addend, rt, bn=bn)
            elif addend is not None:
                raise RuntimeError("GradLink: a skip gradient was deposited but this node computes no input gradient")
        if entries:
            _group_conv3x3(entries, True, L.RSSF_BF16)         # (a single entry runs through the one-problem kernels inside the entry point)
        grads = []
        for i in range(n_items):
            d = st[i]
            if not d["fuse_apply"]:
                _conv_wgrad(d["spec"], d["draw"], d["xh"], [d["tw"]], None, rt)
            p_gamma, p_beta, p_w = ctx.params[i]
            grads += [None if dxs[i] is None else _nchw(dxs[i]), None if d["dres_out"] is None else _nchw(d["dres_out"]),
                      grad_result(p_gamma, d["dgamma"], d["dg_direct"], rt), grad_result(p_beta, d["dbeta"], d["db_direct"], rt), None, None,
                      grad_result(p_w, d["tw"], d["wd"], rt)]
        return (None, None, *grads)


def _group_wgrad3x3(items, dyhs, sums, ns, pscale, rt):
    """Weight gradients (+ fused BatchNorm-backward apply) of <= RSSF_GROUP_MAX plain 3x3 layers as one launch, under the step's
    WgradPlan (the split-K partials of every item wait for the plan's batched reduction).  False: the plan could not serve the
    items (call sequence changed, a gradient buffer already has a deferred job) - the caller launches them one by one."""
    lib = L.load()
    plan = rt.wgrad_plan
    n = len(items)
    keys = []
    for d in items:
        xh, spec = d["xh"], d["spec"]
        B, H, W, C = xh.shape
        keys.append((id(spec), B, H, W, C, H, W, spec.cout, xh.dtype, (d["tw"].data_ptr(),)))
    # all-or-nothing: the plan's cursor only moves when every item of the launch gets its slot
    if plan is not None:
        if plan.recording or not plan.built:
            return False                 # the recording step (and a plan that fell back for good) takes the calls one by one
        i0 = plan.cursor
        if i0 + n > len(plan.keys) or any(plan.keys[i0 + k] != keys[k] for k in range(n)):
            return False
        if any(plan.duplicate_target([d["tw"].data_ptr()]) for d in items):
            raise RuntimeError("WgradPlan: a grouped weight gradient targets a buffer that already has a deferred job in this step")
    arr = (L.Wgrad3x3Item * n)()
    jobs, refs, held = [], [], []
    for k, d in enumerate(items):
        xh, spec = d["xh"], d["spec"]
        B, H, W, C = xh.shape
        if plan is not None:
            ws, job, ref = plan.slot(keys[k])
        else:                            # no plan (a model run outside a Trainer): own workspace, immediate second stage
            ws = torch.empty(lib.rssf_conv_wgrad_workspace_elems(B, H, W, C, spec.cout, 9), device=xh.device, dtype=torch.float32)
            job = ref = None
        jobs.append(job)
        refs.append(ref)
        held.append(ws)
        q = arr[k]
        q.dout, q.in_, q.dw, q.workspace = 0, xh.data_ptr(), d["tw"].data_ptr(), ws.data_ptr()
        q.defer_reduce = 0 if job is None else ctypes.addressof(job)
        q.bn_dy, q.bn_raw, q.bn_ss, q.bn_mi, q.bn_sums = dyhs[k].data_ptr(), d["raw"].data_ptr(), d["ss"].data_ptr(), d["mi"].data_ptr(), sums[k].data_ptr()
        q.bn_res = 0 if d["rph"] is None else d["rph"].data_ptr()
        q.draw, q.dres = d["draw"].data_ptr(), (0 if d["dres"] is None else d["dres"].data_ptr())
        q.dgamma, q.dbeta = d["dgamma"].data_ptr(), d["dbeta"].data_ptr()
        q.in_ss, q.in_act = (0, 0) if d["xpre"] is None else (d["xpre"][0].data_ptr(), d["xpre"][1])
        q.bn_n, q.pscale = ns[k], (pscale if d["tr"] else 1.0)
        q.bn_act, q.bn_training = d["act"], int(d["tr"])
        q.B, q.H, q.W, q.Cin, q.Cout = B, H, W, C, spec.cout
    L.check(lib.rssf_conv3x3_wgrad_group(ctypes.cast(arr, ctypes.c_void_p), n, L.RSSF_BF16, L.stream()), "rssf_conv3x3_wgrad_group")
    for job, ref in zip(jobs, refs):
        if job is not None and ref is not None and bytes(job) != bytes(ref):
            raise RuntimeError("WgradPlan: a deferred weight-gradient reduction changed between steps")
    return True


def conv_bn_act_group(items, stats_out=None, stats_in=None):
    """items: dicts with x, conv, bn, act and optionally res_pre, grad_sink, grad_deposit, grad_accum (see conv_bn_act).  Returns the list of
    outputs.  One SyncBN exchange for the whole group (see _ConvBNActGroup); falls back to individual nodes for a single item.
    stats_out / stats_in: GroupBwdLink of this group / of the group whose item i produced this group's x_i (see BnBwdLink)."""
    if len(items) == 1:
        d = items[0]
        return [conv_bn_act(d["x"], d["conv"], d["bn"], d.get("act", ACT_NONE), res_pre=d.get("res_pre"), grad_sink=d.get("grad_sink"),
                            grad_deposit=d.get("grad_deposit"), grad_accum=d.get("grad_accum"),
                            stats_out=None if stats_out is None else stats_out.items[0], stats_in=None if stats_in is None else stats_in.items[0],
                            defer_apply=bool(d.get("defer_apply", False)))]
    rt = current()
    metas, flat = [], []
    for d in items:
        conv, bn = d["conv"], d["bn"]
        if conv.bias is not None:
            raise NotImplementedError("conv_bn_act_group: bias-free convolutions only")
        training = bn.training or not bn.track_running_stats
        if training and bn.track_running_stats:
            bn._rssf_steps = getattr(bn, "_rssf_steps", 0) + 1
        sync = isinstance(bn, nn.SyncBatchNorm) or rt.sync_all_bn
        metas.append((spec_of([conv]), d.get("act", ACT_NONE), training, 0.1 if bn.momentum is None else bn.momentum, bn.eps, sync,
                      d.get("grad_sink"), d.get("grad_deposit"), d.get("grad_accum"), bool(d.get("defer_apply", False))))
        flat += [d["x"], d.get("res_pre"), bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.weight]
    return list(_ConvBNActGroup.apply(metas, (stats_out, stats_in), *flat))


class _ConvBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec, weight, bias):
        L.require_gpu(x)
        xh = _nhwc(x)
        rt = current()
        out = _conv_forward(spec, xh, [weight], None if bias is None else bias.float().contiguous(), None, rt)
        ctx.save_for_backward(xh, weight)
        ctx.rt = rt
        ctx.meta = (spec, bias is not None, x.requires_grad)
        ctx.params = (weight, bias)
        return _nchw(out)

    @staticmethod
    def backward(ctx, dy):
        spec, has_bias, x_req = ctx.meta
        xh, weight = ctx.saved_tensors
        dyh = _nhwc(dy)
        if dyh.dtype != xh.dtype:
            dyh = dyh.to(xh.dtype)
        dyh = _pad_channels(dyh)                 # once for both gradient launches (the 6-class head: 6 -> 8 channels)
        rt = ctx.rt
        dx = _nchw(_conv_dgrad(spec, dyh, [weight], xh.shape, None, rt)) if x_req else None
        p_w, p_b = ctx.params
        tw, wd = grad_target(p_w, rt)
        tb, bd = grad_target(p_b, rt) if has_bias else (None, False)
        _conv_wgrad(spec, dyh, xh, [tw], tb, rt)
        return dx, None, grad_result(p_w, tw, wd, rt), (grad_result(p_b, tb, bd, rt) if has_bias else None)


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW):
        L.require_gpu(x)
        xh = _nhwc(x)
        B, IH, IW, C = xh.shape
        out = torch.empty(B, OH, OW, C, device=x.device, dtype=x.dtype)
        L.check(L.load().rssf_upsample_bilinear(L.ptr(xh), L.ptr(out), B, IH, IW, OH, OW, C, 0, L.dtype_code(xh), L.stream()),
                "rssf_upsample_bilinear")
        ctx.shape = (B, IH, IW, OH, OW, C)
        return _nchw(out)

    @staticmethod
    def backward(ctx, dy):
        B, IH, IW, OH, OW, C = ctx.shape
        dyh = _nhwc(dy)
        dx = torch.empty(B, IH, IW, C, device=dy.device, dtype=dy.dtype)
        L.check(L.load().rssf_upsample_bilinear(L.ptr(dyh), L.ptr(dx), B, IH, IW, OH, OW, C, 1, L.dtype_code(dyh), L.stream()),
                "rssf_upsample_bilinear(bwd)")
        return _nchw(dx), None, None


class _NearestAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acc, x, scale):
        L.require_gpu(x)
        xh = _nhwc(x)
        B, IH, IW, C = xh.shape
        ah = None
        if acc is not None:
            ah = _nhwc(acc)
            if ah.dtype != xh.dtype or ah.shape != (B, IH * scale, IW * scale, C):
                raise RuntimeError("upsample_nearest_add: accumulator dtype/shape mismatch")
        out = torch.empty(B, IH * scale, IW * scale, C, device=x.device, dtype=x.dtype)
        L.check(L.load().rssf_upsample_nearest_add(L.ptr(ah), L.ptr(xh), L.ptr(out), B, IH, IW, scale, C, 0, L.dtype_code(xh), L.stream()),
                "rssf_upsample_nearest_add")
        ctx.shape = (B, IH, IW, scale, C, acc is not None)
        return _nchw(out)

    @staticmethod
    def backward(ctx, dy):
        B, IH, IW, scale, C, has_acc = ctx.shape
        dyh = _nhwc(dy)
        dx = torch.empty(B, IH, IW, C, device=dy.device, dtype=dy.dtype)
        L.check(L.load().rssf_upsample_nearest_add(None, L.ptr(dyh), L.ptr(dx), B, IH, IW, scale, C, 1, L.dtype_code(dyh), L.stream()),
                "rssf_upsample_nearest_add(bwd)")
        return (dy if has_acc else None), _nchw(dx), None


class _BilinearConcat(torch.autograd.Function):
    """cat([F.interpolate(f, size, 'bilinear', align_corners=True) for f in feats], dim=1) written slice by slice into one
    channels-last buffer (a feature map already at `size` is the identity case of the same kernel); the backward reads each
    slice of the concatenated gradient in place."""

    @staticmethod
    def forward(ctx, OH, OW, *feats):
        L.require_gpu(*feats)
        fh = [_nhwc(f) for f in feats]
        B = fh[0].shape[0]
        ctot = sum(f.shape[3] for f in fh)
        out = torch.empty(B, OH, OW, ctot, device=fh[0].device, dtype=fh[0].dtype)
        lib, es, off, shapes = L.load(), out.element_size(), 0, []
        for f in fh:
            _, IH, IW, C = f.shape
            L.check(lib.rssf_upsample_bilinear_slice(L.ptr(f), out.data_ptr() + off * es, B, IH, IW, OH, OW, C, ctot, 0, L.dtype_code(f),
                                                     L.stream()), "rssf_upsample_bilinear_slice")
            shapes.append((IH, IW, C))
            off += C
        ctx.meta = (B, OH, OW, ctot, shapes)
        return _nchw(out)

    @staticmethod
    def backward(ctx, dy):
        B, OH, OW, ctot, shapes = ctx.meta
        dyh = _nhwc(dy)
        lib, es, off, grads = L.load(), dyh.element_size(), 0, []
        for i, (IH, IW, C) in enumerate(shapes):
            if ctx.needs_input_grad[2 + i]:
                dx = torch.empty(B, IH, IW, C, device=dy.device, dtype=dy.dtype)
                L.check(lib.rssf_upsample_bilinear_slice(dyh.data_ptr() + off * es, L.ptr(dx), B, IH, IW, OH, OW, C, ctot, 1,
                                                         L.dtype_code(dyh), L.stream()), "rssf_upsample_bilinear_slice(bwd)")
                grads.append(_nchw(dx))
            else:
                grads.append(None)
            off += C
        return (None, None, *grads)


def upsample_bilinear_concat(feats, size):
    """torch.cat([f resized to `size` (bilinear, align_corners=True) for f in feats], dim=1), channels-last, in one buffer."""
    return _BilinearConcat.apply(int(size[0]), int(size[1]), *feats)


def upsample_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True)."""
    return _Bilinear.apply(x, int(size[0]), int(size[1]))


def head_upsample_softmax(logits, size, want_probs=True, want_pred=False):
    """Inference head: F.interpolate(logits, size, 'bilinear', align_corners=True).softmax(1) [and .argmax(1)] as ONE launch
    (no autograd: evaluation only).  Returns (probs logical [B,K,OH,OW] fp32 channels-last or None, pred [B,OH,OW] int32 or None)."""
    L.require_gpu(logits)
    lh = _nhwc(logits.detach())
    B, IH, IW, K = lh.shape
    OH, OW = int(size[0]), int(size[1])
    probs = torch.empty(B, OH, OW, K, device=lh.device, dtype=torch.float32) if want_probs else None
    pred = torch.empty(B, OH, OW, device=lh.device, dtype=torch.int32) if want_pred else None
    L.check(L.load().rssf_head_upsample_softmax(L.ptr(lh), L.ptr(probs), L.ptr(pred), B, IH, IW, OH, OW, K, L.dtype_code(lh), L.stream()),
            "rssf_head_upsample_softmax")
    return (None if probs is None else _nchw(probs)), pred


def image_to_channels_last(x, dtype):
    """Network input: fp32 image [B,C,H,W] (NCHW or channels-last memory, C <= 8 / 4) -> logical [B,Cp,H,W] tensor of `dtype` in
    channels-last memory with the channels zero-padded to the 16-byte vector width (Cp = 8 bf16 / 4 fp32), in ONE launch - the
    cast, the layout copy and the padding the first convolution needs (forward and weight gradient).  No autograd: the image
    carries no gradient."""
    L.require_gpu(x)
    B, C, H, W = x.shape
    cp = 8 if dtype == torch.bfloat16 else 4
    out = torch.empty(B, H, W, cp, device=x.device, dtype=dtype)
    sb, sc, sh, sw = x.stride()
    L.check(L.load().rssf_image_to_nhwc(L.ptr(x), L.ptr(out), B, C, H, W, sb, sc, sh, sw, L.dtype_code(out), L.stream()),
            "rssf_image_to_nhwc")
    return _nchw(out)


def max_pool_3x3_s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1), forward only (inference stem of the ResNet-50 CAM path)."""
    L.require_gpu(x)
    xh = _nhwc(x.detach())
    B, IH, IW, C = xh.shape
    out = torch.empty(B, (IH - 1) // 2 + 1, (IW - 1) // 2 + 1, C, device=xh.device, dtype=xh.dtype)
    L.check(L.load().rssf_maxpool3x3s2(L.ptr(xh), L.ptr(out), B, IH, IW, C, L.dtype_code(xh), L.stream()), "rssf_maxpool3x3s2")
    return _nchw(out)


class _NearestSum(torch.autograd.Function):
    """sum_k nn.Upsample(scale_factor=s_k, mode='nearest')(t_k) (s_k = 1: t_k itself) in one launch; backward: the block sums of the
    output gradient for the up-sampled terms in one launch, the gradient itself for the others."""

    @staticmethod
    def forward(ctx, scales, *terms):
        L.require_gpu(*terms)
        ths = [_nhwc(t) for t in terms]
        k0 = scales.index(1) if 1 in scales else None
        if k0 is not None:
            B, OH, OW, C = ths[k0].shape
        else:
            B, ih, iw, C = ths[0].shape
            OH, OW = ih * scales[0], iw * scales[0]
        for t, sc in zip(ths, scales):
            if t.dtype != ths[0].dtype or tuple(t.shape) != (B, OH // sc, OW // sc, C) or OH % sc or OW % sc:
                raise RuntimeError("upsample_nearest_sum: term of shape %s / scale %d does not fit [%d, %d, %d, %d]" % (tuple(t.shape), sc, B, OH, OW, C))
        out = torch.empty(B, OH, OW, C, device=ths[0].device, dtype=ths[0].dtype)
        ptrs = (ctypes.c_void_p * len(ths))(*[t.data_ptr() for t in ths])
        L.check(L.load().rssf_upsample_nearest_sum(ptrs, _ia(list(scales)), len(ths), L.ptr(out), B, OH, OW, C, 0, L.dtype_code(out), L.stream()),
                "rssf_upsample_nearest_sum")
        ctx.meta = (tuple(scales), B, OH, OW, C)
        return _nchw(out)

    @staticmethod
    def backward(ctx, dy):
        scales, B, OH, OW, C = ctx.meta
        dyh = _nhwc(dy)
        outs = [None if sc == 1 else torch.empty(B, OH // sc, OW // sc, C, device=dy.device, dtype=dy.dtype) for sc in scales]
        if any(o is not None for o in outs):
            ptrs = (ctypes.c_void_p * len(outs))(*[None if o is None else o.data_ptr() for o in outs])
            L.check(L.load().rssf_upsample_nearest_sum(ptrs, _ia(list(scales)), len(outs), L.ptr(dyh), B, OH, OW, C, 1, L.dtype_code(dyh), L.stream()),
                    "rssf_upsample_nearest_sum(bwd)")
        return (None, *[dy if o is None else _nchw(o) for o in outs])


_NEAREST_SUM = os.environ.get("RSSF_NEAREST_SUM", "1") != "0"      # A/B switch: the fuse sum of an output in one launch


def fuse_sum(terms, scales):
    """sum_k up(terms[k], scales[k]) (nearest, scale 1 = the term itself), in the order given - the `y = y + fuse[i][j](x[j])` loop of
    HighResolutionModule.forward (_hrnet_rssformer.py:424-435).  One launch where the kernel takes the operands (<= 4 dense terms of
    one dtype, channels a multiple of the vector), else the chain of pairwise launches."""
    terms, scales = list(terms), [int(sc) for sc in scales]
    if len(terms) == 1 and scales[0] == 1:
        return terms[0]
    t0 = terms[0]
    v = 8 if t0.dtype == torch.bfloat16 else 4
    if (_NEAREST_SUM and 2 <= len(terms) <= 4 and t0.is_cuda and t0.dtype in (torch.bfloat16, torch.float32) and t0.shape[1] % v == 0
            and all(t.dtype == t0.dtype and t.is_cuda for t in terms)):
        return _NearestSum.apply(tuple(scales), *terms)
    low = None
    for t, sc in zip(terms, scales):
        if sc == 1:
            low = t if low is None else add(low, t)
        else:
            low = upsample_nearest_add(low, t, sc)
    return low


def upsample_nearest_add(acc, x, scale):
    """(acc or 0) + nn.Upsample(scale_factor=scale, mode='nearest')(x)."""
    return _NearestAdd.apply(acc, x, int(scale))


class _CGFLLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, aux, ignore_index):
        L.require_gpu(logits, labels, aux)
        lh = _nhwc(logits)
        B, H, W, K = lh.shape
        labels = labels.contiguous()
        if labels.dtype != torch.int64 or labels.shape != (B, H, W):
            raise RuntimeError("cgfl_loss: labels must be int64 [B,H,W]")
        auxf = aux.detach().float().contiguous()
        acc = torch.empty(B, L.LOSS_ACC_ELEMS, device=lh.device, dtype=torch.float32)      # RSSF_LOSS_ACC_ELEMS (include/rssf.h) per sample
        out = torch.empty(2, device=lh.device, dtype=torch.float32)
        L.check(L.load().rssf_cgfl_loss_fwd(L.ptr(lh), L.ptr(labels), L.ptr(auxf), L.ptr(acc), L.ptr(out), B, H * W, K, auxf.shape[1],
                                            ignore_index, int(current().deterministic), L.dtype_code(lh), L.stream()), "rssf_cgfl_loss_fwd")
        ctx.save_for_backward(lh, labels, out)
        ctx.ignore_index = ignore_index
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        lh, labels, out = ctx.saved_tensors
        B, H, W, K = lh.shape
        dl = torch.empty_like(lh)
        d = dloss.detach().float().reshape(1).contiguous()
        L.check(L.load().rssf_cgfl_loss_bwd(L.ptr(lh), L.ptr(labels), L.ptr(out), L.ptr(d), L.ptr(dl), B, H * W, K, ctx.ignore_index,
                                            L.dtype_code(lh), L.stream()), "rssf_cgfl_loss_bwd")
        return _nchw(dl), None, None, None      # the aux head receives no gradient (the bracket is detached, CGFL.py:75-97)


def aux_head(f0, linear):
    """`linear(AdaptiveAvgPool2d(1)(f0).flatten(1))` for the image-level auxiliary scores (hrnet_aux.py:86-87, 99-100), forward only -
    the loss uses them under no_grad.  f0: logical NCHW channels-last; returns fp32 [B, K]."""
    L.require_gpu(f0)
    fh = _nhwc(f0.detach())
    B, H, W, C = fh.shape
    K = linear.out_features
    lib = L.load()
    ws = torch.empty(lib.rssf_aux_head_workspace_elems(B, C), device=fh.device, dtype=torch.float32)
    out = torch.empty(B, K, device=fh.device, dtype=torch.float32)
    w = linear.weight.detach()
    bias = None if linear.bias is None else linear.bias.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    L.check(lib.rssf_aux_head_fwd(L.ptr(fh), L.ptr(w), L.ptr(bias), L.ptr(ws), L.ptr(out), B, H * W, C, K, L.dtype_code(fh), L.stream()),
            "rssf_aux_head_fwd")
    return out


def cgfl_loss(logits, labels, aux, ignore_index=-1):
    """SegmentationLossaux 'ce' branch: logits logical NCHW (channels-last), labels int64 [B,H,W], aux [B,7]."""
    # detached HERE, not inside the node: the reference's loss only uses the aux scores under no_grad (CGFL.py:75-97), so the
    # aux head is not part of the autograd graph at all (its parameters keep grad None; torch.optim.SGD skips them)
    return _CGFLLoss.apply(logits, labels, aux.detach(), int(ignore_index))


def conv_bn_act(x, convs, bn, act=ACT_NONE, res_pre=None, res_post=None, grad_sink=None, grad_deposit=None, grad_accum=None,
                stats_out=None, stats_in=None, defer_apply=False, post_relu=False):
    """convs: one nn.Conv2d or a list of up to 3 summed convs; bn: nn.BatchNorm2d / nn.SyncBatchNorm.
    grad_sink / grad_deposit: GradLink of a residual block (see GradLink); grad_accum: GradAccum of a multi-consumer input;
    stats_out / stats_in: BnBwdLink of this layer / of the layer that produced x (see BnBwdLink); defer_apply (needs stats_out): return
    the RAW convolution output - the single consumer, given the link as stats_in, applies BatchNorm + activation on load
    (see can_defer_apply)."""
    convs = convs if isinstance(convs, (list, tuple)) else [convs]
    spec = spec_of(convs)
    training = bn.training or not bn.track_running_stats
    sync = isinstance(bn, nn.SyncBatchNorm) or current().sync_all_bn
    if training and bn.track_running_stats:
        bn._rssf_steps = getattr(bn, "_rssf_steps", 0) + 1      # num_batches_tracked, flushed by flush_bn_counters()
    weights = [c.weight for c in convs]
    biases = [c.bias for c in convs if c.bias is not None]
    if biases and len(biases) != len(convs):
        raise NotImplementedError("conv_bn_act: either all or none of the summed convs carry a bias")
    mom = 0.1 if bn.momentum is None else bn.momentum
    if post_relu:           # relu(act(bn(conv(x))) + res_post) in the layer's own BatchNorm pass (see ACT_POST_RELU)
        act = act | ACT_POST_RELU
    return _ConvBNAct.apply(x, res_pre, res_post, bn.weight, bn.bias, bn.running_mean, bn.running_var, spec, act, training, mom, bn.eps,
                            sync, len(biases), (grad_sink, grad_deposit, grad_accum, stats_out, stats_in, bool(defer_apply)), *weights, *biases)


def residual_link(x, res):
    """GradLink for `conv_a(x) ... + res` when the skip IS the block input and gradients flow into it (else None)."""
    return GradLink() if (res is x and torch.is_grad_enabled() and x.requires_grad) else None


def conv_bias(x, conv):
    return _ConvBias.apply(x, spec_of([conv]), conv.weight, conv.bias)


class _LinearShape:
    """What ConvSpec reads of a convolution, for an nn.Linear run as a 1 x 1 convolution over channels-last tokens."""

    def __init__(self, lin):
        self.in_channels, self.out_channels = lin.in_features, lin.out_features
        self.stride = self.kernel_size = self.dilation = (1, 1)
        self.padding = (0, 0)
        self.groups = 1


def conv_nhwc(xh, conv, addend=None):
    """Inference-only convolution + bias (+ addend) on a channels-last activation [B, H, W, C] -> [B, OH, OW, Cout].  `conv`: an
    nn.Conv2d (square, ungrouped; more than 19 taps run as chained launches) or an nn.Linear (1 x 1 over the tokens).  No autograd:
    the CAM extraction paths run under no_grad - where the packed weights are then kept until a parameter changes (`_pack`)."""
    L.require_gpu(xh)
    if isinstance(conv, nn.Linear):
        spec = conv.__dict__.get("_rssf_spec")
        if spec is None:
            spec = conv.__dict__["_rssf_spec"] = ConvSpec([_LinearShape(conv)])
        w = conv.weight.detach().view(conv.out_features, conv.in_features, 1, 1)
    else:
        spec, w = spec_of([conv]), conv.weight.detach()
    if addend is not None and spec.parts is not None:
        raise NotImplementedError("conv_nhwc: an addend on a tap-split convolution")
    bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
    return _conv_forward(spec, xh if xh.is_contiguous() else xh.contiguous(), [w], bias, None, current(), addend=addend,
                         cache_pack=not torch.is_grad_enabled())


def _specs_of_module(m):
    sp = list(m.__dict__.get("_rssf_specs", {}).values())
    if m.__dict__.get("_rssf_spec") is not None:
        sp.append(m.__dict__["_rssf_spec"])
    out = []
    for s_ in sp:
        out.append(s_)
        out += list(s_.parts or [])
    return out


def invalidate_packed(model):
    """Drop the packed-weight copies the inference path keeps on a model's convolutions (`_pack(cache=True)`).  The cache is keyed
    on (address, version counter) of the weights: writers that bypass the counter - the trainer's fused SGD kernel, `.data`
    assignments, raw pointers - must call this before the model is used through conv_nhwc again.  Returns the number
    of entries dropped."""
    n = 0
    for m in model.modules():
        for sp in _specs_of_module(m):
            n += len(sp.__dict__.get("_packed", {}))
            sp.__dict__.pop("_packed", None)
    return n


def packed_weights(model):
    """The packed-weight tensors currently cached on a model's convolutions (a captured graph that reads them keeps this list alive)."""
    return [hit[1] for m in model.modules() for sp in _specs_of_module(m) for hit in sp.__dict__.get("_packed", {}).values()]


def flush_bn_counters(model, extra=0):
    """Materialise the lazily counted `num_batches_tracked` buffers (kept off the hot path: 330 tiny launches).
    `extra`: steps executed by hipGraph replay (no Python ran for them)."""
    for m in model.modules():
        if getattr(m, "num_batches_tracked", None) is None or not hasattr(m, "_rssf_steps"):
            continue
        k = m._rssf_steps + extra
        if k:
            m.num_batches_tracked += k
            m._rssf_steps = 0


def run_sequential(seq, x, res_pre=None, act_last=None, grad_accum=None, _chain=None):
    """Execute an nn.Sequential of the reference's shape (Conv2d, BatchNorm2d[, ReLU][, Upsample] or nested
    Sequentials thereof) through the fused ops.  res_pre / act_last: the sequence must END in a Conv2d + BatchNorm2d pair
    (possibly inside a nested Sequential), which then computes act_last(bn(conv(.)) + res_pre) in its own epilogue - the
    `fuse(x0) + low` -> ReLU of a HighResolutionModule output without separate add / clamp launches.
    grad_accum: GradAccum of the sequence's INPUT (handed to its first convolution)."""
    mods = list(seq)
    tail = res_pre is not None or act_last is not None
    i = 0
    chain = _chain if _chain is not None else [None]       # [BnBwdLink of the layer that produced x]: inside a sequence every
    while i < len(mods):                                     # activation has ONE consumer, the next convolution
        m = mods[i]
        if isinstance(m, nn.Sequential):
            last = tail and i + 1 == len(mods)
            x = run_sequential(m, x, res_pre if last else None, act_last if last else None, grad_accum=grad_accum, _chain=chain)
            grad_accum = None
            tail = tail and not last
            i += 1
        elif isinstance(m, nn.Conv2d):
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.modules.batchnorm._BatchNorm):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                link = bwd_stats_link()
                if tail and not relu and i + 2 == len(mods):
                    x = conv_bn_act(x, m, mods[i + 1], ACT_NONE if act_last is None else act_last, res_pre=res_pre, grad_accum=grad_accum,
                                    stats_out=link, stats_in=chain[0])
                    tail = False
                else:
                    x = conv_bn_act(x, m, mods[i + 1], ACT_RELU if relu else ACT_NONE, grad_accum=grad_accum, stats_out=link, stats_in=chain[0])
                chain[0] = link
                grad_accum = None
                i += 3 if relu else 2
            else:
                x = conv_bias(x, m)
                chain[0] = None
                i += 1
        elif isinstance(m, nn.Upsample) and m.mode == "nearest":
            x = upsample_nearest_add(None, x, int(m.scale_factor))
            chain[0] = None
            i += 1
        elif isinstance(m, (nn.Upsample, nn.UpsamplingBilinear2d)) and m.mode == "bilinear" and m.align_corners:
            s = m.scale_factor
            x = upsample_bilinear(x, (int(x.shape[2] * s), int(x.shape[3] * s)))
            chain[0] = None
            i += 1
        else:
            raise NotImplementedError("run_sequential: no HIP kernel for %s on the RSSFormer path" % type(m).__name__)
    if tail:
        raise NotImplementedError("run_sequential: res_pre / act_last need a sequence that ends in Conv2d + BatchNorm2d")
    if grad_accum is not None:
        raise NotImplementedError("run_sequential: grad_accum needs a sequence that starts with Conv2d + BatchNorm2d")
    return x
