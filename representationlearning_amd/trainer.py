"""Data-parallel training step for RSSFormer (the slice of the external `ever` `th_amp_ddp` trainer the path needs:
train.py:79 + configs/base/loveda.py:68-113 of the reference): bf16 compute with fp32 master weights, sum of
`*loss` entries, gradient all-reduce over RCCL overlapped with backward, global-norm clip (35) + SGD(0.9, wd 1e-4)
with poly LR.  Trainer internals of `ever` are un-vendored => "parity unpinned" (SURVEY.md §8c); the semantics
implemented here are torch.optim.SGD / clip_grad_norm_ / DDP-mean-of-shard-gradients / nn.SyncBatchNorm.

MI355X-first layout: all parameters live in ONE flat fp32 buffer (and one flat gradient buffer + one momentum
buffer), so gradient exchange is a handful of large flat buckets (xGMI is per-link bound: few big collectives),
the clip norm is one reduction and the optimizer is one fused HIP launch (csrc/optim.hip).
"""
import os
import weakref

import torch
import torch.distributed as dist

from . import nnf, ops, rccl


class FlatParams:
    """Re-seats every parameter (and its .grad) of `model` as a view into flat fp32 buffers."""

    def __init__(self, model):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        names = [n for n, _ in named]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        # 16-byte aligned slices - with ONE exception, by name: the two 7x7 SpatialAttention kernels of a transformer block
        # (`...atrous_block1.conv1.weight` followed by `...atrous_block2.conv1.weight`, 98 floats each,
        # multihead_isa_pool_attention.py:30-31) sit WITHOUT a gap: they are then ONE [2, 2, 7, 7] operand of the gate kernels
        # (autograd._gate_kernels: a view, not a stack copy per block and step) and the pair ends on a 16-byte boundary again.
        # `self.pairs` holds the index of each pair's first parameter: ranges_of() never starts a range on the second one alone.
        self.pairs = set()
        for i, s in enumerate(sizes):
            tight = (names[i].endswith("atrous_block1.conv1.weight") and i + 1 < len(sizes)
                     and names[i + 1] == names[i].replace("atrous_block1", "atrous_block2") and self.params[i + 1].shape == self.params[i].shape
                     and s % 4 != 0 and (2 * s) % 4 == 0 and self.offsets[-1] % 4 == 0)
            if tight:
                self.pairs.add(i)
            self.offsets.append(self.offsets[-1] + s if tight else (self.offsets[-1] + s + 3) // 4 * 4)
        n = self.offsets[-1]
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.mom = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            v = self.flat[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.numel = n

    def zero_grad(self):
        ops.zero_(self.grad)
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-seat the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def ranges_of(self, keep):
        """Merged [start, end) ranges of the flat buffer covered by the parameters whose index is in `keep`."""
        keep = set(keep)
        for i in self.pairs:                 # a tightly packed pair goes in or out as a whole: every range starts 16-byte aligned
            if (i in keep) != (i + 1 in keep):
                # (widening the range silently would hand a gradient-free parameter weight decay and momentum, which torch.optim.SGD skips)
                raise RuntimeError("FlatParams.ranges_of: parameters %d / %d are packed as one aligned pair but only one of them is live" % (i, i + 1))
        out = []
        for i in sorted(keep):
            s, e = self.offsets[i], self.offsets[i + 1]
            if out and out[-1][1] == s:
                out[-1][1] = e
            else:
                out.append([s, e])
        return [tuple(r) for r in out]


class GradBuckets:
    """Flat gradient buckets all-reduced (sum) as soon as every parameter in them has its gradient, i.e. in
    reverse-registration (= roughly reverse-autograd: head/neck first, stem last) order, overlapped with the rest of backward:
    * direct RCCL communicator: on a side stream forked from the compute stream at the moment the bucket completes and
      joined before the clip (inside a captured step these are parallel branches of the hipGraph),
    * torch.distributed: async_op work handles on the process group's own stream."""

    def __init__(self, flat, comm, nbuckets=6, before_launch=None, side_streams=None):
        self.flat, self.comm = flat, comm
        self.side_streams = side_streams        # callable -> the streams (besides the compute stream) gradients are produced on
        # called on the compute stream right before a bucket's collective is enqueued: the trainer flushes the deferred split-K
        # reductions of the weight gradients (nnf.WgradPlan) there, so that the bucket's gradients are final
        self.before_launch = before_launch
        n = flat.numel
        target = max(1, n // nbuckets)
        self.bounds = []          # (start, end) over the flat buffer, built from the END (last layers first)
        self.members = []         # parameter indices per bucket
        end, cur = n, []
        for i in range(len(flat.params) - 1, -1, -1):
            cur.append(i)
            # (a bucket never starts at the second kernel of a tightly packed pair: bucket slices stay 16-byte aligned)
            if (end - flat.offsets[i] >= target and flat.offsets[i] % 4 == 0) or i == 0:
                self.bounds.append((flat.offsets[i], end))
                self.members.append(cur)
                end, cur = flat.offsets[i], []
        self.bucket_of = {}
        for b, mem in enumerate(self.members):
            for i in mem:
                self.bucket_of[i] = b
        self.live = [len(m) for m in self.members]      # parameters per bucket that DO receive a gradient (see set_live)
        self.pending = list(self.live)
        self.handles = []
        self.launched = [False] * len(self.members)
        self.index_of = {id(p): i for i, p in enumerate(flat.params)}
        self.side = torch.cuda.Stream() if comm.direct else None
        self.used = set()
        self.compute_stream = None
        # The post-accumulate-grad hook fires once per parameter and backward pass, after the LAST node that uses the parameter
        # has run - also when that node returned None because its kernels accumulated straight into p.grad (nnf's direct
        # gradient accumulation): AccumulateGrad is scheduled for undefined gradients too (tests/test_dp_gloo.py pins this).
        # So the hook alone is the "gradient complete, kernels enqueued" signal; signalling by hand as well counted every
        # parameter twice and launched buckets half-way (caught by tests/test_gpu_dp.py at world 2).
        self.hook_handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(flat.params)]

    def close(self):
        """Remove the hooks (parameters -> hooks -> this object -> parameters is a reference cycle: left alone, the trainer and its
        captured hipGraph would be released by the cyclic collector at an arbitrary later moment - possibly in the middle of another
        trainer's graph capture, which crashed the capture / replay of that graph)."""
        for h in self.hook_handles:
            h.remove()
        self.hook_handles = []
        self.before_launch = None

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e = self.bounds[b]
        g = self.flat.grad[s:e]
        # The kernels that produced this bucket's gradients are already enqueued on the step's COMPUTE stream (recorded by
        # begin()).  Not torch.cuda.current_stream(): this runs inside a post-accumulate-grad hook, and autograd executes an
        # AccumulateGrad node on the stream its parameter was first used on (the warm-up stream, the default stream, ...) - ordering
        # the collective after THAT stream leaves it unordered against the backward kernels: a captured step then carries a race
        # that shows up as NaN losses as soon as the replay starts on an idle GPU (tools/dp_nan_probe.py).
        cs = self.compute_stream if (self.compute_stream is not None or not g.is_cuda) else torch.cuda.current_stream()
        if os.environ.get("RSSF_TEST_FAIL_CAPTURE") == "1" and g.is_cuda and b == len(self.members) // 2:
            with torch.cuda.stream(cs):                 # test hook: an error in the middle of a CAPTURED backward (side streams forked)
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("injected failure inside the captured step (RSSF_TEST_FAIL_CAPTURE)")
        # Gradients of the HRNet branches / fuse paths are produced on SIDE streams (nnf.parallel_map / fork_side): the step's
        # stream joins them here, so that "after the compute stream" really means "after every kernel that wrote this bucket".
        if g.is_cuda and self.side_streams is not None:
            capturing = torch.cuda.is_current_stream_capturing() if cs == torch.cuda.current_stream() else None
            for s in self.side_streams():
                if capturing is None:
                    with torch.cuda.stream(cs):
                        capturing = torch.cuda.is_current_stream_capturing()
                with torch.cuda.stream(s):
                    s_cap = torch.cuda.is_current_stream_capturing()
                if s_cap == capturing:             # (a stream that is not part of the running capture holds no work of this step)
                    cs.wait_stream(s)
        fn = self.before_launch() if self.before_launch is not None else None       # weak reference to the trainer's method
        if fn is not None:
            if g.is_cuda:
                with torch.cuda.stream(cs):
                    fn()
            else:
                fn()
        if self.comm.direct:
            side = self._bucket_stream()
            side.wait_stream(cs)
            with torch.cuda.stream(side):
                self.comm.allreduce_bucket_(g)
            self.used.add(side)
        elif g.is_cuda:
            with torch.cuda.stream(cs):                 # the process group orders its work after the stream current at the call
                self.handles.append(self.comm.allreduce_bucket_async(g))
        else:
            self.handles.append(self.comm.allreduce_bucket_async(g))

    def set_live(self, live_ids):
        """Parameters the loss does not reach (RSSFormer's `headaux`) never fire their hook: counted as pending they kept their
        bucket - bucket 0, head and neck, the first gradients to be ready - waiting until finish(), i.e. without any overlap
        (ADVICE r2).  Called before the first backward with the ids of the reachable leaves."""
        self.live = [sum(1 for i in m if id(self.flat.params[i]) in live_ids) for m in self.members]
        self.pending = list(self.live)

    def _bucket_stream(self):
        """Stream of the bucket all-reduces: the LAST side stream of the step when the step has side streams (its own otherwise).
        A captured step then has four concurrent branches with or without data parallelism; with a fifth one (a stream of their
        own) this ROCm's hipGraph runtime crashed in hip::Graph::UpdateStreams when a LATER graph of the process was launched
        (tests/test_gpu_trainer.py in one process; rocgdb backtrace in DESIGN.md).  The collectives belong to their own
        communicator, so sharing the stream with that side stream's SyncBN exchanges orders nothing wrongly."""
        pool = self.side_streams() if self.side_streams is not None else []
        return pool[-1] if (pool and os.environ.get("RSSF_BUCKET_OWN_STREAM") != "1") else self.side

    def begin(self):
        self.used = set()
        self.compute_stream = torch.cuda.current_stream() if self.flat.grad.is_cuda else None
        self.pending = list(self.live)
        self.launched = [False] * len(self.members)
        self.handles = []

    def finish(self):
        """Launch buckets whose parameters never received a gradient (headaux: the aux head gets none), then wait."""
        for b in range(len(self.members)):
            self._launch(b)
        for h in self.handles:
            h.wait()
        for side in self.used:
            (self.compute_stream or torch.cuda.current_stream()).wait_stream(side)
        self.used = set()


def flush_bn_counters(trainer):
    """num_batches_tracked += steps taken (eager steps counted per layer, graph replays counted once per replay)."""
    nnf.flush_bn_counters(trainer.model, extra=trainer._replayed - trainer._replayed_flushed)
    trainer._replayed_flushed = trainer._replayed


def poly_lr(base_lr, power, max_iters, it):
    return base_lr * (1.0 - min(it, max_iters - 1) / max_iters) ** power     # configs/base/loveda.py:93-99


def reachable_parameters(loss):
    """ids of the leaf tensors the autograd graph of `loss` reaches - the parameters torch.optim.SGD would see a gradient
    for (`p.grad is not None`).  RSSFormer's `headaux` is not among them: the loss uses its output under no_grad only
    (module/CGFL.py:75-97)."""
    seen, stack, leaves = set(), [loss.grad_fn], set()
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        v = getattr(fn, "variable", None)
        if v is not None:
            leaves.add(id(v))
        stack.extend(f for f, _ in fn.next_functions)
    return leaves


class Trainer:
    def __init__(self, model, base_lr=0.01, momentum=0.9, weight_decay=1e-4, max_norm=35.0, power=0.9, max_iters=30000,
                 bf16=True, sync_bn=True, nbuckets=6, use_graph=True, deterministic=None):
        self._replayed = 0
        self._replayed_flushed = 0
        self._steps = 0                  # steps taken by THIS object (`it` may start above 0: a resumed run)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # RSSF_FORCE_DP=1 exercises the collective plumbing (buckets, SyncBN exchanges) even on one rank (tests)
        dp = self.world > 1 or (dist.is_initialized() and os.environ.get("RSSF_FORCE_DP") == "1")
        self.model = model
        self.flat = FlatParams(model)
        self.rt = nnf.Runtime()
        if deterministic is not None:
            self.rt.deterministic = bool(deterministic)
        # Data path of DP: direct RCCL communicators when possible (collectives enqueued like kernels, so the whole step can
        # still be one hipGraph): one for the SyncBN exchanges on the compute stream, one for the gradient buckets on a side
        # stream.  Else torch.distributed (eager launches; buckets as async work on the process group's stream).
        # Every stream that issues collectives has its OWN communicator (one communicator's collectives must be issued in one order on
        # every rank, and two streams give none): the SyncBN exchanges of the step's main stream, those of each of the three side
        # streams (HRNet branches 1..3 / the fuse outputs 1.. beside the transformer block: nnf.parallel_map, nnf.fork_side) and the
        # gradient buckets on their stream.
        # The SyncBN exchanges themselves: the peer-to-peer kernel of csrc/p2p.hip (one channel per stream) when its start-up
        # self-test passes on this node, RCCL all-reduces (one communicator per stream) otherwise.
        self.comm = self.grad_comm = self.p2p = None
        self.side_comms, self._rccl = [], []
        if dp:
            # side streams that issue exchanges: with the lock-step walk of the HighResolutionModules (the default) only the fuse
            # outputs 1.. run beside the main stream (nnf.fork_side); the branch-by-branch walk (RSSF_LOCKSTEP=0) uses three
            nside = 0 if os.environ.get("RSSF_DP_SIDE_COMMS", "1") == "0" else (3 if os.environ.get("RSSF_LOCKSTEP", "1") == "0" else 1)
            comms = rccl.create(2)
            if comms is not None:
                self._rccl = list(comms)
                self.comm, self.grad_comm = comms
                self.p2p = rccl.create_p2p(1 + nside, self.grad_comm)
                if self.p2p is not None:
                    self.comm, self.side_comms = self.p2p.channel(0), [self.p2p.channel(1 + k) for k in range(nside)]
                elif nside:
                    more = rccl.create(nside)
                    if more is not None:
                        self._rccl += more
                        self.side_comms = list(more)
            else:
                self.comm, self.grad_comm = rccl.TorchComm(), rccl.TorchComm()
                if os.environ.get("RSSF_SYNCBN") == "p2p" and torch.cuda.is_available():
                    # asked for explicitly: the peer-to-peer exchange does not need RCCL (tests: two ranks on one GPU over gloo)
                    self.p2p = rccl.create_p2p(1 + nside, self.grad_comm)
                    if self.p2p is not None:
                        self.comm, self.side_comms = self.p2p.channel(0), [self.p2p.channel(1 + k) for k in range(nside)]
            self._broadcast_initial_state()
        overlap = os.environ.get("RSSF_GRAD_OVERLAP", "1") != "0"
        # (a WEAK reference to the bound method: a strong one would tie this trainer - and its captured graph - into the reference
        # cycle of the gradient hooks, see GradBuckets.close)
        rt, dev0 = self.rt, self.flat.flat.device            # (not `self`: no reference cycle through the closure)
        self.buckets = GradBuckets(self.flat, self.grad_comm, nbuckets, before_launch=weakref.WeakMethod(self._flush_wgrad),
                                   side_streams=lambda: rt.side_streams.get(dev0, []) if rt.branch_streams else []) \
            if (dp and overlap) else None
        self.rt.comm = self.comm
        self.rt.sync_all_bn = bool(sync_bn and dp)
        self.rt.force_collectives = dp and self.world == 1
        self.rt.direct = True
        self.hp = dict(base_lr=base_lr, momentum=momentum, wd=weight_decay, max_norm=max_norm, power=power, max_iters=max_iters)
        self.bf16 = bf16
        self.it = 0
        dev = self.flat.flat.device
        self.sqnorm = torch.zeros(ops.SQNORM_ELEMS, device=dev, dtype=torch.float32)
        self.lr_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        # torch.optim.SGD skips parameters whose gradient is None (no weight decay, no momentum): the flat update covers only
        # the ranges of parameters the loss reaches, found from the autograd graph of the first step
        self.sgd_ranges = None
        # Whole-step hipGraph: the step is ~3 k short launches and the Python/launch overhead (~65 ms) exceeds the GPU
        # time, so after `graph_warmup` eager steps the step (fwd + loss + bwd + exchange + clip + SGD) is captured once and
        # replayed.  In DP this needs the direct RCCL communicators (collectives issued through torch.distributed cannot be
        # captured: its watchdog thread polls their events).  RSSF_GRAPH=1 forces graphs, RSSF_GRAPH=0 disables them.
        env = os.environ.get("RSSF_GRAPH")
        # (BOTH communicators: the peer-to-peer SyncBN exchange can also run beside torch.distributed gradient buckets - two ranks
        # on one GPU over gloo - and those cannot be captured)
        self.use_graph = (env == "1") or (env != "0" and use_graph and (not dp or (self.comm.direct and self.grad_comm.direct)))
        self.graph_warmup = 3
        # HRNet branches on side streams: parallel branches of the captured graph (+1 % at B=16); eager launches are host-bound
        # and the SyncBN exchanges of a DP step must reach the communicator in one fixed order, so both keep a single stream
        self.rt.branch_streams = self.use_graph and os.environ.get("RSSF_BRANCH_STREAMS", "1") != "0"
        if dp:
            # data parallel with a communicator per side stream: the same stream layout as the single-GPU step.  With ONE communicator
            # for all SyncBN exchanges (torch.distributed, RSSF_DP_SIDE_COMMS=0) the HighResolutionModules walk their branches and fuse
            # paths in lock-step groups on the step's stream instead (one exchange per group, in a fixed order); RSSF_GROUP_STREAMS=1
            # then runs the items of a group on parallel streams around it (measured slower: two fork/joins per group).
            self.rt.stream_comms = self.side_comms if (self.side_comms and self.rt.branch_streams) else None
            self.rt.group_streams = self.rt.branch_streams and self.rt.stream_comms is None and os.environ.get("RSSF_GROUP_STREAMS", "0") == "1"
        self.pack_plan = nnf.PackPlan() if os.environ.get("RSSF_PACK_PLAN", "1") != "0" else None
        # batched split-K reductions instead of ~330 (nnf.WgradPlan): ONE launch at the end of backward, or - with gradient buckets
        # in flight during backward - one per bucket, issued right before the bucket's all-reduce (GradBuckets.before_launch)
        self.wgrad_plan = nnf.WgradPlan() if os.environ.get("RSSF_WGRAD_PLAN", "1") != "0" else None
        self.graph = None
        self._static = None
        self._side = None
        self._seed = None                # d loss / d loss = 1, allocated once

    def _flush_wgrad(self):
        wp = self.rt.wgrad_plan
        if wp is not None:
            wp.flush()

    def _broadcast_initial_state(self):
        """What DistributedDataParallel does at construction: rank 0's parameters and buffers everywhere."""
        if self.world == 1:
            return
        dist.broadcast(self.flat.flat, src=0)
        bufs = [b for b in self.model.buffers() if b.is_floating_point()]
        if bufs:
            packed = torch.cat([b.detach().reshape(-1).float() for b in bufs])
            dist.broadcast(packed, src=0)
            o = 0
            for b in bufs:
                b.copy_(packed[o:o + b.numel()].view_as(b))
                o += b.numel()

    def _eager_step(self, img, target):
        if not self.model.training:          # Module.train() walks all 1 300 sub-modules: not once per step
            self.model.train()
        self.flat.zero_grad()
        if self.buckets is not None:
            self.buckets.begin()
        with nnf.use(self.rt):
            # one zero-fill and one weight re-pack for the whole step (nnf.ZeroPool / nnf.PackPlan)
            nnf.step_begin(self.flat.flat.device, self.pack_plan, wgrad_plan=self.wgrad_plan)
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
                    out = self.model(img, target)
                terms = [v for k, v in out.items() if k.endswith("loss")]
                loss = terms[0] if len(terms) == 1 else sum(terms[1:], terms[0])      # (Python's sum() starts at 0: 0 + loss was a launch)
                if self.sgd_ranges is None:
                    live = reachable_parameters(loss)
                    self.sgd_ranges = self.flat.ranges_of([i for i, p in enumerate(self.flat.params) if id(p) in live])
                    if self.buckets is not None:
                        self.buckets.set_live(live)
                if self._seed is None or self._seed.dtype != loss.dtype or self._seed.device != loss.device:
                    self._seed = torch.ones((), device=loss.device, dtype=loss.dtype)
                loss.backward(self._seed)                 # the seed gradient is a constant of the trainer, not a fill per step
                if self.rt.branch_streams:
                    # the kernels add parameter gradients straight into the flat buffer (no AccumulateGrad node, so autograd's
                    # end-of-backward join of "leaf streams" knows nothing of the side streams): join them here, before the
                    # deferred reductions, the norm and the update read that buffer.  Inside a capture these are graph edges.
                    cur = torch.cuda.current_stream()
                    for st in self.rt.side_streams.get(self.flat.flat.device, []):
                        cur.wait_stream(st)
            finally:
                nnf.step_end()
        if self.buckets is not None:
            self.buckets.finish()
        elif self.grad_comm is not None:
            self.grad_comm.allreduce_bucket_(self.flat.grad)
        hp = self.hp
        ops.grad_sqnorm(self.flat.grad, self.sqnorm)       # unreached parameters hold zeros: same norm as clip_grad_norm_
        for s, e in self.sgd_ranges:
            ops.sgd_step_(self.flat.flat[s:e], self.flat.grad[s:e], self.flat.mom[s:e], self.sqnorm, 1.0 / self.world, hp["max_norm"],
                          0.0, hp["momentum"], hp["wd"], False, lr_dev=self.lr_dev)
        return loss.detach()

    def _capture(self, img, target):
        self._static = (img.clone(), {k: v.clone() for k, v in target.items()})
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # Python runs the step once while capturing (no kernel executes): undo its num_batches_tracked bookkeeping, the
        # replay that follows is what counts
        counters = [(m, m._rssf_steps) for m in self.model.modules() if hasattr(m, "_rssf_steps")]
        try:
            # thread_local: the NCCL watchdog thread keeps polling events of earlier collectives, which a global-mode
            # capture forbids ("operation not permitted when stream is capturing")
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                try:
                    self._static_loss = self._eager_step(*self._static)
                finally:
                    # An exception in the middle of the step (a collective that cannot be captured, an allocation failure, ...)
                    # leaves the side streams it had forked into INSIDE the capture: ending the capture then fails with
                    # hipErrorStreamCaptureUnjoined and, on this ROCm, those streams STAY in capture mode - the eager steps that
                    # follow die on their first synchronising call ("operation not permitted when stream is capturing";
                    # found with bench.py on two ranks over gloo).  Join every stream of this step that is still capturing
                    # back into the capture's stream, so that the capture ends cleanly whatever happened.
                    self._join_capturing_streams()
        except Exception as e:       # fall back to eager launches, loudly
            print("[rssf] hipGraph capture failed (%s: %s); continuing with eager launches" % (type(e).__name__, e), flush=True)
            self.use_graph = False
            self.rt.branch_streams = False       # (eager launches are host-bound: one stream, as without graphs)
            self._static = None
            torch.cuda.synchronize()
            return False
        finally:
            for m, k in counters:
                m._rssf_steps = k
        self.graph = g
        return True

    def _agree_on_capture(self, ok):
        """Data parallel: the captured and the eager step issue DIFFERENT exchange sequences (the eager fall-back gives up the side
        stream and with it that stream's communicator), so a capture that failed on one rank only - an allocation failure, say -
        would leave the ranks in different collectives: all ranks keep the graph, or none does (MIN over the ranks, outside the
        capture)."""
        if self.world <= 1 or not dist.is_initialized():
            return ok
        dev = self.flat.flat.device if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return True
        if ok:
            print("[rssf] another rank failed to capture the step; all ranks continue with eager launches", flush=True)
            self.graph = self._static = None
            self.use_graph = False
            self.rt.branch_streams = False
            torch.cuda.synchronize()
        return False

    def _join_capturing_streams(self):
        cur = torch.cuda.current_stream()
        if not torch.cuda.is_current_stream_capturing():
            return
        pool = [st for lst in self.rt.side_streams.values() for st in lst]
        if self.buckets is not None and getattr(self.buckets, "side", None) is not None:
            pool.append(self.buckets.side)
        for st in pool:
            if st == cur:
                continue
            with torch.cuda.stream(st):
                capturing = torch.cuda.is_current_stream_capturing()
            if capturing:
                cur.wait_stream(st)

    def _fits_static(self, img, target):
        s_img, s_tgt = self._static
        return (img.shape == s_img.shape and img.dtype == s_img.dtype and set(target) == set(s_tgt)
                and all(v.shape == s_tgt[k].shape and v.dtype == s_tgt[k].dtype for k, v in target.items()))

    def step(self, img, target):
        """One optimisation step; returns the (detached, on-device) loss."""
        hp = self.hp
        self.lr_dev.fill_(poly_lr(hp["base_lr"], hp["power"], hp["max_iters"], self.it))
        if self.use_graph and self._steps >= self.graph_warmup:
            if self.graph is None and not self._agree_on_capture(self._capture(img, target)):
                loss = self._eager_step(img, target)
            elif not self._fits_static(img, target):
                loss = self._eager_step(img, target)       # e.g. a last partial batch: the captured step has fixed shapes
            else:
                s_img, s_tgt = self._static
                if img.data_ptr() != s_img.data_ptr():
                    s_img.copy_(img)
                for k, v in target.items():
                    if v.data_ptr() != s_tgt[k].data_ptr():
                        s_tgt[k].copy_(v)
                self.graph.replay()
                self._replayed += 1
                loss = self._static_loss.clone()           # the static tensor is overwritten by the next replay
        elif self.use_graph:
            # warm-up steps run on a side stream, as hipGraph capture requires (the autograd threads and lazily
            # initialised library state must have seen a non-default stream before the capture starts)
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                loss = self._eager_step(img, target)
            torch.cuda.current_stream().wait_stream(self._side)
        else:
            loss = self._eager_step(img, target)
        self.it += 1
        self._steps += 1
        return loss

    def state_dict(self):
        """What a resumed run needs besides the model's own state_dict: the step counter (poly LR) and the momentum buffer, keyed
        by parameter name so that it survives a different flat layout."""
        names = {id(p): k for k, p in self.model.named_parameters()}
        mom = {}
        for p, o in zip(self.flat.params, self.flat.offsets):
            mom[names[id(p)]] = self.flat.mom[o:o + p.numel()].view_as(p).detach().cpu().clone()
        return dict(it=self.it, momentum=mom)

    def load_state_dict(self, sd):
        names = {id(p): k for k, p in self.model.named_parameters()}
        for p, o in zip(self.flat.params, self.flat.offsets):
            m = sd["momentum"].get(names[id(p)])
            if m is not None:
                self.flat.mom[o:o + p.numel()].view_as(p).copy_(m)
        self.it = int(sd["it"])

    def check_exchange(self):
        """Raise if a peer-to-peer SyncBN exchange ever ran into its bounded spin (RSSF_P2P_TIMEOUT_MS: a debugging bound - in
        production the variable stays UNSET and the spin is unbounded).  A rank that waited out the bound added a stale word to
        its BatchNorm totals: the replicas have diverged and the run must stop, loudly.  Call where the host
        synchronises anyway (train.py: the log interval's float(loss)); the status word is one 4-byte device read."""
        if self.p2p is None:
            return
        late = self.p2p.timed_out()
        if late:
            raise RuntimeError("peer-to-peer SyncBN exchange: rank %d waited out RSSF_P2P_TIMEOUT_MS for rank %d - its BatchNorm "
                               "statistics are stale and the replicas have diverged; stopping" % (dist.get_rank() if dist.is_initialized() else 0, late - 1))

    def close(self):
        """Destroy the RCCL communicators (before torch.distributed's process group goes away) and release the captured step NOW,
        not whenever the garbage collector gets to it."""
        if self.buckets is not None:
            self.buckets.close()
        self.graph = self._static = self._static_loss = None
        if self.p2p is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.p2p.destroy()
        for c in {id(c): c for c in [self.comm, self.grad_comm] + list(self.side_comms) + list(self._rccl) if c is not None}.values():
            c.destroy()
        self.comm = self.grad_comm = self.p2p = self.rt.comm = self.rt.stream_comms = None
        self.side_comms, self._rccl = [], []


def init_distributed():
    """One process per GPU; rendezvous from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RSSF_ONE_GPU=1 (test hook): every rank on device 0 over gloo - the multi-process plumbing of bench.py / train.py on a one-GPU
    # box (RCCL refuses two ranks on one device; the SyncBN exchange is then torch.distributed's or, with RSSF_SYNCBN=p2p, the
    # peer-to-peer kernel).  Says nothing about speed.
    one_gpu = os.environ.get("RSSF_ONE_GPU") == "1"
    if torch.cuda.is_available():
        torch.cuda.set_device(0 if one_gpu else local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() and not one_gpu else "gloo", rank=rank, world_size=world)
    return rank, local, world


def shutdown_distributed(trainer=None):
    if trainer is not None:
        trainer.close()
    if dist.is_initialized():
        dist.destroy_process_group()
