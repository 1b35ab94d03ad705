"""Data-parallel training step for RSSFormer (the slice of the external `ever` `th_amp_ddp` trainer the path needs:
train.py:79 + configs/base/loveda.py:68-113 of the reference): bf16 compute with fp32 master weights, sum of
`*loss` entries, gradient all-reduce over RCCL overlapped with backward, global-norm clip (35) + SGD(0.9, wd 1e-4)
with poly LR.  Trainer internals of `ever` are un-vendored => "parity unpinned" (SURVEY.md §8c); the semantics
implemented here are torch.optim.SGD / clip_grad_norm_ / DDP-mean-of-shard-gradients.

MI355X-first layout: all parameters live in ONE flat fp32 buffer (and one flat gradient buffer + one momentum
buffer), so gradient exchange is a handful of large flat buckets (xGMI is per-link bound: few big collectives),
the clip norm is one reduction and the optimizer is one fused HIP launch (csrc/optim.hip).
"""
import math
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import nnf, ops, rccl


class FlatParams:
    """Re-seats every parameter (and its .grad) of `model` as a view into flat fp32 buffers."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + (s + 3) // 4 * 4)     # 16-byte aligned slices
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
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-seat the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class GradBuckets:
    """Flat gradient buckets all-reduced (sum) as soon as every parameter in them has its gradient, i.e. in
    reverse-registration (= roughly reverse-autograd) order on the communicator's own stream."""

    def __init__(self, flat, nbuckets=6, group=None):
        self.flat, self.group = flat, group
        n = flat.numel
        target = max(1, n // nbuckets)
        self.bounds = []          # (start, end) over the flat buffer, built from the END (last layers first)
        self.members = []         # parameter indices per bucket
        end, cur = n, []
        for i in range(len(flat.params) - 1, -1, -1):
            cur.append(i)
            if end - flat.offsets[i] >= target or i == 0:
                self.bounds.append((flat.offsets[i], end))
                self.members.append(cur)
                end, cur = flat.offsets[i], []
        self.bucket_of = {}
        for b, mem in enumerate(self.members):
            for i in mem:
                self.bucket_of[i] = b
        self.pending = [len(m) for m in self.members]
        self.handles = []
        self.launched = [False] * len(self.members)
        self.index_of = {id(p): i for i, p in enumerate(flat.params)}
        for i, p in enumerate(flat.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def param_ready(self, p):
        """Called by the fused backward nodes that accumulate straight into p.grad (no AccumulateGrad hook fires)."""
        i = self.index_of.get(id(p))
        if i is not None:
            self._make_hook(i)(p)

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
        self.handles.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin(self):
        self.pending = [len(m) for m in self.members]
        self.launched = [False] * len(self.members)
        self.handles = []

    def finish(self):
        """Launch buckets whose parameters never received a gradient (headaux: the aux head gets none), then wait."""
        for b in range(len(self.members)):
            self._launch(b)
        for h in self.handles:
            h.wait()


def flush_bn_counters(trainer):
    """num_batches_tracked += steps taken (eager steps counted per layer, graph replays counted once per replay)."""
    nnf.flush_bn_counters(trainer.model, extra=trainer._replayed - trainer._replayed_flushed)
    trainer._replayed_flushed = trainer._replayed


def poly_lr(base_lr, power, max_iters, it):
    return base_lr * (1.0 - min(it, max_iters - 1) / max_iters) ** power     # configs/base/loveda.py:93-99


class Trainer:
    def __init__(self, model, base_lr=0.01, momentum=0.9, weight_decay=1e-4, max_norm=35.0, power=0.9, max_iters=30000,
                 bf16=True, sync_bn=True, nbuckets=6, use_graph=True):
        self._replayed = 0
        self._replayed_flushed = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # RSSF_FORCE_DP=1 exercises the collective plumbing (buckets, SyncBN all-reduces) even on one rank (tests)
        dp = self.world > 1 or (dist.is_initialized() and os.environ.get("RSSF_FORCE_DP") == "1")
        self.model = model
        self.flat = FlatParams(model)
        # Data path of DP: a direct RCCL communicator when possible (collectives on the compute stream, so the whole step
        # can still be one hipGraph; the 128 MB flat gradient is ONE all-reduce after backward, ~1 ms over xGMI), else
        # torch.distributed with bucketed, hook-driven all-reduces overlapped with backward.
        self.comm = rccl.init() if dp else None
        self.buckets = GradBuckets(self.flat, nbuckets) if (dp and self.comm is None) else None
        nnf.set_sync_bn(sync_bn and dp, force=dp and self.world == 1)
        nnf.set_direct_grad(True, self.buckets.param_ready if self.buckets is not None else None)
        self.hp = dict(base_lr=base_lr, momentum=momentum, wd=weight_decay, max_norm=max_norm, power=power, max_iters=max_iters)
        self.bf16 = bf16
        self.it = 0
        dev = self.flat.flat.device
        self.sqnorm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.lr_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        # Whole-step hipGraph: the step is ~3 k short launches and the Python/launch overhead (~65 ms) exceeds the GPU
        # time, so after `graph_warmup` eager steps the step (fwd + loss + bwd + clip + SGD) is captured once and
        # replayed.  In DP this needs the direct RCCL communicator (collectives issued through torch.distributed cannot be
        # captured: its watchdog thread polls their events).  RSSF_GRAPH=1 forces graphs, RSSF_GRAPH=0 disables them.
        env = os.environ.get("RSSF_GRAPH")
        self.use_graph = (env == "1") or (env != "0" and use_graph and (not dp or self.comm is not None))
        self.graph_warmup = 3
        # HRNet branches on side streams: parallel branches of the captured graph (+1 % at B=16); eager launches are host-bound
        # and DP drives one RCCL communicator, so both keep a single stream
        nnf.set_branch_streams(self.use_graph and not dp and os.environ.get("RSSF_BRANCH_STREAMS", "1") != "0")
        self.pack_plan = nnf.PackPlan() if os.environ.get("RSSF_PACK_PLAN", "1") != "0" else None
        self.graph = None
        self._static = None
        self._side = None

    def _eager_step(self, img, target):
        if not self.model.training:          # Module.train() walks all 1 300 sub-modules: not once per step
            self.model.train()
        self.flat.zero_grad()
        if self.buckets is not None:
            self.buckets.begin()
        # one zero-fill and one weight re-pack for the whole step (nnf.ZeroPool / nnf.PackPlan)
        nnf.step_begin(self.flat.flat.device, self.pack_plan)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
                out = self.model(img, target)
            loss = sum(v for k, v in out.items() if k.endswith("loss"))
            loss.backward()
        finally:
            nnf.step_end()
        if self.buckets is not None:
            self.buckets.finish()
        elif self.comm is not None:
            self.comm.all_reduce_(self.flat.grad)
        hp = self.hp
        ops.grad_sqnorm(self.flat.grad, self.sqnorm)
        ops.sgd_step_(self.flat.flat, self.flat.grad, self.flat.mom, self.sqnorm, 1.0 / self.world, hp["max_norm"], 0.0,
                      hp["momentum"], hp["wd"], False, lr_dev=self.lr_dev)
        return loss.detach()

    def _capture(self, img, target):
        self._static = (img.clone(), {k: v.clone() for k, v in target.items()})
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            # thread_local: the NCCL watchdog thread keeps polling events of earlier collectives, which a global-mode
            # capture forbids ("operation not permitted when stream is capturing")
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._static_loss = self._eager_step(*self._static)
        except Exception as e:       # fall back to eager launches, loudly
            print("[rssf] hipGraph capture failed (%s: %s); continuing with eager launches" % (type(e).__name__, e), flush=True)
            self.use_graph = False
            torch.cuda.synchronize()
            return False
        self.graph = g
        return True

    def step(self, img, target):
        """One optimisation step; returns the (detached, on-device) loss."""
        hp = self.hp
        self.lr_dev.fill_(poly_lr(hp["base_lr"], hp["power"], hp["max_iters"], self.it))
        if self.use_graph and self.it >= self.graph_warmup:
            if self.graph is None and not self._capture(img, target):
                loss = self._eager_step(img, target)
            else:
                s_img, s_tgt = self._static
                if img.data_ptr() != s_img.data_ptr():
                    s_img.copy_(img)
                for k, v in target.items():
                    if v.data_ptr() != s_tgt[k].data_ptr():
                        s_tgt[k].copy_(v)
                self.graph.replay()
                self._replayed += 1
                loss = self._static_loss
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
        return loss


def init_distributed():
    """One process per GPU; rendezvous from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
    return rank, local, world
