"""CPU, world_size 2, gloo: the data-parallel plumbing — flat parameter/gradient buffers, bucketed all-reduce fired
from post-accumulate hooks (incl. a parameter that never receives a gradient, like `headaux`), mean-of-shards result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _DirectLinear(torch.autograd.Function):
    """x @ w.T whose backward ACCUMULATES the weight gradient straight into w.grad and returns None for it - what every
    librssf backward node does under the trainer's flat gradient buffer (representationlearning_amd.nnf.grad_target)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        w.grad += g.t() @ x
        return g @ w, None


class Net(nn.Module):
    def __init__(self, direct=False):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 16)
        self.c = nn.Linear(16, 4)
        self.d = nn.Parameter(torch.randn(4, 4) * 0.3)      # used through the direct-accumulation node when `direct`
        self.unused = nn.Linear(3, 7)          # never in the graph (cf. headaux: SURVEY §2.2)
        self.direct = direct

    def forward(self, x):
        h = self.c(torch.relu(self.b(torch.relu(self.a(x)))))
        # the parameter is used TWICE: its bucket may only go out after the second use has run backward
        return _DirectLinear.apply(_DirectLinear.apply(h, self.d), self.d) if self.direct else (h @ self.d.t()) @ self.d.t()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from representationlearning_amd.trainer import FlatParams, GradBuckets
    from representationlearning_amd.rccl import TorchComm
    net = Net(direct=True)
    flat = FlatParams(net)
    buckets = GradBuckets(flat, TorchComm(), nbuckets=3)
    assert len(buckets.bounds) >= 2 and buckets.bounds[0][1] == flat.numel and buckets.bounds[-1][0] == 0
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 8)
    for step in range(2):                       # two steps: state re-arms correctly
        flat.zero_grad()
        buckets.begin()
        net(x).square().mean().backward()
        buckets.finish()
    g = flat.grad.clone() / world
    # parameters are still views of the flat buffer and the model output is unchanged by re-seating
    assert all(p.data_ptr() == flat.flat.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    q.put((rank, g.numpy().copy(), x.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_mean_of_shard_gradients():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # independent single-process emulation: mean of the two shard gradients
    grads = []
    for _, _, x in res:
        net = Net()
        net(torch.from_numpy(x)).square().mean().backward()
        grads.append([torch.zeros_like(p) if p.grad is None else p.grad for p in net.parameters()])
    from representationlearning_amd.trainer import FlatParams
    ref = FlatParams(Net())
    for i, (p, o) in enumerate(zip(ref.params, ref.offsets)):
        want = (grads[0][i] + grads[1][i]) / 2
        for _, g, _ in res:
            torch.testing.assert_close(torch.from_numpy(g)[o:o + p.numel()].view_as(p), want, rtol=1e-6, atol=1e-7)
