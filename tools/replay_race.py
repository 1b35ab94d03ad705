"""Replays one captured training step many times from an unchanged state (lr = 0, deterministic statistics) and reports what is
not bit-identical between replays: forward module outputs, every gradient tensor in backward execution order, and the inputs /
outputs of the gate backward inside the attention node.  A replayed hipGraph runs its branches truly concurrently, so this is
where an intra-kernel problem that needs contention shows up (found with it: DESIGN.md lesson 23).

  python tools/replay_race.py [replays=60] [batch=2] [size=128]      (GPU box, from the repository root; RSSF_FORK_FUSE etc. are honoured;
  ALT=1: every compared replay follows a replay on a different batch - stale reads across streams become visible)

Nothing in the package is modified: the hooks are installed from here (autograd pre-hooks through a wrapped Tensor.backward,
global module forward hooks, a wrapped ops.gate_weights_bwd)."""
import os, sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
os.environ.setdefault("RSSF_BRANCH_STREAMS", "1")
from representationlearning_amd.trainer import Trainer
from representationlearning_amd.configs import synthetic_batch
from representationlearning_amd import ops
from test_gpu_trainer import _mk

R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 128
grads, fwd, gate = [], [], []            # (name, tensor) lists; tensors of the captured step are static: read them after a replay

_backward = torch.Tensor.backward
def backward(self, *a, **k):
    del grads[:]
    seen, stack = set(), [self.grad_fn]
    while stack:
        nd = stack.pop()
        if nd is None or nd in seen:
            continue
        seen.add(nd)
        nd.register_prehook(lambda gs, name=type(nd).__name__: grads.extend((name, g) for g in gs if g is not None) or None)
        stack.extend(f for f, _ in nd.next_functions)
    return _backward(self, *a, **k)
torch.Tensor.backward = backward

def fwd_hook(mod, args, out):
    outs = out if isinstance(out, (list, tuple)) else [out]
    fwd.extend((type(mod).__name__, o) for o in outs if isinstance(o, torch.Tensor) and o.is_cuda)
torch.nn.modules.module.register_module_forward_hook(fwd_hook)

_gate_bwd = ops.gate_weights_bwd
def gate_bwd(domega, *a, **kw):
    out = _gate_bwd(domega, *a, **kw)
    gate.append(("gate.domega", domega.clone()))
    gate.append(("gate.dpooled", out.clone()))
    return out
ops.gate_weights_bwd = gate_bwd

img, lab = synthetic_batch(BATCH, SIZE, seed=5)
t = Trainer(_mk(6), bf16=True, base_lr=0.0, use_graph=True, deterministic=True)
while t.graph is None or t._replayed < 1:
    if t.graph is None:
        del fwd[:], gate[:]
    t.step(img, dict(cls=lab))
# the parameter gradients the kernels add straight into the trainer's flat buffer (no autograd tensor carries them)
names = {id(p): k for k, p in t.model.named_parameters()}
pgrad = [(names[id(p)], t.flat.grad[o:o + p.numel()]) for p, o in zip(t.flat.params, t.flat.offsets)]
groups = (("forward", fwd), ("gate", gate), ("grad", grads), ("param", pgrad))
print("captured:", ", ".join("%d %s tensors" % (len(g), n) for n, g in groups), flush=True)

ALT = os.environ.get("ALT", "0") == "1"        # a replay on ANOTHER batch before every compared one: with lr = 0 a kernel that reads a
img2, lab2 = synthetic_batch(BATCH, SIZE, seed=11)   # buffer BEFORE this replay's producer wrote it finds the previous replay's identical
                                                 # values and goes unseen - unless the previous replay worked on different data
def replay():
    if ALT:
        t.step(img2, dict(cls=lab2))
        torch.cuda.synchronize()
    t.step(img, dict(cls=lab))
    torch.cuda.synchronize()

replay()
ref = [[x.clone() for _, x in g] for _, g in groups]
# per group and replay: (index, name, shape, differing elements, max |diff|, max |ref|, first index) of every tensor that differs
found = [[] for _ in groups]
for r in range(R):
    replay()
    for gi, (gname, g) in enumerate(groups):
        d = []
        for i, ((name, x), a) in enumerate(zip(g, ref[gi])):
            if not torch.equal(x, a):
                if gname == "param" and (x - a).abs().max().item() <= 1e-5 * a.abs().max().item():
                    continue                   # (fp32 atomics of the transformer's parameter gradients: order noise, not what is looked for)
                ne = x != a
                d.append((i, name, tuple(x.shape), int(ne.sum()), (x.float() - a.float()).abs().max().item(), a.float().abs().max().item(), ne.nonzero()[0].tolist()))
        found[gi].append(d)
for gi, (gname, g) in enumerate(groups):
    cnt = {}
    for d in found[gi]:
        for e in d:
            cnt[e[0]] = cnt.get(e[0], 0) + 1
    # parameter gradients summed with fp32 atomics differ in the last bits on most replays: not what is looked for here
    noise = {i for i, c in cnt.items() if c > 0.5 * R}
    bad = [(r, [e for e in d if e[0] not in noise]) for r, d in enumerate(found[gi])]
    bad = [(r, d) for r, d in bad if d]
    print("%-8s %3d of %d replays differ from the first (%d always-noisy tensors ignored: %s)"
          % (gname, len(bad), R, len(noise), sorted({g[i][0] for i in noise})[:8]), flush=True)
    for r, d in bad[:6]:
        i, name, shape, n, md, mr, idx = d[0]
        print("   replay %d: %d tensors, first #%d %s %s: %d elements, max |diff| %.3g of max %.3g, first index %s" % (r, len(d), i, name, shape, n, md, mr, idx), flush=True)
