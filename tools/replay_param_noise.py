"""Replays one captured step (lr = 0, deterministic) R times and reports, per parameter, how far its gradient in the flat buffer strays
between replays: max over replays of |g - g_first|, with the gradient's own scale beside it.  Order noise of fp32 atomics sits at
1e-10; anything above ABS (default 1e-6) is a launch that produced something else.
   python tools/replay_param_noise.py [replays=300] [batch=2] [size=128]     (ALT=1: a replay on another batch before each one)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
os.environ.setdefault("RSSF_BRANCH_STREAMS", "1")
import torch
from test_gpu_trainer import _mk
from representationlearning_amd.trainer import Trainer
from representationlearning_amd.configs import synthetic_batch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ALT = os.environ.get("ALT", "0") == "1"
ABS = float(os.environ.get("ABS", "1e-6"))
img, lab = synthetic_batch(BATCH, SIZE, seed=5)
img2, lab2 = synthetic_batch(BATCH, SIZE, seed=11)
t = Trainer(_mk(6), bf16=True, base_lr=0.0, use_graph=os.environ.get("GRAPH", "1") == "1", deterministic=True)
while (t.graph is None or t._replayed < 1) if t.use_graph else t._steps < 4:
    t.step(img, dict(cls=lab))
torch.cuda.synchronize()
names = {id(p): k for k, p in t.model.named_parameters()}
order = [(names[id(p)], o, p.numel()) for p, o in zip(t.flat.params, t.flat.offsets)]
ref = t.flat.grad.clone()
worst = torch.zeros_like(ref)
hits = {}
for r in range(R):
    if ALT:
        t.step(img2, dict(cls=lab2))
    t.step(img, dict(cls=lab))
    torch.cuda.synchronize()
    d = (t.flat.grad - ref).abs()
    worst = torch.maximum(worst, d)
    if float(d.max()) > ABS:
        for n, o, k in order:
            if float(d[o:o + k].max()) > ABS:
                hits.setdefault(n, []).append(r)
                if os.environ.get("DUMP") and len(hits[n]) <= 2 and k <= 512:
                    dd = (t.flat.grad - ref)[o:o + k]
                    nz = (dd.abs() > ABS).nonzero().flatten().tolist()
                    print("   %s replay %d: offset %d (byte %d mod 128 = %d), %d of %d elements off: idx %s" % (n, r, o, 4 * o, (t.flat.grad.data_ptr() + 4 * o) % 128, len(nz), k, nz[:40]))
                    print("      diff", [float("%.3g" % v) for v in dd[nz[:12]].tolist()], " ref", [float("%.3g" % v) for v in ref[o:o + k][nz[:12]].tolist()])
print("replays %d, graph %s, ALT %s: %d parameters strayed by more than %g" % (R, t.graph is not None, ALT, len(hits), ABS))
for n, o, k in order:
    if n in hits:
        print("  %-62s %3d replays (first %s)  max |diff| %.3g   max |grad| %.3g" % (n, len(hits[n]), hits[n][:6], float(worst[o:o + k].max()), float(ref[o:o + k].abs().max())))
