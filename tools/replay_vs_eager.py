"""Which parameter leaves the eager trajectory first?  A deterministic eager trainer and a graph-replaying one step side by side
(same seed, same batch, lr > 0); after every step the flat parameter and gradient buffers are compared per parameter.
   python tools/replay_vs_eager.py [branch_streams=1] [steps=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSSF_BRANCH_STREAMS"] = sys.argv[1] if len(sys.argv) > 1 else "1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from test_gpu_trainer import _mk
from representationlearning_amd.trainer import Trainer
from representationlearning_amd.configs import synthetic_batch

img, lab = synthetic_batch(2, 128, seed=5)
te = Trainer(_mk(6), bf16=True, base_lr=0.002, use_graph=False, deterministic=True)
tg = Trainer(_mk(6), bf16=True, base_lr=0.002, use_graph=True, deterministic=True)
names = {id(p): k for k, p in tg.model.named_parameters()}
order = [(names[id(p)], o, p.numel()) for p, o in zip(tg.flat.params, tg.flat.offsets)]
for i in range(steps):
    a = float(te.step(img, dict(cls=lab)))
    b = float(tg.step(img, dict(cls=lab)))
    torch.cuda.synchronize()
    rows = []
    for what, x, y in (("grad", tg.flat.grad, te.flat.grad), ("weight", tg.flat.flat, te.flat.flat)):
        d = (x - y).abs()
        if float(d.max()) == 0:
            continue
        for n, o, k in order:
            dm = float(d[o:o + k].max())
            if dm > 0:
                rows.append((dm / (float(y[o:o + k].abs().max()) + 1e-30), what, n, dm))
    rows.sort(reverse=True)
    print("step %2d loss eager %.7f graph %.7f  %d differing (grad/weight, parameter) pairs" % (i, a, b, len(rows)), flush=True)
    for r, what, n, dm in rows[:int(os.environ.get('TOP', '5'))]:
        print("      %-6s %-60s rel %.2e abs %.2e" % (what, n, r, dm))
    if a != b and abs(a - b) / abs(a) > 1e-3:
        break
