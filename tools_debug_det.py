import sys, torch
sys.path.insert(0, ".")
from tests.test_gpu_trainer import _mk
from representationlearning_amd.trainer import Trainer
from representationlearning_amd.configs import synthetic_batch
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
m = _mk(1)
tr = Trainer(m, bf16=bf16, base_lr=0.0, weight_decay=0.0)
img, lab = synthetic_batch(2, 128, seed=5)
gs = []
for i in range(3):
    l = tr.step(img, dict(cls=lab)); gs.append(tr.flat.grad.clone()); print("loss", float(l))
names = [n for n, p in m.named_parameters() if p.requires_grad]
worst = []
for n, p, o in zip(names, tr.flat.params, tr.flat.offsets):
    a, b = gs[1][o:o + p.numel()], gs[2][o:o + p.numel()]
    d = float((a - b).norm() / (b.norm() + 1e-20))
    worst.append((d, n, float(b.norm())))
worst.sort(reverse=True)
for w in worst[:25]: print("%.3e  %-70s |g|=%.3e" % w)
print("params with rel diff > 1e-3:", sum(1 for w in worst if w[0] > 1e-3), "of", len(worst))
