"""Loss curve of N training steps on the fixed synthetic batch (hipGraph replay), to see the optimisation actually optimise."""
import sys, torch
sys.path.insert(0, ".")
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer
registry.register_all(); torch.manual_seed(2333)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tr = Trainer(registry.MODEL["RSSFormer"](rssformer_config("base")).cuda(), bf16=True)
img, lab = synthetic_batch(16, 512, seed=2333)
losses = []
for i in range(n):
    l = tr.step(img, dict(cls=lab))
    if i % 20 == 0 or i == n - 1:
        torch.cuda.synchronize()
        losses.append((i, round(float(l), 4)))
print("losses", losses)
