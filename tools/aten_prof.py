"""Which torch (non-rssf) kernels a training step still launches: aten op + input shapes, run on the GPU box."""
import os, sys, torch
os.environ["RSSF_GRAPH"] = "0"
sys.path.insert(0, ".")
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer
registry.register_all()
torch.manual_seed(0)
model = registry.MODEL["RSSFormer"](rssformer_config("base")).cuda()
tr = Trainer(model, bf16=True, sync_bn=True)
img, lab = synthetic_batch(16, 512, seed=1)
tgt = dict(cls=lab)
for _ in range(3): tr.step(img, tgt)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(img, tgt)
    torch.cuda.synchronize()
import collections
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if not e.name.startswith("aten::") or dt <= 0:
        continue
    stack = [f for f in (e.stack or []) if "representationlearning_amd" in f or "trainer.py" in f][:3]
    if not stack:                                    # backward-thread event: attribute to the autograd node that is its parent
        p_ = e.cpu_parent
        while p_ is not None and not ("Backward" in p_.name or p_.name.startswith("autograd::engine")):
            p_ = p_.cpu_parent
        stack = [p_.name if p_ is not None else "?"]
    key = (e.name, str(e.input_shapes)[:90], " <- ".join(s.split("/")[-1][:70] for s in stack))
    rows[key][0] += 1
    rows[key][1] += dt
tot = sum(v[1] for v in rows.values())
print("ATen kernels in one eager step: %d launches, %.1f us" % (sum(v[0] for v in rows.values()), tot))
for (name, shapes, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%4d %8.1f us  %-22s %-90s %s" % (n, us, name, shapes, where))
