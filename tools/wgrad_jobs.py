"""Census of the split-K weight-gradient partial planes of one training step (what the batched second stage reads):
python tools/wgrad_jobs.py  ->  one line per job shape, sorted by bytes."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import _lib
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer

_lib.load()
registry.register_all()
torch.manual_seed(2333)
os.environ["RSSF_GRAPH"] = "0"
model = registry.MODEL["RSSFormer"](rssformer_config(sys.argv[1] if len(sys.argv) > 1 else "base")).cuda()
tr = Trainer(model, bf16=True, sync_bn=True)
img, lab = synthetic_batch(16, 512, seed=2333)
for _ in range(3):
    tr.step(img, dict(cls=lab))
torch.cuda.synchronize()
plan = tr.wgrad_plan if hasattr(tr, "wgrad_plan") else tr.rt.wgrad_plan
rows = collections.Counter()
for j in plan.jobs_host:
    rows[(j.ntaps, j.cout, j.cin, j.ksplit)] += 1
tot = 0
out = []
for (nt, co, ci, ks), n in rows.items():
    b = nt * co * ci * ks * 4 * n
    tot += b
    out.append((b, "x%-3d taps %2d cout %4d cin %4d ksplit %4d   %8.1f MB" % (n, nt, co, ci, ks, b / 1e6)))
for b, s in sorted(out, reverse=True):
    print(s)
print("jobs %d  partial planes read per step: %.1f MB   arena %.1f MB" % (len(plan.jobs_host), tot / 1e6, plan.arena.numel() * 4 / 1e6))
