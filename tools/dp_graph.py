"""Single-rank NCCL process group + forced DP plumbing (buckets, SyncBN all-reduces): eager vs hipGraph (run on the GPU box)."""
import os, sys, time, torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RSSF_FORCE_DP"] = "1"
import torch.distributed as dist
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer
registry.register_all()
torch.manual_seed(0)
model = registry.MODEL["RSSFormer"](rssformer_config("base")).cuda()
tr = Trainer(model, base_lr=0.01, max_iters=1000, bf16=True, sync_bn=True)
print("dp buckets:", tr.buckets is not None, "syncbn exchange:", type(tr.comm).__name__, "use_graph:", tr.use_graph)
img, lab = synthetic_batch(16, 512, seed=1)
tgt = dict(cls=lab)
losses = []
for i in range(12):
    if i == 6:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    losses.append(float(tr.step(img, tgt)))
torch.cuda.synchronize()
print("ms/step %.1f" % ((time.perf_counter() - t0) / 6 * 1e3), "graph captured:", tr.graph is not None)
print("losses", [round(l, 4) for l in losses])
n0 = tr.comm.n_syncbn
tr.use_graph = False
tr._eager_step(img, tgt)
torch.cuda.synchronize()
print("SyncBN exchanges per step: main stream %d, side streams %s" % (tr.comm.n_syncbn - n0, [c.n_syncbn // 5 for c in tr.side_comms]))
tr.close()
dist.destroy_process_group()
