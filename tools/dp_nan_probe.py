"""1-rank forced-DP training steps with real updates: losses for one combination of RSSF_GRAPH / RSSF_GRAD_OVERLAP / lock-step."""
import os, sys, torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
if os.environ.get("NOFORCE") != "1":
    os.environ["RSSF_FORCE_DP"] = "1"
import torch.distributed as dist
sys.path.insert(0, ".")
if os.environ.get("NODIST") != "1":
    dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer
registry.register_all(); torch.manual_seed(0)
B, S = int(os.environ.get("B", 16)), int(os.environ.get("S", 512))
tr = Trainer(registry.MODEL["RSSFormer"](rssformer_config("base")).cuda(), bf16=True, sync_bn=os.environ.get("SYNC", "1") == "1",
             max_iters=int(os.environ.get("MAXIT", 30000)))
img, lab = synthetic_batch(B, S, seed=1)
losses = []
for i in range(12):
    if i == 6 and os.environ.get("SYNC6") == "1":
        torch.cuda.synchronize()
    if i == 6 and os.environ.get("SLEEP6") == "1":
        import time; time.sleep(1.0)
    losses.append(float(tr.step(img, dict(cls=lab))))
    if os.environ.get("TRACE2") == "1" and i == 6:
        s_img, s_tgt = tr._static
        print("static img equal", bool(torch.equal(s_img, img)), "static labels equal", bool(torch.equal(s_tgt["cls"], lab)),
              "label range", int(s_tgt["cls"].min()), int(s_tgt["cls"].max()), "lr", float(tr.lr_dev), flush=True)
        m = tr.model
        with torch.no_grad():
            print("param finite before?", "flat nan count", int((~torch.isfinite(tr.flat.flat)).sum()), "grad nan count", int((~torch.isfinite(tr.flat.grad)).sum()),
                  "mom nan", int((~torch.isfinite(tr.flat.mom)).sum()), flush=True)
            bufs = [(k, b) for k, b in m.named_buffers() if b.is_floating_point()]
            print("nan buffers", [k for k, b in bufs if not torch.isfinite(b).all()][:5], flush=True)
    if os.environ.get("TRACE") == "1" and 4 <= i <= 7:
        torch.cuda.synchronize() if os.environ.get("TRACESYNC") == "1" else None
        names = [k for k, p in tr.model.named_parameters() if p.requires_grad]
        badg = [k for k, p, o in zip(names, tr.flat.params, tr.flat.offsets) if not torch.isfinite(tr.flat.grad[o:o + p.numel()]).all()]
        badp = [k for k, p in zip(names, tr.flat.params) if not torch.isfinite(p).all()]
        badb = [k for k, b in tr.model.named_buffers() if b.is_floating_point() and not torch.isfinite(b).all()]
        print("step", i, "loss", losses[-1], "nan grads", len(badg), badg[:4], "nan params", len(badp), badp[:4], "nan buffers", len(badb), badb[:4],
              "gnorm", float(tr.sqnorm[0]) ** 0.5, flush=True)
print("GRAPH=%s OVERLAP=%s SYNC=%s graph=%s losses %s" % (os.environ.get("RSSF_GRAPH"), os.environ.get("RSSF_GRAD_OVERLAP"), os.environ.get("SYNC", "1"),
      tr.graph is not None, [round(l, 4) for l in losses]), flush=True)
tr.close()
if dist.is_initialized():
    dist.destroy_process_group()
