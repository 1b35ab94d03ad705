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
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=40, max_shapes_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=40, max_src_column_width=110))
