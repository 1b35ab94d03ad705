import os, sys, torch, torch.distributed as dist
sys.path.insert(0, ".")
from tests.test_gpu_trainer import _mk, _run
from representationlearning_amd.trainer import Trainer
from representationlearning_amd import nnf
a = _run(Trainer(_mk(1), bf16=False)); b = _run(Trainer(_mk(1), bf16=False))
print("plain A", a); print("plain B", b)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RSSF_FORCE_DP="1")
dist.init_process_group("nccl", rank=0, world_size=1)
tr = Trainer(_mk(1), bf16=False, sync_bn=False); print("dp nosync", _run(tr))
tr = Trainer(_mk(1), bf16=False, sync_bn=True); print("dp sync  ", _run(tr))
# buckets but hooks disabled: launch everything at finish only
tr = Trainer(_mk(1), bf16=False, sync_bn=False)
tr.buckets._launch_orig = tr.buckets._launch
import types
def hookless(self, b):
    pass
orig = tr.buckets._launch
tr.buckets._make_hook = lambda i: (lambda p: None)
nnf.set_direct_grad(True, None)
print("dp finish-only", _run(tr))
dist.destroy_process_group()
