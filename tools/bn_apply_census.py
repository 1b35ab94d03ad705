"""Which separate BatchNorm finalize+apply passes (rssf_bn_finalize_apply, the one-problem form) a Base step launches: rows, channels,
activation, residuals, MB moved.   python tools/bn_apply_census.py [batch=16]"""
import os, sys, collections, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSSF_GRAPH"] = "0"
import torch
from representationlearning_amd import _lib as L
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer
lib = L.load()
log = collections.Counter()
real = lib.rssf_bn_finalize_apply
def hook(raw, stats, gamma, beta, rm, rv, mi, ss, res_pre, res_post, y, rows, C, act, *rest):
    log[(int(rows), int(C), int(act), bool(res_pre), bool(res_post))] += 1
    return real(raw, stats, gamma, beta, rm, rv, mi, ss, res_pre, res_post, y, rows, C, act, *rest)
lib.rssf_bn_finalize_apply = hook
registry.register_all()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = registry.MODEL["RSSFormer"](rssformer_config("base")).cuda()
tr = Trainer(model, bf16=True, use_graph=False)
img, lab = synthetic_batch(B, 512, seed=1)
tr.step(img, dict(cls=lab)); tr.step(img, dict(cls=lab)); log.clear()
tr.step(img, dict(cls=lab)); torch.cuda.synchronize()
for (rows, C, act, rp, rq), n in sorted(log.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1]):
    mb = rows * C * 2 * (2 + rp + rq) / 1e6
    print("x%-3d rows %8d  C %4d  act %d  res_pre %d res_post %d   %6.1f MB each" % (n, rows, C, act, rp, rq, mb))
