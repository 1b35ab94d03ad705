"""Every convolution launch of one eager training step (Base 16 x 512^2 bf16) with its shape: forward / data gradient / weight gradient.
    python tools/conv_shapes.py   (on the GPU box)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSSF_GRAPH"] = "0"
import torch
from representationlearning_amd import nnf
from representationlearning_amd.configs import rssformer_config
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer

log = collections.Counter()
f0, d0, w0 = nnf._conv_forward, nnf._conv_dgrad, nnf._conv_wgrad
def fwd(spec, xh, *a, **k):
    log[("fwd", tuple(xh.shape), spec.cout, spec.ntaps, spec.stride)] += 1
    return f0(spec, xh, *a, **k)
def dgr(spec, dout, weights, in_shape, *a, **k):
    log[("dgrad", tuple(in_shape), spec.cout, spec.ntaps, spec.stride)] += 1
    return d0(spec, dout, weights, in_shape, *a, **k)
def wgr(spec, dout, xh, *a, **k):
    log[("wgrad", tuple(xh.shape), spec.cout, spec.ntaps, spec.stride)] += 1
    return w0(spec, dout, xh, *a, **k)
nnf._conv_forward, nnf._conv_dgrad, nnf._conv_wgrad = fwd, dgr, wgr
registry.register_all()
model = registry.MODEL["RSSFormer"](rssformer_config("base")).cuda()
tr = Trainer(model, bf16=True, sync_bn=False, use_graph=False)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(BATCH, 3, 512, 512, device="cuda"); y = torch.randint(0, 6, (BATCH, 512, 512), device="cuda")
tr.step(x, dict(cls=y)); log.clear()
tr.step(x, dict(cls=y)); torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: (kv[0][0], -kv[0][1][1] * kv[0][1][2] * kv[0][1][3] * kv[0][2] * kv[0][3])):
    kind, shp, cout, nt, st = k
    B, H, W, C = shp
    gf = 2.0 * B * H * W * C * cout * nt / st / st / 1e9
    print("%-6s x%-3d in %-22s cout %4d taps %2d stride %d   %8.2f GFLOP" % (kind, v, shp, cout, nt, st, gf))
