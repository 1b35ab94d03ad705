import sys, torch
sys.path.insert(0, ".")
from representationlearning_amd.core import registry
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd import nnf
registry.register_all()
m = registry.MODEL["RSSFormer"](rssformer_config("base")).cuda().train()
x, y = synthetic_batch(2, 128)
orig = nnf._ConvBNAct.forward
seen = {}
def fwd(ctx, x_, *a):
    spec = a[6]
    key = (str(x_.dtype), spec.cin, spec.cout, spec.ntaps, spec.stride, None if a[0] is None else str(a[0].dtype))
    seen[key] = seen.get(key, 0) + 1
    return orig(ctx, x_, *a)
nnf._ConvBNAct.forward = staticmethod(fwd)
with torch.autocast("cuda", dtype=torch.bfloat16):
    out = m(x, dict(cls=y))
for k, v in sorted(seen.items(), key=str): print(k, v)
print("---- per-module output dtypes")
from representationlearning_amd.module.baseline.base_hrnet._hrnet_rssformer import HighResolutionModule
def hook(mod, inp, out):
    print(type(mod).__name__, [str(t.dtype) for t in inp[0]], "->", [str(t.dtype) for t in out])
hs = [mm.register_forward_hook(hook) for mm in m.modules() if isinstance(mm, HighResolutionModule)]
with torch.autocast("cuda", dtype=torch.bfloat16):
    out = m(x, dict(cls=y))
mod = m.backbone.hrnet.stage2[0]
with torch.autocast("cuda", dtype=torch.bfloat16):
    a = torch.randn(2, 32, 32, 32, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(2, 64, 16, 16, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    t = nnf.run_sequential(mod.fuse_layers[1][0], a); print("fuse10", t.dtype)
    low = 0; low = low + b; print("low", low.dtype)
    yy = t + low; print("sum", yy.dtype); print("relu", mod.relu(yy).dtype)
    u = nnf.run_sequential(mod.fuse_layers[0][1], b); print("fuse01 (with upsample)", u.dtype)
