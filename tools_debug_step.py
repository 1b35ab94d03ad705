import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
from tests.test_gpu_model import build
from oracle.procedural import seeded_input, proc_labels
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = build(variant).train()
x = seeded_input((2, 3, S, S), 7).cuda(); y = proc_labels(2, S, S, 6, 8).cuda()
import representationlearning_amd.autograd as AG
orig = AG.GatedWindowCrossAttention.forward
def hook(ctx, x_, y_, *a):
    print("attn fwd", x_.dtype, y_.dtype, tuple(x_.shape), x_.is_contiguous(), flush=True)
    return orig(ctx, x_, y_, *a)
AG.GatedWindowCrossAttention.forward = staticmethod(hook)
ob = AG.GatedWindowCrossAttention.backward
def bhook(ctx, d):
    print("attn bwd", d.dtype, tuple(d.shape), flush=True)
    r = ob(ctx, d); torch.cuda.synchronize(); print("  ok", flush=True); return r
AG.GatedWindowCrossAttention.backward = staticmethod(bhook)
with torch.autocast("cuda", dtype=torch.bfloat16):
    loss = m(x, dict(cls=y))["fc_loss"]
print("loss", float(loss), flush=True)
loss.backward()
torch.cuda.synchronize()
print("done")
