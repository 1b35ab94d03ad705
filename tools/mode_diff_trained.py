"""Per-stage bf16 drift on a TRAINED model: Base 16 x 512^2 trained for 300 replayed bf16 steps on the bench's fixed
batch (tests/test_gpu_model.py::test_base_trained_model_bf16_vs_fp32_end_to_end), then the same weights forwarded in fp32-I/O mode and in
bf16: relative deviation of every stage output (free-running: each stage sees the drifted input of its mode), logits, argmax agreement
by fp32 top-2 margin.   python tools/mode_diff_trained.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import nnf
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry
from representationlearning_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
registry.register_all()
torch.manual_seed(2333)
m = registry.MODEL["RSSFormer"](rssformer_config("base")).cuda()
tr = Trainer(m, bf16=True)
img, lab = synthetic_batch(16, 512, seed=2333)
first = float(tr.step(img, dict(cls=lab)))
for _ in range(steps - 1):
    last = tr.step(img, dict(cls=lab))
print("trained %d steps: loss %.4f -> %.4f" % (steps, first, float(last)))
taps = {}
cur = [None]
hr = m.backbone.hrnet


def hook(name):
    def f(mod, i, o):
        t = o[0] if isinstance(o, (list, tuple)) else o
        taps[cur[0]][name] = t.detach().float()
    return f


hr.layer1.register_forward_hook(hook("layer1"))
for st in (2, 3, 4):
    for k, mod in enumerate(getattr(hr, "stage%d" % st)):
        mod.branches[0].register_forward_hook(hook("stage%d.%d.branch0 (BasicBlocks)" % (st, k)))
        mod.transformer.register_forward_hook(hook("stage%d.%d.transformer" % (st, k)))
        mod.register_forward_hook(hook("stage%d.%d (fused, output 0)" % (st, k)))
rt = nnf.Runtime(); rt.deterministic = True
m.train()
res = {}
for mode in ("fp32", "bf16"):
    cur[0] = mode; taps[mode] = {}
    with torch.no_grad(), nnf.use(rt), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16"):
        loss = m(img, dict(cls=lab))["fc_loss"]
    res[mode] = (float(loss), m._last_logits.float())
(l32, g32), (l16, g16) = res["fp32"], res["bf16"]
print("loss fp32 %.5f bf16 %.5f (%+.1f %%)" % (l32, l16, 100 * (l16 - l32) / l32))
print("%-40s %10s %10s" % ("stage output (free-running)", "rel dev", "|fp32| mean"))
for k in taps["fp32"]:
    a, b = taps["fp32"][k], taps["bf16"][k]
    print("%-40s %10.4f %10.4f" % (k, float((a - b).norm() / a.norm()), float(a.abs().mean())))
print("%-40s %10.4f %10.4f" % ("logits", float((g16 - g32).norm() / g32.norm()), float(g32.abs().mean())))
top2 = g32.topk(2, dim=1).values
margin, sd = top2[:, 0] - top2[:, 1], float(g32.std())
agree = g32.argmax(1) == g16.argmax(1)
for tau in (0.0, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0):
    sel = margin > tau * sd
    print("  fp32 top-2 margin > %.2f std: %5.1f %% of the pixels, argmax agreement %.4f" % (tau, 100 * float(sel.float().mean()), float(agree[sel].float().mean())))

