"""bf16 (benchmark mode) vs fp32-I/O mode of the same training-mode forward: relative error per stage, logits, and the argmax
agreement as a function of the fp32 top-2 margin.   python tools/mode_diff.py base 16 512 [large 4 1024 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import nnf
from representationlearning_amd.configs import rssformer_config, synthetic_batch
from representationlearning_amd.core import registry

registry.register_all()


def run(variant, B, S, mode, taps):
    torch.manual_seed(2333)
    m = registry.MODEL["RSSFormer"](rssformer_config(variant)).cuda().train()
    hr = m.backbone.hrnet
    def hook(name):
        def f(mod, i, o):
            t = o[0] if isinstance(o, (list, tuple)) else o
            taps[name] = t.detach().float()
        return f
    hr.layer1.register_forward_hook(hook("layer1"))
    for st in (2, 3, 4):
        for k, mod in enumerate(getattr(hr, "stage%d" % st)):
            mod.register_forward_hook(hook("stage%d.%d" % (st, k)))
            mod.transformer.register_forward_hook(hook("stage%d.%d.transformer" % (st, k)))
            mod.branches[0].register_forward_hook(hook("stage%d.%d.branch0" % (st, k)))
    img, lab = synthetic_batch(B, S, seed=2333)
    rt = nnf.Runtime(); rt.deterministic = True
    with nnf.use(rt), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16"):
        loss = m(img, dict(cls=lab))["fc_loss"]
    taps["logits"] = m._last_logits.detach().float()
    return float(loss)


args = sys.argv[1:]
for i in range(0, len(args), 3):
    variant, B, S = args[i], int(args[i + 1]), int(args[i + 2])
    t32, t16 = {}, {}
    l32 = run(variant, B, S, "fp32", t32)
    l16 = run(variant, B, S, "bf16", t16)
    print("==== %s %dx%d  loss fp32 %.6f bf16 %.6f (%.2f %%)" % (variant, B, S, l32, l16, 100 * (l16 - l32) / l32), flush=True)
    for k in t32:
        a, b = t32[k], t16[k]
        print("  %-28s rel %.4f   |ref| %.4f" % (k, float((a - b).norm() / a.norm()), float(a.abs().mean())))
    g32, g16 = t32["logits"], t16["logits"]
    top2 = g32.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    sd = float(g32.std()); rms = float((g16 - g32).pow(2).mean().sqrt())
    print("  logits std %.4f  rms err %.4f (%.3f std)" % (sd, rms, rms / sd))
    agree = g32.argmax(1) == g16.argmax(1)
    for tau in (0.0, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0):
        sel = margin > tau * sd
        print("    margin > %.2f std: %5.1f %% of pixels, agreement %.4f" % (tau, 100 * float(sel.float().mean()), float(agree[sel].float().mean()) if sel.any() else float("nan")))
    del t32, t16
    torch.cuda.empty_cache()
