#!/usr/bin/env python3
"""Golden-vector generator.  BUILD CONTAINER ONLY (needs /root/reference).

Imports the reference's own RSSFormer implementation (with the ~60-line `ever`/`timm`
stand-ins under oracle/refimport/stubs — those two packages are un-vendored, unpinned
third-party deps), loads *procedural* weights (oracle/procedural.py),
runs small cases and stores inputs' recipes + outputs as tests/golden/*.npz.

The reference has no tests/fixtures of its own for this path, so these
vectors — outputs of the reference itself run here — are what pins the CPU oracle and,
through it, the HIP path.  Nothing from the reference's source text is stored.

Usage:  python oracle/make_golden.py            (rewrites tests/golden/)
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/RSSFormer-TIP2023"
sys.path.insert(0, os.path.join(HERE, "refimport", "stubs"))
sys.path.insert(1, REF)
sys.path.insert(2, ROOT)
warnings.filterwarnings("ignore")

from oracle.procedural import procedural_state, proc_input, proc_labels, seeded_state, seeded_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def npy(t):
    return t.detach().cpu().numpy()


def load_proc(module):
    sd = module.state_dict()
    module.load_state_dict(procedural_state(sd))
    return module


def load_seeded(module, seed=1234):
    module.load_state_dict(seeded_state(module.state_dict(), seed))
    return module


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def case_mhca():
    from module.baseline.base_hrnet.modules.DAL import Mhca
    for C, nw, tag in ((32, 3, "c32"), (18, 2, "c18"), (48, 2, "c48")):
        m = load_proc(Mhca(C, 2, dropout=0.0)).train()
        x = proc_input((49, nw, C), 0.3).requires_grad_()
        y = proc_input((49, nw, C), 1.1).requires_grad_()
        out = m(x, y, y)
        (out * proc_input(out.shape, 2.0)).sum().backward()
        g = {("g_" + k.replace(".", "_")): npy(p.grad) for k, p in m.named_parameters()}
        save(f"mhca_{tag}", out=npy(out), gx=npy(x.grad), gy=npy(y.grad), **g)


def case_attention():
    """InterlacedPoolAttention2: gate + pad/permute + Mhca, incl. pad 0 / odd / even and N%C!=0."""
    from module.baseline.base_hrnet.modules.multihead_isa_pool_attention import InterlacedPoolAttention2
    for (B, C, H, W) in ((1, 32, 7, 7), (2, 32, 10, 10), (1, 32, 14, 14), (1, 32, 20, 12),
                         (1, 18, 9, 11), (1, 48, 8, 8)):
        m = load_proc(InterlacedPoolAttention2(C, 2, window_size=7, rpe=True, dropout=0.0)).train()
        cap = {}
        m.weight_levels.register_forward_hook(lambda mod, i, o: cap.__setitem__("lv", o))
        x = proc_input((B, H * W, C), 0.2).requires_grad_()
        y = proc_input((B, H * W, C), 0.8).requires_grad_()
        out = m(x, y, H, W)
        (out * proc_input(out.shape, 1.7)).sum().backward()
        g = {("g_" + k.replace(".", "_")): npy(p.grad) for k, p in m.named_parameters()}
        save(f"attn_B{B}_C{C}_H{H}_W{W}", out=npy(out), gate_logits=npy(cap["lv"]),
             gx=npy(x.grad), gy=npy(y.grad), **g)


def case_block():
    from module.baseline.base_hrnet.modules.MTFM import GeneralTransformerBlock
    for (B, C, H, W) in ((1, 32, 10, 10), (2, 32, 14, 9), (1, 18, 8, 8)):
        for mode in ("train", "eval"):
            m = load_seeded(GeneralTransformerBlock(C, C, 2))
            m.train(mode == "train")
            low = seeded_input((B, C, H, W), 11).requires_grad_()
            high = seeded_input((B, C, H, W), 12).requires_grad_()
            out = m(low, high)
            arrs = dict(out=npy(out))
            if mode == "train":
                out.square().mean().backward()
                arrs.update(glow=npy(low.grad), ghigh=npy(high.grad))
                arrs.update({("g_" + k.replace(".", "_")): npy(p.grad) for k, p in m.named_parameters()})
                arrs.update({("b_" + k.replace(".", "_")): npy(v) for k, v in m.named_buffers()
                             if "running" in k})
            save(f"block_{mode}_B{B}_C{C}_H{H}_W{W}", **arrs)


def case_mlp():
    from module.baseline.base_hrnet.modules.ffn_block import MlpDWBN
    import torch.nn as nn
    for (B, C, H, W) in ((2, 32, 16, 13), (1, 18, 30, 30)):
        m = load_seeded(MlpDWBN(C, 4 * C, C, nn.GELU, nn.GELU, 0.0)).train()
        z = seeded_input((B, H * W, C), 21).requires_grad_()
        out = m(z, H, W)
        (out * seeded_input(out.shape, 22)).sum().backward()
        arrs = dict(out=npy(out), gz=npy(z.grad))
        arrs.update({("g_" + k.replace(".", "_")): npy(p.grad) for k, p in m.named_parameters()})
        arrs.update({("b_" + k.replace(".", "_")): npy(v) for k, v in m.named_buffers() if "running" in k})
        save(f"mlp_B{B}_C{C}_H{H}_W{W}", **arrs)


def case_loss():
    from module.CGFL import SegmentationLossaux
    from ever.interface import AttrDict
    crit = SegmentationLossaux(AttrDict.wrap(dict(ignore_index=-1, ce=dict())))
    B, K, H, W = 3, 6, 12, 10
    variants = {}
    y0 = proc_labels(B, H, W, 6, 2)
    variants["mixed"] = y0
    y1 = y0.clone(); y1[1] = -1                      # one all-ignore sample
    variants["one_all_ignore"] = y1
    y2 = y0.clone(); y2[2] = torch.where(y2[2] > 0, torch.zeros_like(y2[2]), y2[2])  # no foreground
    variants["no_fg"] = y2
    y3 = y0.clone().clamp(min=1)                      # all foreground, nothing ignored
    variants["all_fg"] = y3
    for tag, y in variants.items():
        lg = (proc_input((B, K, H, W), 0.4) * 2.0).requires_grad_()
        aux = (proc_input((B, 7), 1.3)).requires_grad_()
        loss = crit(lg, y, aux)["fc_loss"]
        loss.backward()
        save(f"loss_{tag}", y=npy(y), loss=npy(loss), glogits=npy(lg.grad),
             gaux=np.zeros(1) if aux.grad is None else npy(aux.grad))


def build_model(variant, classes=6):
    from module.baseline.hrnet_aux import HRNetFusion
    from module.baseline.base_hrnet._hrnet_rssformer import HighResolutionNet, model_extra
    import torch.nn as nn
    cfg = dict(backbone=dict(hrnet_type="hrnetv2_w32", pretrained=False), neck=dict(in_channels=480),
               head=dict(in_channels=480, upsample_scale=4.0), classes=classes,
               loss=dict(ignore_index=-1, ce=dict()))
    table = {"tiny": ("hrnetv2_w32s", 270, 18), "base": ("hrnetv2_w32", 480, 32),
             "large": ("hrnetv2_w48", 720, 48)}[variant]
    cfg["neck"]["in_channels"] = table[1]
    cfg["head"]["in_channels"] = table[1]
    m = HRNetFusion(cfg)
    if variant != "base":   # SURVEY §8 "variants": swap the backbone table, widen headaux
        m.backbone.hrnet = HighResolutionNet(model_extra[table[0]], False, False, -1)
        m.headaux = nn.Sequential(nn.Linear(table[2], 7))
    return m


def case_neck_head():
    m = load_proc(build_model("base")).train()
    feats = [proc_input((2, c, 12 // s, 8 // s), 0.3 * i) for i, (c, s) in
             enumerate(((32, 1), (64, 2), (128, 4), (256, 4)))]
    feats[3] = proc_input((2, 256, 2, 1), 0.9)
    feats = [f.requires_grad_() for f in feats]
    t, f0 = m.neck(feats)
    aux = m.headaux(m.avg_pool(f0).flatten(1))
    lg = m.head(t)
    (lg * proc_input(lg.shape, 0.7)).sum().backward()
    arrs = dict(logits=npy(lg), aux=npy(aux))
    arrs.update({f"gf{i}": npy(f.grad) for i, f in enumerate(feats)})
    arrs["g_head_w"] = npy(m.head[0].weight.grad)
    arrs["g_neck_w_sum"] = npy(m.neck.fuse_conv[0].weight.grad.sum((2, 3)))
    save("neck_head", **arrs)


def case_model(variant, B, S, tag):
    torch.manual_seed(0)
    m = load_seeded(build_model(variant)).train()
    x = seeded_input((B, 3, S, S), 7)
    y = proc_labels(B, S, S, 6, 8)
    taps = {}
    hr = m.backbone.hrnet
    hr.layer1.register_forward_hook(lambda mod, i, o: taps.__setitem__("layer1", o))
    for st in (2, 3, 4):
        getattr(hr, f"stage{st}").register_forward_hook(
            lambda mod, i, o, st=st: taps.__setitem__(f"stage{st}", o[0]))
    m.head.register_forward_hook(lambda mod, i, o: taps.__setitem__("logits", o))
    m.headaux.register_forward_hook(lambda mod, i, o: taps.__setitem__("aux", o))
    loss = m(x, dict(cls=y))["fc_loss"]
    loss.backward()
    arrs = dict(loss=npy(loss), aux=npy(taps["aux"]))
    lg = taps["logits"]
    arrs["logits_sample"] = npy(lg[:, :, ::max(1, S // 8), ::max(1, S // 8)])
    for k in ("layer1", "stage2", "stage3", "stage4", "logits"):
        t = taps[k]
        arrs[f"sum_{k}"] = npy(t.double().sum())
        arrs[f"abs_{k}"] = npy(t.double().abs().mean())
    # gradient norms per parameter (sorted-name order) + a few full small grads
    names, gn = [], []
    for k, p in sorted(m.named_parameters()):
        names.append(k)
        gn.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    arrs["grad_names"] = np.array(names)
    arrs["grad_norms"] = np.array(gn)
    # conditioning yardstick: the same step by the reference in fp64.  Gradients routed through max()/argmax
    # (gate channel-max, alpha's max(M), ReLU kinks) are discontinuous, so a few parameter gradients differ by
    # 1-20 % between fp32 and fp64 of the SAME code; tests scale their tolerance by this distance.
    m64 = load_seeded(build_model(variant)).double().train()
    loss64 = m64(x.double(), dict(cls=y))["fc_loss"]
    loss64.backward()
    arrs["loss64"] = npy(loss64)
    arrs["grad_norms64"] = np.array([0.0 if p.grad is None else float(p.grad.norm())
                                     for k, p in sorted(m64.named_parameters())])
    arrs["g_head_w"] = npy(m.head[0].weight.grad)
    arrs["g_conv1_w"] = npy(hr.conv1.weight.grad)
    arrs["g_s2_q"] = npy(hr.stage2[0].transformer.attn.attn.q_proj.weight.grad)
    arrs["rm_bn1"] = npy(hr.bn1.running_mean)
    arrs["rv_bn1"] = npy(hr.bn1.running_var)
    # eval-mode probabilities on the same input (after the train step's running-stat update)
    m.eval()
    with torch.no_grad():
        pr = m(x)
    arrs["eval_probs_sample"] = npy(pr[:, :, ::max(1, S // 8), ::max(1, S // 8)])
    arrs["eval_argmax_hist"] = np.bincount(npy(pr.argmax(1)).ravel(), minlength=6)
    save(f"model_{tag}", **arrs)


def case_eval():
    """Eval path of the reference (eval.py:55-71): model.eval() -> softmax probabilities, test-time augmentation over
    Scale transforms (module/tta.py:12-24,118-135), argmax, ignore(-1) mask, confusion matrix."""
    from module.tta import tta, Scale
    torch.manual_seed(0)
    m = load_seeded(build_model("tiny")).eval()
    x = seeded_input((2, 3, 64, 64), 11)
    y = proc_labels(2, 64, 64, 6, 9)
    with torch.no_grad():
        probs = m(x)
        scales = (0.5, 1.0, 1.5)
        out = tta(m, x, tta_config=[Scale(scale_factor=s) for s in scales])
    pred, pred_tta = probs.argmax(1), out.argmax(1)
    keep = y != -1
    k = 6
    cm = torch.bincount(y[keep] * k + pred[keep], minlength=k * k).reshape(k, k)
    cm_tta = torch.bincount(y[keep] * k + pred_tta[keep], minlength=k * k).reshape(k, k)
    save("eval_tiny_2x64", probs=npy(probs), tta=npy(out), scales=np.array(scales), y=npy(y), pred=npy(pred), pred_tta=npy(pred_tta),
         cm=npy(cm), cm_tta=npy(cm_tta))


def case_cam():
    """BASELINE config 5's conv-only relative: WaveCAM ResNet-50 CAM inference (net/resnet50_cam.py:112-126) on a synthetic
    VOC-sized 321 x 321 image and its horizontal flip.  The constructor downloads ImageNet weights (resnet50.py:106-111): no
    network here, so model_zoo.load_url is replaced by an empty dict (load_state_dict(strict=False) then keeps the init) and the
    seeded weights are loaded on top."""
    import importlib
    wc = "/root/reference/WaveCAM-TMM2023"
    sys.path.insert(0, wc)
    for k in [k for k in sys.modules if k == "net" or k.startswith("net.") or k == "misc" or k.startswith("misc.")]:
        del sys.modules[k]
    r50 = importlib.import_module("net.resnet50")
    r50.model_zoo.load_url = lambda url: {}
    cam_mod = importlib.import_module("net.resnet50_cam")
    from oracle import cam_cpu
    torch.manual_seed(0)
    m = cam_mod.CAM(stride=16, n_classes=20)
    m.eval()               # (the reference's Net.train() override returns None: no chaining)
    sd = m.state_dict()
    # the reference registers every backbone module three times (resnet50.*, stage*.*, backbone.*): seed the canonical copy and
    # let the aliases follow (they are the same tensors)
    canon = {k: v for k, v in sd.items() if k.startswith("resnet50.") or k == "classifier.weight"}
    seeded = seeded_state(canon, 4321)
    with torch.no_grad():
        for k, v in seeded.items():
            sd[k].copy_(v)
    x1 = seeded_input((1, 3, 321, 321), 21)
    x = torch.cat([x1, x1.flip(-1)], 0)
    with torch.no_grad():
        out = m(x)
        sep = m(x, separate=True)
    P = {k: v.clone() for k, v in seeded.items()}
    chk = cam_cpu.cam_forward(x, P)
    assert torch.allclose(chk, out, rtol=1e-4, atol=1e-5), float((chk - out).abs().max())
    save("cam_r50_321", cams=npy(out), sep_sample=npy(sep[:, :, ::4, ::4]), sum_sep=npy(sep.double().sum()), keys=np.array(sorted(canon.keys())),
         all_keys=np.array(list(sd.keys())))
    sys.path.remove(wc)


def case_scd():
    """BASELINE config 5 as worded: SCD's TSCD(mit_b1, stride [4, 2, 2, 1]) class-activation path on synthetic VOC-sized 321 x 321
    images (configs/voc_attn_reg.yaml:1-4,28-33; network/TSCD_model.py:66-79; utils/camutils.py:85-113).  timm / mmcv are not
    vendored: the stubs under oracle/refimport/stubs stand in (mmcv's ConvModule only feeds the decoder output cam_only discards)."""
    import importlib
    scd = "/root/reference/SCD-AAAI2023"
    sys.path.insert(0, scd)
    for k in [k for k in sys.modules if k in ("network", "utils") or k.startswith(("network.", "utils."))]:
        del sys.modules[k]
    tscd_mod = importlib.import_module("network.TSCD_model")
    from oracle import scd_cpu
    torch.manual_seed(0)
    m = tscd_mod.TSCD("mit_b1", num_classes=21, embedding_dim=256, stride=[4, 2, 2, 1], pretrained=False, pooling="gmp").eval()
    sd = m.state_dict()
    m.load_state_dict(seeded_state(sd, 777))
    P = {k: v.clone() for k, v in m.state_dict().items()}
    x = seeded_input((2, 3, 321, 321), 31)
    xc = torch.cat([x, x.flip(-1)], 0)
    with torch.no_grad():
        cam, attn = m(xc, cam_only=True)
        chk_cam, chk_attn = scd_cpu.tscd_cam_only(xc, P)
    assert torch.allclose(chk_cam, cam, rtol=1e-4, atol=1e-5), float((chk_cam - cam).abs().max())
    assert torch.allclose(chk_attn, attn, rtol=1e-4, atol=1e-6), float((chk_attn - attn).abs().max())
    # multi_scale_cam: the reference's own function (it needs pydensecrf / imageio at import: the one function is taken from the
    # module source without executing the module's imports of those)
    import types
    src = open(os.path.join(scd, "utils", "camutils.py")).read()
    start = src.index("def multi_scale_cam(")
    end = src.index("def multi_scale_cam_with_ref_mat(")
    ns = types.ModuleType("camutils_slice")
    ns.__dict__.update(torch=torch, F=torch.nn.functional)
    exec(compile(src[start:end], "camutils.py[multi_scale_cam]", "exec"), ns.__dict__)
    scales = [1, 0.5, 1.5]
    msc = ns.multi_scale_cam(m, x, scales)
    chk = scd_cpu.multi_scale_cam(P, x, scales)
    assert torch.allclose(chk, msc, rtol=1e-4, atol=1e-5), float((chk - msc).abs().max())
    with torch.no_grad():
        cls, seg, attns, pred = m(x)                       # the full forward at inference (eval: dropout off, running statistics)
        o_cls, o_seg, o_attns, o_pred = scd_cpu.tscd_full(x, P)
    for a_, b_ in [(o_cls, cls), (o_seg, seg), (o_pred, pred)] + list(zip(o_attns, attns)):
        assert a_.shape == b_.shape and torch.allclose(a_, b_, rtol=1e-4, atol=1e-5), float((a_ - b_).abs().max())
    full = dict(full_cls=npy(cls), full_seg_sample=npy(seg[:, :, ::3, ::3]), full_pred_sample=npy(pred[:, ::4, ::4]))
    for i, a_ in enumerate(attns):
        full[f"full_attn{i}_shape"] = np.array(a_.shape)
        st = 3 if a_.shape[-1] <= 100 else 7
        full[f"full_attn{i}_sample"] = npy(a_[:, :, ::st, ::st])
    save("scd_mitb1_321", **full, cam_s4=npy(cam), attn_sample=npy(attn[:, ::4, ::4]), attn_sum=npy(attn.double().sum((1, 2))),
         msc_sample=npy(msc[:, :, ::5, ::5]), msc_sum=npy(msc.double().sum((2, 3))), scales=np.array(scales),
         all_keys=np.array(list(sd.keys())), shapes=np.array([",".join(map(str, v.shape)) for v in sd.values()]))
    sys.path.remove(scd)
    for k in [k for k in sys.modules if k in ("network", "utils") or k.startswith(("network.", "utils."))]:
        del sys.modules[k]


def case_state_keys():
    for variant in ("tiny", "base", "large"):
        m = build_model(variant)
        sd = m.state_dict()
        save(f"keys_{variant}", names=np.array(list(sd.keys())),
             shapes=np.array([",".join(map(str, v.shape)) for v in sd.values()]),
             nparams=np.array(sum(p.numel() for p in m.parameters())))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["mhca", "attention", "block", "mlp", "loss", "neck_head", "keys", "models", "eval", "cam", "scd"]
    if "eval" in which: case_eval()
    if "mhca" in which: case_mhca()
    if "attention" in which: case_attention()
    if "block" in which: case_block()
    if "mlp" in which: case_mlp()
    if "loss" in which: case_loss()
    if "neck_head" in which: case_neck_head()
    if "keys" in which: case_state_keys()
    if "cam" in which: case_cam()
    if "scd" in which: case_scd()
    if "models" in which:
        case_model("tiny", 2, 256, "tiny_2x256")      # BASELINE config 1
        case_model("base", 2, 64, "base_2x64")
        case_model("large", 1, 64, "large_1x64")
    if "models" in which or "large256" in which:
        case_model("large", 1, 256, "large_1x256")    # C = 48 window attention over 10 x 10 windows (pad 64 -> 70), all four branches live
