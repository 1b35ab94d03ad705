"""Pins the CPU oracle (oracle/rssformer_cpu.py) to the golden vectors emitted by the real
reference (oracle/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import rssformer_cpu as O
from oracle.procedural import proc_input, proc_labels, seeded_input, seeded_state
from tests.helpers import golden, proc_params, rel_err, seeded_params

TOL = 2e-5   # fp32 CPU vs fp32 CPU (different op order only)


def _sub(P, pre):
    return {k: v for k, v in P.items() if k.startswith(pre)}


def _mhca_template(C):
    t = O.block_template(C)
    return {k[len("attn.attn."):]: v for k, v in t.items() if k.startswith("attn.attn.")}


@pytest.mark.parametrize("C,nw,tag", [(32, 3, "c32"), (18, 2, "c18"), (48, 2, "c48")])
def test_mhca(C, nw, tag):
    g = golden(f"mhca_{tag}")
    P = proc_params(_mhca_template(C))
    x = proc_input((49, nw, C), 0.3).requires_grad_()
    y = proc_input((49, nw, C), 1.1).requires_grad_()
    out = O.mhca(x, y, P, "")
    (out * proc_input(out.shape, 2.0)).sum().backward()
    assert rel_err(out.detach(), g["out"]) < TOL
    assert rel_err(x.grad, g["gx"]) < TOL
    assert rel_err(y.grad, g["gy"]) < TOL
    for k, p in P.items():
        assert rel_err(p.grad, g["g_" + k.replace(".", "_")]) < 5e-5, k


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 7, 7), (2, 32, 10, 10), (1, 32, 14, 14), (1, 32, 20, 12),
                                     (1, 18, 9, 11), (1, 48, 8, 8)])
def test_interlaced_attention(B, C, H, W):
    g = golden(f"attn_B{B}_C{C}_H{H}_W{W}")
    t = {k[len("attn."):]: v for k, v in O.block_template(C).items() if k.startswith("attn.")}
    P = proc_params(t)
    x = proc_input((B, H * W, C), 0.2).requires_grad_()
    y = proc_input((B, H * W, C), 0.8).requires_grad_()
    _, _, _, lv = O.gate(x, y, P, "", H, W)
    assert rel_err(lv.detach(), g["gate_logits"]) < TOL
    out = O.interlaced_attention(x, y, P, "", H, W)
    (out * proc_input(out.shape, 1.7)).sum().backward()
    assert rel_err(out.detach(), g["out"]) < TOL
    assert rel_err(x.grad, g["gx"]) < 5e-5
    assert rel_err(y.grad, g["gy"]) < 5e-5
    for k, p in P.items():
        assert rel_err(p.grad, g["g_" + k.replace(".", "_")]) < 1e-4, k


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 10, 10), (2, 32, 14, 9), (1, 18, 8, 8)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_transformer_block(B, C, H, W, mode):
    g = golden(f"block_{mode}_B{B}_C{C}_H{H}_W{W}")
    P = seeded_params(O.block_template(C))
    low = seeded_input((B, C, H, W), 11).requires_grad_()
    high = seeded_input((B, C, H, W), 12).requires_grad_()
    out = O.transformer_block(low, high, P, "", mode == "train")
    assert rel_err(out.detach(), g["out"]) < TOL
    if mode == "train":
        out.square().mean().backward()
        assert rel_err(low.grad, g["glow"]) < 1e-4
        assert rel_err(high.grad, g["ghigh"]) < 1e-4
        for k, p in P.items():
            key = "g_" + k.replace(".", "_")
            if key in g.files:
                assert rel_err(p.grad, g[key]) < 2e-4, k
            key = "b_" + k.replace(".", "_")
            if key in g.files:
                assert rel_err(p, g[key]) < TOL, k


def test_kat_survey():
    """Known-answer values recorded in SURVEY.md §8c for block(32,32,2), [1,32,10,10]."""
    P = proc_params(O.block_template(32), requires_grad=False)
    low, high = proc_input((1, 32, 10, 10), 0.1), proc_input((1, 32, 10, 10), 0.9)
    out = O.transformer_block(low, high, P, "", False)
    assert abs(float(out.sum()) - 6.694829) < 2e-3
    assert abs(float(out.abs().mean()) - 0.638839) < 1e-5
    np.testing.assert_allclose(out[0, 0, 0, :4].numpy(), [0.0288186, 0.3879926, 0.6827021, 0.8715582], atol=2e-5)


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 16, 13), (1, 18, 30, 30)])
def test_mlp(B, C, H, W):
    g = golden(f"mlp_B{B}_C{C}_H{H}_W{W}")
    t = {k[len("mlp."):]: v for k, v in O.block_template(C).items() if k.startswith("mlp.")}
    P = seeded_params(t)
    z = seeded_input((B, H * W, C), 21).requires_grad_()
    out = O.mlp_dwbn(z, P, "", H, W, True)
    (out * seeded_input(out.shape, 22)).sum().backward()
    assert rel_err(out.detach(), g["out"]) < TOL
    assert rel_err(z.grad, g["gz"]) < 1e-4
    for k, p in P.items():
        for pre in ("g_", "b_"):
            key = pre + k.replace(".", "_")
            if key in g.files:
                val = p.grad if pre == "g_" else p
                assert rel_err(val, g[key]) < 2e-4, k


@pytest.mark.parametrize("tag", ["mixed", "one_all_ignore", "no_fg", "all_fg"])
def test_loss(tag):
    g = golden(f"loss_{tag}")
    y = torch.from_numpy(g["y"])
    lg = (proc_input((3, 6, 12, 10), 0.4) * 2.0).requires_grad_()
    aux = proc_input((3, 7), 1.3).requires_grad_()
    loss = O.cgfl_loss(lg, y, aux)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6 * max(1.0, abs(float(g["loss"])))
    assert rel_err(lg.grad, g["glogits"]) < TOL
    assert aux.grad is None or float(aux.grad.abs().max()) == 0.0   # headaux gets no gradient


def test_neck_head():
    g = golden("neck_head")
    P = proc_params(O.model_template("base"))
    feats = [proc_input((2, c, 12 // s, 8 // s), 0.3 * i) for i, (c, s) in
             enumerate(((32, 1), (64, 2), (128, 4), (256, 4)))]
    feats[3] = proc_input((2, 256, 2, 1), 0.9)
    feats = [f.requires_grad_() for f in feats]
    lg, aux = O.neck_head(feats, P, True)
    (lg * proc_input(lg.shape, 0.7)).sum().backward()
    assert rel_err(lg.detach(), g["logits"]) < TOL
    assert rel_err(aux.detach(), g["aux"]) < TOL
    for i, f in enumerate(feats):
        assert rel_err(f.grad, g[f"gf{i}"]) < 1e-4
    assert rel_err(P["head.0.weight"].grad, g["g_head_w"]) < 1e-4
    assert rel_err(P["neck.fuse_conv.0.weight"].grad.sum((2, 3)), g["g_neck_w_sum"]) < 1e-4


@pytest.mark.parametrize("variant", ["tiny", "base", "large"])
def test_state_dict_keys(variant):
    g = golden(f"keys_{variant}")
    t = O.model_template(variant)
    assert set(t.keys()) == set(g["names"].tolist())
    shapes = dict(zip(g["names"].tolist(), g["shapes"].tolist()))
    for k, v in t.items():
        assert ",".join(map(str, v.shape)) == shapes[k], k
    n = sum(v.numel() for k, v in t.items() if torch.is_floating_point(v) and "running" not in k)
    assert n == int(g["nparams"])


@pytest.mark.parametrize("variant,B,S,tag", [("tiny", 2, 256, "tiny_2x256"), ("base", 2, 64, "base_2x64"),
                                             ("large", 1, 64, "large_1x64"), ("large", 1, 256, "large_1x256")])
def test_full_model(variant, B, S, tag):
    """BASELINE config 1 (Tiny 2x3x256x256, one CPU train step) + small Base/Large."""
    g = golden(f"model_{tag}")
    P = seeded_params(O.model_template(variant))
    x = seeded_input((B, 3, S, S), 7)
    y = proc_labels(B, S, S, 6, 8)
    taps = {}
    loss = O.model_forward(x, P, True, y, taps)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    st = max(1, S // 8)
    assert rel_err(taps["logits"][:, :, ::st, ::st].detach(), g["logits_sample"]) < 1e-4
    assert rel_err(taps["aux"].detach(), g["aux"]) < 1e-4
    for k in ("layer1", "stage2", "stage3", "stage4", "logits"):
        assert abs(float(taps[k].double().abs().mean()) - float(g[f"abs_{k}"])) < 1e-4 * float(g[f"abs_{k}"]), k
    names = g["grad_names"].tolist()
    ref = dict(zip(names, g["grad_norms"].tolist()))
    # parameters feeding a BatchNorm through a bias (mlp.*.bias, norm2.bias, neck conv bias) have a
    # mathematically zero gradient: both sides hold roundoff noise there, so allow an absolute floor.
    floor = 2e-3 * float(np.median(g["grad_norms"]))
    bad = []
    for k in names:
        p = P[k]
        gn = 0.0 if p.grad is None else float(p.grad.double().norm())
        if abs(gn - ref[k]) > 2e-3 * ref[k] + floor:
            bad.append((k, gn, ref[k]))
    assert not bad, bad[:5]
    assert ref["headaux.0.weight"] == 0.0 and ref["headaux.0.bias"] == 0.0
    assert rel_err(P["head.0.weight"].grad, g["g_head_w"]) < 1e-3
    assert rel_err(P["backbone.hrnet.conv1.weight"].grad, g["g_conv1_w"]) < 2e-3
    assert rel_err(P["backbone.hrnet.stage2.0.transformer.attn.attn.q_proj.weight"].grad, g["g_s2_q"]) < 2e-3
    assert rel_err(P["backbone.hrnet.bn1.running_mean"], g["rm_bn1"]) < TOL
    assert rel_err(P["backbone.hrnet.bn1.running_var"], g["rv_bn1"]) < TOL
    with torch.no_grad():
        pr = O.model_forward(x, P, False)
    assert rel_err(pr[:, :, ::st, ::st], g["eval_probs_sample"]) < 1e-4
    hist = np.bincount(pr.argmax(1).numpy().ravel(), minlength=6)
    assert np.abs(hist - g["eval_argmax_hist"]).sum() <= 0.002 * hist.sum()


def test_eval_tta_confusion():
    """Eval path (eval.py:55-71, module/tta.py): probabilities, 3-scale TTA, argmax, ignore mask, confusion matrix."""
    g = golden("eval_tiny_2x64")
    P = seeded_params(O.model_template("tiny"))
    x = seeded_input((2, 3, 64, 64), 11)
    y = torch.from_numpy(g["y"])
    with torch.no_grad():
        probs = O.model_forward(x, P, False)
        out = O.tta_scales(lambda im: O.model_forward(im, P, False), x, g["scales"].tolist())
    assert rel_err(probs, g["probs"]) < 1e-4
    assert rel_err(out, g["tta"]) < 1e-4
    agree = (out.argmax(1).numpy() == g["pred_tta"]).mean()
    assert agree > 0.999, agree
    cm = O.confusion_matrix(y, torch.from_numpy(g["pred_tta"]), 6)
    assert np.array_equal(cm.numpy(), g["cm_tta"])
    assert np.array_equal(O.confusion_matrix(y, torch.from_numpy(g["pred"]), 6).numpy(), g["cm"])


def test_input_pipeline_oracle_known_answers():
    """oracle/input_cpu.py (parity unpinned - albumentations is not installed): hand-derived known answers for the index maps
    and the two-rounding normalisation, so that the checker of tests/test_gpu_input.py is itself pinned to something."""
    import numpy as np
    from oracle import input_cpu as IC
    a = np.arange(6, dtype=np.uint8).reshape(2, 3)                     # [[0 1 2] [3 4 5]]
    assert IC.geometric(a, IC.AUG_HFLIP).tolist() == [[2, 1, 0], [5, 4, 3]]
    assert IC.geometric(a, IC.AUG_VFLIP).tolist() == [[3, 4, 5], [0, 1, 2]]
    assert IC.geometric(a, IC.AUG_ROT90 + 0).tolist() == a.tolist()
    assert IC.geometric(a, IC.AUG_ROT90 + 1).tolist() == [[2, 5], [1, 4], [0, 3]]          # counter-clockwise quarter turn
    assert IC.geometric(a, IC.AUG_ROT90 + 2).tolist() == [[5, 4, 3], [2, 1, 0]]
    assert IC.geometric(a, IC.AUG_ROT90 + 3).tolist() == [[3, 0], [4, 1], [5, 2]]
    img = np.array([[[[0, 128, 255]]]], dtype=np.uint8)                # one pixel
    out, lab = IC.pipeline(img, np.array([[[0]]], dtype=np.uint8), [[0, 0, 0, 0]], 1, (123.675, 116.28, 103.53), (58.395, 57.12, 57.375))
    exp = [(np.float32(v) - np.float32(m)) * (np.float32(1) / np.float32(s)) for v, m, s in zip((0, 128, 255), (123.675, 116.28, 103.53), (58.395, 57.12, 57.375))]
    assert out.dtype == np.float32 and out[0, 0, 0].tolist() == [float(e) for e in exp]
    assert abs(out[0, 0, 0, 0] + 2.117904) < 1e-5 and abs(out[0, 0, 0, 2] - 2.64) < 1e-5          # the familiar ImageNet range
    assert lab.tolist() == [[[-1]]]                                    # LoveDA no-data 0 -> ignore -1


def test_dp_emulation_oracle_consistency():
    """model_forward_dp: one shard == the plain path; two shards pool the BatchNorm statistics but keep the
    loss normalisers local, so the mean of shard losses differs from ONE forward over the concatenated batch."""
    from oracle.procedural import proc_labels, seeded_input
    from tests.helpers import seeded_params
    x = seeded_input((2, 3, 32, 32), 3)
    y = proc_labels(2, 32, 32, 6, 4)
    P1 = seeded_params(O.model_template("tiny"), requires_grad=False)
    P2 = seeded_params(O.model_template("tiny"), requires_grad=False)
    one, _ = O.model_forward_dp([x], [y], P1)
    plain = O.model_forward(x, P2, True, y)
    assert abs(float(one) - float(plain)) < 1e-6 * abs(float(plain))
    P3 = seeded_params(O.model_template("tiny"), requires_grad=False)
    two, parts = O.model_forward_dp([x[:1], x[1:]], [y[:1], y[1:]], P3)
    assert abs(float(two) - 0.5 * (float(parts[0]) + float(parts[1]))) < 1e-6
    assert abs(float(two) - float(plain)) > 1e-4 * abs(float(plain))
    # pooled statistics: running buffers equal those of the full-batch forward
    k = "backbone.hrnet.bn1.running_var"
    assert torch.allclose(P3[k], P2[k], rtol=1e-6, atol=1e-7)


def test_cam_resnet50_oracle_vs_reference_golden():
    """oracle/cam_cpu.py (WaveCAM ResNet-50 CAM inference, BASELINE config 5's conv-only relative) against the output of the
    reference's own net.resnet50_cam.CAM on a 321 x 321 image and its flip."""
    from oracle import cam_cpu
    g = golden("cam_r50_321")
    t = cam_cpu.cam_template()
    assert sorted(t.keys()) == g["keys"].tolist()
    P = seeded_state(t, 4321)
    x1 = seeded_input((1, 3, 321, 321), 21)
    x = torch.cat([x1, x1.flip(-1)], 0)
    with torch.no_grad():
        out = cam_cpu.cam_forward(x, P)
        sep = cam_cpu.cam_forward(x, P, separate=True)
    assert out.shape == (20, 21, 21)
    assert rel_err(out, g["cams"]) < TOL and rel_err(sep[:, :, ::4, ::4], g["sep_sample"]) < TOL
    assert abs(float(sep.double().sum()) - float(g["sum_sep"])) < 1e-4 * abs(float(g["sum_sep"]))


def _scd_template():
    g = golden("scd_mitb1_321")
    return {k: torch.zeros([int(v) for v in s.split(",")] if s else [], dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32)
            for k, s in zip(g["all_keys"].tolist(), g["shapes"].tolist())}


def test_scd_cam_oracle_vs_reference_golden():
    """oracle/scd_cpu.py (SCD's TSCD(mit_b1) cam_only forward and multi_scale_cam: BASELINE config 5 as worded) against the outputs
    of the reference's own model and of its own multi_scale_cam on two 321 x 321 images."""
    from oracle import scd_cpu
    from oracle.procedural import seeded_input, seeded_state
    g = golden("scd_mitb1_321")
    P = seeded_state(_scd_template(), 777)
    x = seeded_input((2, 3, 321, 321), 31)
    with torch.no_grad():
        cam, attn = scd_cpu.tscd_cam_only(torch.cat([x, x.flip(-1)], 0), P)
        msc = scd_cpu.multi_scale_cam(P, x, g["scales"].tolist())
    assert rel_err(cam, g["cam_s4"]) < TOL and rel_err(attn[:, ::4, ::4], g["attn_sample"]) < TOL
    assert rel_err(attn.double().sum((1, 2)), g["attn_sum"]) < TOL
    assert rel_err(msc[:, :, ::5, ::5], g["msc_sample"]) < TOL and rel_err(msc.double().sum((2, 3)), g["msc_sum"]) < TOL
    with torch.no_grad():
        cls, seg, attns, pred = scd_cpu.tscd_full(x, P)
    assert rel_err(cls, g["full_cls"]) < TOL and rel_err(seg[:, :, ::3, ::3], g["full_seg_sample"]) < TOL
    assert rel_err(pred[:, ::4, ::4], g["full_pred_sample"]) < TOL
    for i, a in enumerate(attns):
        st = 3 if a.shape[-1] <= 100 else 7
        assert tuple(a.shape) == tuple(g[f"full_attn{i}_shape"]) and rel_err(a[:, :, ::st, ::st], g[f"full_attn{i}_sample"]) < TOL


def test_scd_state_dict_matches_reference():
    """Same module tree as the reference's TSCD: keys in the same order, same shapes (checkpoints load unchanged)."""
    from representationlearning_amd.scd.network.TSCD_model import TSCD
    g = golden("scd_mitb1_321")
    sd = TSCD("mit_b1", num_classes=21, embedding_dim=256, stride=[4, 2, 2, 1], pretrained=False, pooling="gmp").state_dict()
    assert list(sd.keys()) == g["all_keys"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == g["shapes"].tolist()
