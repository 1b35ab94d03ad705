"""CPU restatement (plain PyTorch, fp32) of the SCD class-activation-map path.  TEST INFRASTRUCTURE.

BASELINE config 5 as it is worded: `TSCD(backbone='mit_b1', stride=[4, 2, 2, 1])(x, cam_only=True)` and `multi_scale_cam` of
SCD-AAAI2023 (configs/voc_attn_reg.yaml:1-4, 28-33).  Functional style over a flat parameter dict keyed by the reference's
`state_dict` names.  Pinned against tests/golden/scd_mitb1_321.npz (oracle/make_golden.py::case_scd imports the reference itself;
its un-vendored imports - timm's DropPath / to_2tuple / trunc_normal_, mmcv's ConvModule - are the stand-ins under
oracle/refimport/stubs; the ConvModule only appears in the decoder whose output cam_only discards).
Citations are into /root/reference/SCD-AAAI2023/."""
import torch
import torch.nn.functional as F

MIT = {  # network/mix_transformer.py:390-435
    "mit_b0": dict(dims=(32, 64, 160, 256), depths=(2, 2, 2, 2)),
    "mit_b1": dict(dims=(64, 128, 320, 512), depths=(2, 2, 2, 2)),
    "mit_b2": dict(dims=(64, 128, 320, 512), depths=(3, 4, 6, 3)),
}
HEADS, SR, MLP_RATIO = (1, 2, 5, 8), (8, 4, 2, 1), 4


def _ln(x, P, pre, eps):
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], eps)


def attention(x, H, W, P, pre, heads, sr):
    """Attention.forward (mix_transformer.py:93-131).  x: LayerNorm'ed tokens [B, N, C].  Returns (projected output, the raw
    q k^T products [B, heads, N, M] the reference hands back as `attn_copy` when sr_ratio == 1)."""
    B, N, C = x.shape
    d = C // heads
    q = F.linear(x, P[pre + "q.weight"], P[pre + "q.bias"]).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    src = x
    if sr > 1:
        src = x.permute(0, 2, 1).reshape(B, C, H, W)
        src = F.conv2d(src, P[pre + "sr.weight"], P[pre + "sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        src = _ln(src, P, pre + "norm.", 1e-5)                     # nn.LayerNorm(dim): the default eps (:70)
    kv = F.linear(src, P[pre + "kv.weight"], P[pre + "kv.bias"]).reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    raw = q @ k.transpose(-2, -1)
    attn = (raw * d ** -0.5).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(out, P[pre + "proj.weight"], P[pre + "proj.bias"]), raw


def mlp(x, H, W, P, pre):
    """Mlp.forward (:45-52) with DWConv (:377-388)."""
    B, N, _ = x.shape
    x = F.linear(x, P[pre + "fc1.weight"], P[pre + "fc1.bias"])
    C = x.shape[-1]
    x = F.conv2d(x.transpose(1, 2).reshape(B, C, H, W), P[pre + "dwconv.dwconv.weight"], P[pre + "dwconv.dwconv.bias"], padding=1, groups=C)
    x = F.gelu(x.flatten(2).transpose(1, 2))
    return F.linear(x, P[pre + "fc2.weight"], P[pre + "fc2.bias"])


def block(x, H, W, P, pre, heads, sr):
    """Block.forward (:165-171), drop_path = identity at inference; norm_layer eps 1e-6 (:401)."""
    a, raw = attention(_ln(x, P, pre + "norm1.", 1e-6), H, W, P, pre + "attn.", heads, sr)
    x = x + a
    return x + mlp(_ln(x, P, pre + "norm2.", 1e-6), H, W, P, pre + "mlp."), raw


def encoder(x, P, backbone="mit_b1", stride=(4, 2, 2, 1), pre="encoder.", pooled_attn=False):
    """MixVisionTransformer.forward_features (:330-372): the four stage outputs [B, C, H, W] and the raw attention products of
    the sr_ratio == 1 blocks (None for the others: the reference returns pooled copies nothing on the CAM path reads)."""
    cfg = MIT[backbone]
    outs, raws = [], []
    for i in range(4):
        k = 7 if i == 0 else 3
        pe = f"{pre}patch_embed{i + 1}."
        x = F.conv2d(x, P[pe + "proj.weight"], P[pe + "proj.bias"], stride=stride[i], padding=k // 2)     # OverlapPatchEmbed (:205-212)
        B, C, H, W = x.shape
        x = _ln(x.flatten(2).transpose(1, 2), P, pe + "norm.", 1e-5)
        for j in range(cfg["depths"][i]):
            x, raw = block(x, H, W, P, f"{pre}block{i + 1}.{j}.", HEADS[i], SR[i])
            if SR[i] > 1 and pooled_attn:                                   # attn_copy of the reduced stages (:119-129)
                Bq, hd, _, M = raw.shape
                raw = F.avg_pool3d(raw.reshape(Bq, hd, H, W, M), kernel_size=(SR[i], SR[i], 1), stride=(SR[i], SR[i], 1)).reshape(-1, hd, M, M)
            raws.append(raw if (SR[i] == 1 or pooled_attn) else None)
        x = _ln(x, P, f"{pre}norm{i + 1}.", 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs, raws


def tscd_cam_only(x, P, backbone="mit_b1", stride=(4, 2, 2, 1)):
    """TSCD.forward(x, cam_only=True) (network/TSCD_model.py:66-79)."""
    feats, raws = encoder(x, P, backbone, stride)
    attn_cat = torch.cat(raws[-2:], dim=1)
    attn_pred = torch.sigmoid(F.conv2d(attn_cat, P["attn_proj.weight"], P["attn_proj.bias"]))[:, 0]
    return F.conv2d(feats[3], P["classifier.weight"]), attn_pred


def tscd_full(x, P, backbone="mit_b1", stride=(4, 2, 2, 1), pooling="gmp"):
    """TSCD.forward(x) at inference (TSCD_model.py:66-88): (class scores, segmentation logits, the attention products of every block -
    pooled over sr x sr query patches where sr_ratio > 1, mix_transformer.py:119-129 -, the attention prediction)."""
    feats, attns = encoder(x, P, backbone, stride, pooled_attn=True)
    c1, c2, c3, c4 = feats
    n = x.shape[0]
    ups = []
    for name, c in (("c4", c4), ("c3", c3), ("c2", c2), ("c1", c1)):                # SegFormerHead.forward (segformer_head.py:58-81)
        t = F.linear(c.flatten(2).transpose(1, 2), P[f"decoder.linear_{name}.proj.weight"], P[f"decoder.linear_{name}.proj.bias"])
        t = t.permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
        ups.append(t if name == "c1" else F.interpolate(t, size=c1.shape[2:], mode="bilinear", align_corners=False))
    f = F.conv2d(torch.cat(ups, 1), P["decoder.linear_fuse.conv.weight"])
    f = F.relu(F.batch_norm(f, P["decoder.linear_fuse.bn.running_mean"], P["decoder.linear_fuse.bn.running_var"],
                            P["decoder.linear_fuse.bn.weight"], P["decoder.linear_fuse.bn.bias"], False, 0.0, 1e-5))
    seg = F.conv2d(f, P["decoder.linear_pred.weight"], P["decoder.linear_pred.bias"])
    pool = F.adaptive_max_pool2d if pooling == "gmp" else F.adaptive_avg_pool2d
    cls = F.conv2d(pool(c4, (1, 1)), P["classifier.weight"]).view(-1, P["classifier.weight"].shape[0])
    pred = torch.sigmoid(F.conv2d(torch.cat(attns[-2:], 1), P["attn_proj.weight"], P["attn_proj.bias"]))[:, 0]
    return cls, seg, attns, pred


def multi_scale_cam(P, inputs, scales, backbone="mit_b1", stride=(4, 2, 2, 1)):
    """utils/camutils.py:85-113."""
    b, c, h, w = inputs.shape

    def one(x):
        cam, _ = tscd_cam_only(torch.cat([x, x.flip(-1)], 0), P, backbone, stride)
        cam = F.interpolate(cam, size=(h, w), mode="bilinear", align_corners=False)
        return F.relu(torch.max(cam[:b], cam[b:].flip(-1)))

    cams = [one(inputs)]
    for s in scales:
        if s != 1.0:
            cams.append(one(F.interpolate(inputs, size=(int(s * h), int(s * w)), mode="bilinear", align_corners=False)))
    cam = torch.sum(torch.stack(cams, 0), 0)
    cam = cam + F.adaptive_max_pool2d(-cam, (1, 1))
    return cam / (F.adaptive_max_pool2d(cam, (1, 1)) + 1e-5)
