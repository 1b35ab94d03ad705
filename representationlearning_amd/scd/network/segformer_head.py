"""SegFormer all-MLP decoder of the SCD model, inference (reference: SCD-AAAI2023/network/segformer_head.py:12-81; its `linear_fuse`
is an mmcv ConvModule = Conv2d(bias=False) + SyncBN + ReLU registered as `conv` / `bn` / `activate`).

`TSCD(..., cam_only=True)` evaluates the decoder and discards the result (TSCD_model.py:71,77-79), so the CAM path skips it; the
full forward runs it on librssf launches: four 1 x 1 projections, three align_corners=False up-samplings to the stride-4 grid,
concat, 1 x 1 convolution + BatchNorm (running statistics) + ReLU, 1 x 1 prediction."""
import torch
import torch.nn as nn

from ... import nnf, ops


class MLP(nn.Module):
    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)


class _FuseConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.activate = nn.ReLU(inplace=True)


class SegFormerHead(nn.Module):
    def __init__(self, feature_strides=None, in_channels=128, embedding_dim=256, num_classes=20, **kwargs):
        super().__init__()
        self.in_channels, self.num_classes, self.feature_strides = in_channels, num_classes, feature_strides
        for i in (4, 3, 2, 1):
            setattr(self, f"linear_c{i}", MLP(input_dim=in_channels[i - 1], embed_dim=embedding_dim))
        self.dropout = nn.Dropout2d(0.1)
        self.linear_fuse = _FuseConv(embedding_dim * 4, embedding_dim)
        self.linear_pred = nn.Conv2d(embedding_dim, num_classes, kernel_size=1)

    def forward(self, feats):
        """feats: the four stage outputs (channels-last NCHW views).  Returns the class logits [B, num_classes, H/4, W/4]."""
        if self.training:
            raise NotImplementedError("SegFormerHead (HIP): inference only - call .eval()")
        hs = [f.permute(0, 2, 3, 1) for f in feats]
        size = hs[0].shape[1:3]
        parts = []
        for i in (4, 3, 2, 1):                                                  # the reference concatenates [_c4, _c3, _c2, _c1]
            t = nnf.conv_nhwc(hs[i - 1], getattr(self, f"linear_c{i}").proj)
            parts.append(t if i == 1 else ops.resize_bilinear_nhwc(t, size))
        fused = nnf.conv_bn_act(torch.cat(parts, dim=3).permute(0, 3, 1, 2), self.linear_fuse.conv, self.linear_fuse.bn, nnf.ACT_RELU)
        return nnf.conv_nhwc(fused.permute(0, 2, 3, 1), self.linear_pred).permute(0, 3, 1, 2)       # Dropout2d: identity at inference
