"""Parameter tree of the SegFormer decoder the SCD model carries (reference: SCD-AAAI2023/network/segformer_head.py:12-81; its
`linear_fuse` is an mmcv ConvModule = Conv2d(bias=False) + SyncBN + ReLU registered as `conv` / `bn` / `activate`).

`TSCD(..., cam_only=True)` evaluates the decoder and discards the result (TSCD_model.py:71,77-79): the CAM path does not need it, so
only the parameters exist here - checkpoints of the reference load unchanged - and calling it says so."""
import torch.nn as nn


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
        raise NotImplementedError("SegFormerHead (HIP): the segmentation decoder is outside the CAM path this build covers")
