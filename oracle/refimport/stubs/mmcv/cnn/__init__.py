"""Stand-in for the un-vendored `mmcv.cnn.ConvModule` import of SCD-AAAI2023/network/segformer_head.py:10 (golden generation only).
mmcv's ConvModule(in, out, k, norm_cfg=dict(type='SyncBN')) registers `conv` (bias-free when a norm follows), `bn` and `activate`
(ReLU).  The decoder that uses it is evaluated and DISCARDED on the cam_only path the goldens pin (TSCD_model.py:71,77-79), so
only the parameter names matter here; BatchNorm2d stands for SyncBN (single process)."""
import torch.nn as nn


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=norm_cfg is None)
        if norm_cfg is not None:
            self.bn = nn.BatchNorm2d(out_channels)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if hasattr(self, "bn"):
            x = self.bn(x)
        return self.activate(x)
