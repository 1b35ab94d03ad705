"""MlpDWBN (reference: modules/ffn_block.py:207-287): 1x1 conv -> SyncBN -> GELU -> {1x1 + 3x3 dil 6 + 3x3 dil 12}
dense convs summed -> SyncBN -> GELU -> 1x1 conv -> SyncBN -> GELU.

STATUS: the convolutions / batch-norms of this module currently dispatch to ATen-ROCm (MIOpen) on channels-last
tensors; the fused implicit-GEMM HIP kernels replace them (hot-path table row A7).  nn.SyncBatchNorm is
kept so the cross-rank statistics semantics of the reference hold under data parallelism."""
import torch.nn as nn


class MlpDWBN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, dw_act_layer=nn.GELU,
                 drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Conv2d(in_features, hidden_features, kernel_size=1)
        self.act1 = act_layer()
        self.norm1 = nn.SyncBatchNorm(hidden_features)
        self.dw = nn.Conv2d(hidden_features, hidden_features, 1, 1)
        self.dw6 = nn.Conv2d(hidden_features, hidden_features, 3, 1, padding=6, dilation=6)
        self.dw12 = nn.Conv2d(hidden_features, hidden_features, 3, 1, padding=12, dilation=12)
        self.act2 = dw_act_layer()
        self.norm2 = nn.SyncBatchNorm(hidden_features)
        self.fc2 = nn.Conv2d(hidden_features, out_features, kernel_size=1)
        self.act3 = act_layer()
        self.norm3 = nn.SyncBatchNorm(out_features)

    def forward_nhwc(self, t):
        """t: logical NCHW tensor in channels_last memory format."""
        t = self.act1(self.norm1(self.fc1(t)))
        t = self.dw(t) + self.dw6(t) + self.dw12(t)
        t = self.act2(self.norm2(t))
        return self.act3(self.norm3(self.fc2(t)))

    def forward(self, x, H, W):
        if x.dim() != 3:
            raise RuntimeError("Unsupported input shape: {}".format(x.shape))
        B, N, C = x.shape
        if N != H * W:
            raise RuntimeError("MlpDWBN (HIP): class-token inputs are not on the RSSFormer path")
        t = x.reshape(B, H, W, C).permute(0, 3, 1, 2)          # channels-last view, no copy
        return self.forward_nhwc(t).permute(0, 2, 3, 1).reshape(B, N, -1)
