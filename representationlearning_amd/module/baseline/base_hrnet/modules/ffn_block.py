"""MlpDWBN (reference: modules/ffn_block.py:207-287): 1x1 conv -> SyncBN -> GELU -> {1x1 + 3x3 dil 6 + 3x3 dil 12}
dense convs summed -> SyncBN -> GELU -> 1x1 conv -> SyncBN -> GELU.

Three fused launches-groups on the hand-written HIP kernels: fc1 -> BN -> GELU; the three hidden convolutions as ONE
19-tap implicit GEMM (K = 19*4C) -> BN -> GELU; fc2 -> BN -> GELU (+ the block's residual).  nn.SyncBatchNorm modules
are kept (state_dict + "always synchronised" semantics of the reference under data parallelism)."""
import os

import torch.nn as nn

from ..... import nnf

_DEFER_NORM2 = os.environ.get("RSSF_DEFER_MLP_APPLY", "1") != "0"       # A/B switch: norm2 + GELU applied by fc2's kernels on load


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

    def forward_nhwc(self, t, residual=None, post_relu=False):
        """t: logical NCHW tensor in channels_last memory; residual (optional) is added AFTER the last GELU; post_relu: a ReLU on that
        sum (the one HighResolutionModule.forward puts behind the transformer block, _hrnet_rssformer.py:435) in the same pass."""
        for a in (self.act1, self.act2, self.act3):
            if not isinstance(a, nn.GELU):
                raise NotImplementedError("MlpDWBN (HIP): GELU activations only (the RSSFormer configuration)")
        # each hidden activation has ONE consumer: its BatchNorm-backward statistics ride on that consumer's data-gradient launch
        l1, l2 = nnf.bwd_stats_link(), nnf.bwd_stats_link()
        if l1 is not None:
            # the tap sum's weight gradient reads its input TRANSPOSED (csrc/conv_wgrad_planes.hip): norm1's apply writes that copy too -
            # where that kernel takes the shape (the apply's own shape test is wider: a copy nobody reads would be written)
            l1.want_planes = nnf.wgrad_planes_ok(t, self.fc1.out_channels, [self.dw, self.dw6, self.dw12])
        t = nnf.conv_bn_act(t, self.fc1, self.norm1, nnf.ACT_GELU, stats_out=l1)
        # GELU(norm2(sum)) is consumed by fc2 alone: where fc2's kernels can apply norm2 + GELU on the raw sum while they load it, that
        # activation is never written (nnf.can_defer_apply: one 67 MB write + read less per block and direction)
        defer = _DEFER_NORM2 and l2 is not None and nnf.can_defer_apply(t, [self.dw, self.dw6, self.dw12], self.fc2)
        t = nnf.conv_bn_act(t, [self.dw, self.dw6, self.dw12], self.norm2, nnf.ACT_GELU, stats_out=l2, stats_in=l1, defer_apply=defer)
        return nnf.conv_bn_act(t, self.fc2, self.norm3, nnf.ACT_GELU, res_post=residual, stats_in=l2, post_relu=post_relu)

    def forward(self, x, H, W, residual=None, post_relu=False):
        if x.dim() != 3:
            raise RuntimeError("Unsupported input shape: {}".format(x.shape))
        B, N, C = x.shape
        if N != H * W:
            raise RuntimeError("MlpDWBN (HIP): class-token inputs are not on the RSSFormer path")
        t = x.reshape(B, H, W, C).permute(0, 3, 1, 2)          # channels-last view, no copy
        r = None if residual is None else residual.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        return self.forward_nhwc(t, r, post_relu).permute(0, 2, 3, 1).reshape(B, N, -1)
