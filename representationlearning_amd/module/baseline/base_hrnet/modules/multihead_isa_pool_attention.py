"""InterlacedPoolAttention2 + SpatialAttention (reference: modules/multihead_isa_pool_attention.py:101-188).
Parameter containers with the reference's attribute names; forward runs the HIP gate + fused window attention."""
import torch.nn as nn

from ..... import autograd as AG
from .DAL import Mhca as MHA_
from .multihead_isa_attention import LocalPermuteModule, PadBlock


class SpatialAttention(nn.Module):
    """Holds the 7x7 (2->1, no bias) conv of one saliency gate; evaluated inside rssf_gate_weights_fwd."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 7), "kernel size must be 3 or 7"
        if kernel_size != 7:
            raise NotImplementedError("SpatialAttention (HIP): the RSSFormer path uses kernel_size=7")
        self.conv1 = nn.Conv2d(2, 1, kernel_size, padding=3, bias=False)


class InterlacedPoolAttention2(nn.Module):
    def __init__(self, embed_dim, num_heads, window_size=7, rpe=True, **kwargs):
        super().__init__()
        if window_size != 7:
            raise NotImplementedError("InterlacedPoolAttention2 (HIP): built for 7x7 windows")
        self.dim, self.num_heads, self.window_size, self.with_rpe = embed_dim, num_heads, window_size, rpe
        self.attn = MHA_(embed_dim, num_heads, **kwargs)      # rpe/window_size are swallowed here, as in the reference
        self.pad_helper = PadBlock(window_size)
        self.permute_helper = LocalPermuteModule(window_size)
        self.atrous_block1 = SpatialAttention(7)
        self.atrous_block2 = SpatialAttention(7)
        self.weight_levels = nn.Conv2d(2, 2, kernel_size=1, stride=1, padding=0)

    def gate_params(self):
        return (self.atrous_block1.conv1.weight, self.atrous_block2.conv1.weight,
                self.weight_levels.weight, self.weight_levels.bias)

    def forward(self, x, y, H, W, **kwargs):
        """x, y: normalised tokens [B, N, C] -> attention term [B, N, C]."""
        return AG.PrenormedGatedWindowCrossAttention.apply(x, y, *self.gate_params(), *self.attn.proj_params(),
                                                           H, W, self.num_heads)
