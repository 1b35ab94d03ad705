"""GeneralTransformerBlock (reference: modules/MTFM.py:48-113).

x = low, y = high (NCHW, channels-last memory).  X <- X + Attn(LN1(X), LN1(Y)) ; X <- X + Mlp(LN2(X)).
LN1 (both streams), the saliency gate, pad/permute, Mhca and the first residual are ONE autograd node backed by
the fused HIP kernels; LN2 is a HIP kernel; the MLP is `MlpDWBN`."""
from functools import partial

import torch
import torch.nn as nn

from ..... import autograd as AG
from .ffn_block import MlpDWBN
from .multihead_isa_pool_attention import InterlacedPoolAttention2 as InterlacedPoolAttention


class GeneralTransformerBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, num_heads, window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6)):
        super().__init__()
        if drop_path != 0.0 or attn_drop != 0.0 or drop != 0.0:
            raise NotImplementedError("GeneralTransformerBlock (HIP): dropout / drop-path are 0 on the RSSFormer path")
        if inplanes != planes:
            raise NotImplementedError("GeneralTransformerBlock (HIP): inplanes == planes on the RSSFormer path")
        self.dim, self.out_dim, self.num_heads = inplanes, planes, num_heads
        self.window_size, self.mlp_ratio = window_size, mlp_ratio
        self.attn = InterlacedPoolAttention(self.dim, num_heads=num_heads, window_size=window_size, rpe=True,
                                            dropout=attn_drop)
        self.norm1 = norm_layer(self.dim)
        self.norm2 = norm_layer(self.out_dim)
        self.drop_path = nn.Identity()
        self.mlp = MlpDWBN(in_features=self.dim, hidden_features=int(self.dim * mlp_ratio), out_features=self.out_dim,
                           act_layer=act_layer, dw_act_layer=act_layer, drop=drop)

    def forward(self, x, y, mask=None, post_relu=False):
        """post_relu: relu() of the block's output inside the MLP's last BatchNorm pass (what HighResolutionModule.forward applies to
        it, _hrnet_rssformer.py:435) - no separate clamp launch forward, no mask launch backward."""
        B, C, H, W = x.shape
        xt = x.permute(0, 2, 3, 1).reshape(B, H * W, C)          # free when x is channels-last
        yt = y.permute(0, 2, 3, 1).reshape(B, H * W, C)
        if torch.is_autocast_enabled():                          # ATen ops upstream may hand over mixed fp32/bf16
            dt = torch.get_autocast_dtype("cuda")
            xt, yt = xt.to(dt), yt.to(dt)
        elif xt.dtype != yt.dtype:
            yt = yt.to(xt.dtype)
        x1 = AG.GatedWindowCrossAttention.apply(xt, yt, self.norm1.weight, self.norm1.bias, *self.attn.gate_params(),
                                                *self.attn.attn.proj_params(), H, W, self.num_heads)
        # LN2 and the skip around the MLP as one node: the two gradients of x1 meet in the LayerNorm-backward launch
        z, skip = AG.LayerNormTokensRes.apply(x1, self.norm2.weight, self.norm2.bias)
        x2 = self.mlp(z, H, W, residual=skip, post_relu=post_relu)        # x1 + Mlp(LN2(x1)), residual fused into the last BN/GELU pass
        return x2.reshape(B, H, W, C).permute(0, 3, 1, 2)

    def extra_repr(self):
        return "num_heads={}, window_size={}, mlp_ratio={}".format(self.num_heads, self.window_size, self.mlp_ratio)
