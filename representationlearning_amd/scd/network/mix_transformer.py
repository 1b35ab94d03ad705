"""Mix-Transformer encoder (SegFormer MiT-B0..B5) of the SCD code-drop on librssf kernels, inference only.

Reference: SCD-AAAI2023/network/mix_transformer.py - Mlp :17-52, Attention :55-131, Block :133-171, OverlapPatchEmbed :174-212,
MixVisionTransformer :215-375, DWConv :377-388, mit_b0..b5 :390-435.  Same module tree, hence the same `state_dict` keys; the
computation is re-laid out for the GPU: tokens stay channels-last [B, H, W, C] from the patch embedding to the stage output
(the reference's flatten / transpose / reshape pairs are views of that one layout), every Linear is a 1 x 1 `rssf_conv_gather`
with its bias and - for `proj` and `fc2` - the residual in the epilogue, the strided `sr` / patch-embedding convolutions are
tap-split gather launches, DWConv + GELU is one `rssf_dwconv3x3` pass and q k^T -> softmax -> v is `rssf_mha_fwd`
(keys / values staged once per 64 queries, probabilities never leave registers).  Only the raw attention logits the caller asks
for are written (TSCD reads those of the last two blocks)."""
import math
from functools import partial

import torch
import torch.nn as nn

from ... import nnf, ops

GELU = 2            # activation code of rssf_dwconv3x3 / the BatchNorm passes


def _layer_norm(xh, ln):
    y, _ = ops.layernorm_fwd(xh, ln.weight.detach(), ln.bias.detach(), eps=ln.eps)
    return y


def _init(m):
    """The reference's `_init_weights` rule (every class there carries a copy): truncated-normal Linear weights, unit LayerNorm,
    fan-out-scaled normal convolutions, zero biases."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
    if isinstance(m, (nn.Linear, nn.LayerNorm, nn.Conv2d)) and m.bias is not None:
        nn.init.zeros_(m.bias)


class DWConv(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, xh, act=0):
        c = self.dwconv
        return ops.dwconv3x3(xh, c.weight.detach(), None if c.bias is None else c.bias.detach(), act)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        if act_layer is not nn.GELU or drop != 0.0:
            raise NotImplementedError("Mlp (HIP): GELU without dropout is what the MiT configurations use")
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.dwconv = DWConv(hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)
        self.apply(_init)

    def forward(self, xh, residual=None):
        """fc2(GELU(DWConv(fc1(x)))) (+ residual in fc2's epilogue)."""
        return nnf.conv_nhwc(self.dwconv(nnf.conv_nhwc(xh, self.fc1), GELU), self.fc2, addend=residual)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, sr_ratio=1):
        super().__init__()
        if dim % num_heads or attn_drop != 0.0 or proj_drop != 0.0:
            raise NotImplementedError("Attention (HIP): dim divisible by num_heads, no dropout")
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)
        self.apply(_init)

    def forward(self, xh, residual=None, want=None):
        """xh: LayerNorm'ed tokens [B, H, W, C].  Returns (proj(attention) (+ residual), extra): extra = the raw q k^T products
        [B, heads, N, M] for want == "logits" (what the reference hands back when sr_ratio == 1, :119-131), the projections
        (q [B, N, C], kv [B, M, 2C]) for want == "qkv" (all TSCD's attention head needs), else None.  For the spatially reduced
        stages "logits" is the reference's pooled copy [B, heads, M, M] (:121-129)."""
        B, H, W, C = xh.shape
        q = nnf.conv_nhwc(xh, self.q)
        src = xh
        if self.sr_ratio > 1:
            src = _layer_norm(nnf.conv_nhwc(xh, self.sr), self.norm)
        kv = nnf.conv_nhwc(src, self.kv)
        qt, kv = q.view(B, H * W, C), kv.view(B, -1, 2 * C)
        pooled = want == "logits" and self.sr_ratio > 1
        o, logits = ops.mha_fwd(qt, kv, self.num_heads, self.scale, want == "logits" and not pooled)
        if pooled:
            # avg_pool3d(raw logits as [B, heads, H, W, M], (sr, sr, 1)) (:121-124) = the products of the POOLED queries with the
            # same keys (the mean over a patch of q.k is (mean q).k): the [B, heads, N, M] tensor is never formed
            qp = torch.nn.functional.avg_pool2d(q.permute(0, 3, 1, 2).float(), self.sr_ratio, self.sr_ratio).permute(0, 2, 3, 1)
            qp = qp.reshape(B, -1, C).to(q.dtype).contiguous()
            logits = ops.mha_fwd(qp, kv, self.num_heads, self.scale, True)[1].reshape(-1, self.num_heads, kv.shape[1], kv.shape[1])
        return nnf.conv_nhwc(o.view(B, H, W, C), self.proj, addend=residual), (qt, kv) if want == "qkv" else logits


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop,
                              sr_ratio=sr_ratio)
        self.drop_path = nn.Identity()          # stochastic depth is the identity at inference; the rate is kept for the record
        self.drop_path.drop_prob = drop_path
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.apply(_init)

    def forward(self, xh, want=None):
        if self.training:
            raise NotImplementedError("Block (HIP): inference only - call .eval() (the CAM extraction of the reference runs under no_grad)")
        xh, logits = self.attn(_layer_norm(xh, self.norm1), residual=xh, want=want)
        return self.mlp(_layer_norm(xh, self.norm2), residual=xh), logits


class OverlapPatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.H, self.W = img_size // patch_size, img_size // patch_size
        self.num_patches = self.H * self.W
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)
        self.apply(_init)

    def forward(self, xh):
        """channels-last [B, H, W, Cin] -> LayerNorm'ed tokens [B, H', W', C] (the reference returns (tokens, H', W'))."""
        return _layer_norm(nnf.conv_nhwc(xh, self.proj), self.norm)


class MixVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=(64, 128, 256, 512), num_heads=(1, 2, 4, 8),
                 mlp_ratios=(4, 4, 4, 4), qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                 norm_layer=nn.LayerNorm, depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1), stride=None):
        super().__init__()
        self.num_classes, self.depths, self.embed_dims, self.stride = num_classes, list(depths), list(embed_dims), stride
        chans = [in_chans] + list(embed_dims)
        for i in range(4):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(img_size=img_size // (1, 4, 8, 16)[i], patch_size=7 if i == 0 else 3,
                                                                  stride=stride[i], in_chans=chans[i], embed_dim=chans[i + 1]))
        rates = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        for i in range(4):
            first = sum(depths[:i])
            setattr(self, f"block{i + 1}", nn.ModuleList(
                Block(dim=embed_dims[i], num_heads=num_heads[i], mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                      attn_drop=attn_drop_rate, drop_path=rates[first + j], norm_layer=norm_layer, sr_ratio=sr_ratios[i])
                for j in range(depths[i])))
            setattr(self, f"norm{i + 1}", norm_layer(embed_dims[i]))
        self.apply(_init)

    def forward_features(self, x, last=0, want="logits"):
        """x: image batch [B, 3, H, W] fp32 (any memory format).  Returns (the four stage outputs as channels-last NCHW views, one
        entry per block: for the last `last` blocks (None = all, what the reference returns) what `want` names - "logits": the
        attention products as the reference hands them back, "qkv": the (q, kv) projections - and None for the earlier ones)."""
        if self.training:
            raise NotImplementedError("MixVisionTransformer (HIP): inference only - call .eval()")
        dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else torch.float32
        xh = nnf.image_to_channels_last(x.float(), dtype).permute(0, 2, 3, 1)
        nblocks, k = sum(self.depths), 0
        last = nblocks if last is None else last
        outs, attns = [], []
        for i in range(4):
            xh = getattr(self, f"patch_embed{i + 1}")(xh)
            for blk in getattr(self, f"block{i + 1}"):
                xh, a = blk(xh, want=want if k >= nblocks - last else None)
                attns.append(a)
                k += 1
            xh = _layer_norm(xh, getattr(self, f"norm{i + 1}"))
            outs.append(xh.permute(0, 3, 1, 2))
        return outs, attns

    def forward(self, x, last=0, want="logits"):
        return self.forward_features(x, last, want)


def _variant(embed_dims, depths):
    def build(stride=None, **kwargs):
        return MixVisionTransformer(patch_size=4, embed_dims=embed_dims, num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=True,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=depths, sr_ratios=(8, 4, 2, 1), drop_rate=0.0,
                                    drop_path_rate=0.1, stride=stride)
    return build


mit_b0 = _variant((32, 64, 160, 256), (2, 2, 2, 2))
mit_b1 = _variant((64, 128, 320, 512), (2, 2, 2, 2))
mit_b2 = _variant((64, 128, 320, 512), (3, 4, 6, 3))
mit_b3 = _variant((64, 128, 320, 512), (3, 4, 18, 3))
mit_b4 = _variant((64, 128, 320, 512), (3, 8, 27, 3))
mit_b5 = _variant((64, 128, 320, 512), (3, 6, 40, 3))
