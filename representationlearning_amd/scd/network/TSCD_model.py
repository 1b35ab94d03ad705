"""SCD's network: Mix-Transformer encoder + class-activation / attention heads + SegFormer decoder (reference:
SCD-AAAI2023/network/TSCD_model.py:10-88), inference.  `forward(x, cam_only=True)` - what `multi_scale_cam` calls
(utils/camutils.py:91,103) - and the full forward (class scores, segmentation logits, attention maps) run on librssf kernels: the class
activation map is the classifier's 1 x 1 weights applied to the stage-4 feature, the attention prediction is
sigmoid(attn_proj(raw q k^T of the last two blocks)), formed by `rssf_attn_pred` without the logit tensors.  Same module tree / `state_dict` as the reference."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import nnf, ops
from . import mix_transformer
from .segformer_head import SegFormerHead


class TSCD(nn.Module):
    def __init__(self, backbone, num_classes=None, embedding_dim=256, stride=None, pretrained=None, pooling=None):
        super().__init__()
        self.num_classes, self.embedding_dim, self.stride = num_classes, embedding_dim, stride
        self.feature_strides = [4, 8, 16, 32]
        self.encoder = getattr(mix_transformer, backbone)(stride=stride)
        self.in_channels = self.encoder.embed_dims
        if pretrained:
            if not isinstance(pretrained, str):
                raise FileNotFoundError("pretrained=True: pass the path of the ImageNet MiT checkpoint (the reference reads "
                                        "pretrained/<backbone>.pth, TSCD_model.py:22-26)")
            sd = torch.load(pretrained, map_location="cpu")
            for k in ("head.weight", "head.bias"):
                sd.pop(k, None)
            self.encoder.load_state_dict(sd)
        self.pooling = {"gmp": F.adaptive_max_pool2d, "gap": F.adaptive_avg_pool2d}.get(pooling)
        self.dropout = nn.Dropout2d(0.5)
        self.decoder = SegFormerHead(feature_strides=self.feature_strides, in_channels=self.in_channels, embedding_dim=embedding_dim,
                                     num_classes=num_classes)
        self.attn_proj = nn.Conv2d(16, 1, kernel_size=1, bias=True)
        nn.init.kaiming_normal_(self.attn_proj.weight, a=math.sqrt(5), mode="fan_out")
        self.classifier = nn.Conv2d(self.in_channels[3], num_classes - 1, kernel_size=1, bias=False)

    def get_param_groups(self):
        """[backbone, backbone norms, classification heads, segmentation decoder] (TSCD_model.py:44-62)."""
        groups = [[], [], [self.classifier.weight, self.attn_proj.weight, self.attn_proj.bias], list(self.decoder.parameters())]
        for name, p in self.encoder.named_parameters():
            groups[1 if "norm" in name else 0].append(p)
        return groups

    def _attn_pred(self, qkv):
        # sigmoid(attn_proj(cat(attns[-2:], 1)))[:, 0] from the projections of the last two blocks: the logit tensors are not formed
        return ops.attn_pred(*qkv[-2], *qkv[-1], self.attn_proj.weight.detach(), self.attn_proj.bias.detach(),
                             self.encoder.block4[-1].attn.num_heads)

    def forward(self, x, cam_only=False, seg_detach=True, aux=False):
        if self.training or torch.is_grad_enabled():
            raise NotImplementedError("TSCD (HIP): inference only - .eval() under torch.no_grad() (the class-activation extraction of "
                                      "the reference; its training step is not built)")
        if cam_only:
            feats, qkv = self.encoder(x, last=2, want="qkv")
            cam_s4 = nnf.conv_nhwc(feats[3].permute(0, 2, 3, 1), self.classifier).permute(0, 3, 1, 2)
            return cam_s4, self._attn_pred(qkv)
        # the full forward (TSCD_model.py:68-88): image-level class scores from the pooled stage-4 feature, the segmentation logits,
        # every block's attention products (pooled for the spatially reduced stages) and the attention prediction
        feats, attns = self.encoder(x, last=None, want="logits")
        seg = self.decoder(feats)
        if self.pooling is None:
            raise ValueError("TSCD: pooling='gmp' or 'gap' is needed for the classification output")
        pooled = self.pooling(feats[3].float(), (1, 1)).to(feats[3].dtype).permute(0, 2, 3, 1).contiguous()
        cls_x4 = nnf.conv_nhwc(pooled, self.classifier).reshape(-1, self.num_classes - 1)
        if aux:
            return cls_x4, seg, attns
        return cls_x4, seg, attns, ops.attn_proj_sigmoid(attns[-2], attns[-1], self.attn_proj.weight.detach(), self.attn_proj.bias.detach())
