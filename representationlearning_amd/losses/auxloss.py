"""MCTransAuxLoss (reference: losses/auxloss.py:253-322): image-level "which of {background, foreground} occur"
targets vs the aux scores -> a per-sample scalar l1_b that modulates the focal term (no gradient to the aux head).
On the training path this arithmetic runs inside the fused HIP loss kernel (csrc/loss.hip); this module is the
stand-alone form of the same formula for callers that want l1 itself (tiny [B,7] tensors)."""
import torch
import torch.nn as nn


class MCTransAuxLoss(nn.Module):
    def forward(self, cls_score, label, **unused):
        """cls_score [B, K]; label [B, H, W] float {0,1} foreground map.  Returns (0.0, l1[B])."""
        B, K = cls_score.shape
        flat = label.flatten(1)
        target = torch.zeros_like(cls_score)
        target[:, 0] = (flat == 0).any(1).to(cls_score.dtype)      # unique() contains 0
        target[:, 1] = (flat == 1).any(1).to(cls_score.dtype)      # unique() contains 1
        l1 = (1.0 / (1.0 + torch.exp((cls_score - target).abs()))).sum(1) / (2 * B)     # local batch size (:291-292)
        return 0.0, l1
