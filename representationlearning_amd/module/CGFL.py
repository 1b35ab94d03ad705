"""SegmentationLossaux / softmax_focalloss (reference: module/CGFL.py:72-101, 192-227).

loss = CE_mean(valid) * [ sum_{all pixels} (1 - p_y)(1 - l1_b/7) / (n_valid + B) ]   — the bracket is detached.
STATUS: evaluated with ATen-ROCm ops on the GPU for now; the one-pass HIP loss kernel replaces it."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..losses.auxloss import MCTransAuxLoss


def softmax_focalloss(y_pred, y_true, ignore_index=-1, gamma=None, normalize=False):
    if normalize:
        raise NotImplementedError("softmax_focalloss: normalize=True is not on the RSSFormer path")
    ce = F.cross_entropy(y_pred.float(), y_true, ignore_index=ignore_index)
    with torch.no_grad():
        p = y_pred.float().softmax(dim=1)
        valid = y_true != ignore_index
        idx = torch.where(valid, y_true, torch.zeros_like(y_true)).unsqueeze(1)
        mf = (1.0 - torch.gather(p, 1, idx).squeeze(1)) * (1.0 - gamma / 7.0)[:, None, None]
        scale = mf.sum() / (valid.sum() + p.size(0))
    return ce * scale


class SegmentationLossaux(nn.Module):
    def __init__(self, loss_config):
        super().__init__()
        self.loss_config = loss_config
        self.criterion_aux = MCTransAuxLoss()
        for k in loss_config:
            if k not in ("ce", "ignore_index"):
                raise NotImplementedError("SegmentationLossaux (HIP build): only the `ce` branch of the reference "
                                          "config (configs/baseline/hrnetw32.py:25-33) is built, got '{}'".format(k))

    def forward(self, y_pred, y_true, y_pred2):
        loss_dict = dict()
        if "ce" in self.loss_config:
            fg = ((y_true > 0) & (y_true != -1)).to(y_pred2.dtype)
            _, l1 = self.criterion_aux(y_pred2.float(), fg)
            loss_dict["fc_loss"] = softmax_focalloss(y_pred, y_true, gamma=l1)
        return loss_dict
