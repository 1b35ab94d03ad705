"""SegmentationLossaux / softmax_focalloss (reference: module/CGFL.py:72-101, 192-227).

loss = CE_mean(valid) * [ sum_{all pixels} (1 - p_y)(1 - l1_b/7) / (n_valid + B) ]   — the bracket is detached, so the
aux head gets no gradient.  Evaluated by the one-pass HIP loss kernels (csrc/loss.hip) through nnf.cgfl_loss."""
import torch
import torch.nn as nn

from .. import nnf
from ..losses.auxloss import MCTransAuxLoss


def softmax_focalloss(y_pred, y_true, ignore_index=-1, gamma=None, normalize=False, aux=None):
    """The reference's call shape (CGFL.py:72, 221: `softmax_focalloss(y_pred, y_true, gamma=l1)`): `gamma` is the per-sample l1
    vector [B] that MCTransAuxLoss returns.  SegmentationLossaux passes the aux scores instead (`aux=`) and the fused kernel derives
    l1 from them in its finalize launch."""
    if normalize:
        raise NotImplementedError("softmax_focalloss: normalize=True is not on the RSSFormer path")
    if aux is not None:
        return nnf.cgfl_loss(y_pred, y_true, aux, ignore_index)
    if not torch.is_tensor(gamma):
        raise TypeError("softmax_focalloss: gamma must be the per-sample tensor [B] (the reference unsqueezes it: CGFL.py:84), or pass aux=")
    return nnf.cgfl_loss(y_pred, y_true, gamma.reshape(-1, 1), ignore_index)


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
            loss_dict["fc_loss"] = nnf.cgfl_loss(y_pred, y_true, y_pred2, self.loss_config.get("ignore_index", -1))
        return loss_dict
