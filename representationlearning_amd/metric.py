"""Confusion-matrix pixel metrics (stands in for the un-vendored `er.metric.PixelMetric`, eval.py:47 of the
reference; mIoU semantics follow the in-tree SCD-AAAI2023/utils/evaluate.py:9-35: bincount confusion matrix,
nanmean over classes that occur).  "parity unpinned": the reference ships no fixture for it."""
import torch


class PixelMetric:
    def __init__(self, num_classes, logdir=None, logger=None):
        self.num_classes = num_classes
        self.cm = torch.zeros(num_classes, num_classes, dtype=torch.int64)

    def forward(self, y_true, y_pred):
        y_true = y_true.reshape(-1).to(torch.int64).cpu()
        y_pred = y_pred.reshape(-1).to(torch.int64).cpu()
        k = self.num_classes
        self.cm += torch.bincount(y_true * k + y_pred, minlength=k * k).reshape(k, k)

    def forward_scores(self, y_true, scores, ignore_index=-1):
        """Device path: class scores [B,K,H,W] (logits or probabilities) + labels [B,H,W] -> confusion matrix, in ONE
        kernel (argmax, ignore mask, histogram: rssf_argmax_confusion); nothing but K*K counters leaves the GPU."""
        from . import ops
        if not hasattr(self, "_cm_dev") or self._cm_dev.device != scores.device:
            self._cm_dev = torch.zeros(self.num_classes, self.num_classes, dtype=torch.int64, device=scores.device)
        ops.argmax_confusion(scores, y_true, self._cm_dev, ignore_index, want_pred=False)

    def _sync(self):
        if hasattr(self, "_cm_dev"):
            self.cm += self._cm_dev.cpu()
            self._cm_dev.zero_()

    def iou(self):
        self._sync()
        cm = self.cm.double()
        tp = cm.diag()
        denom = cm.sum(0) + cm.sum(1) - tp
        return (tp / denom).tolist()

    def miou(self):
        v = torch.tensor(self.iou())
        return float(v[~torch.isnan(v)].mean())

    def summary_all(self):
        iou = self.iou()
        acc = float(self.cm.diag().sum() / max(1, int(self.cm.sum())))
        return dict(iou=iou, miou=self.miou(), overall_accuracy=acc)
