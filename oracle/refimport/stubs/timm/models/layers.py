import collections.abc
from itertools import repeat

from torch.nn.init import trunc_normal_  # noqa: F401


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return tuple(repeat(x, 2))


import torch.nn as _nn


class DropPath(_nn.Module):
    """timm.models.layers.DropPath restated for the import of net/wavecam.py:8 only: stochastic depth, identity at rate 0 / eval."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep
