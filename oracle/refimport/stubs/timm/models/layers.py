import collections.abc
from itertools import repeat

from torch.nn.init import trunc_normal_  # noqa: F401


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return tuple(repeat(x, 2))
