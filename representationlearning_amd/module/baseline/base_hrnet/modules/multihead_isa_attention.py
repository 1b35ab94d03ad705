"""Window geometry helpers.  In the reference (modules/multihead_isa_attention.py:364-426) PadBlock /
LocalPermuteModule materialise padded and permuted copies of the feature map; here the fused HIP kernel does
that index math in-kernel, so these classes only describe the geometry (and are used by tests)."""
import math


class PadBlock(object):
    """Center zero-pad H, W up to a multiple of the local group size (floor(P/2) before)."""

    def __init__(self, local_group_size=7):
        self.lgs = local_group_size if isinstance(local_group_size, (tuple, list)) else (local_group_size,) * 2
        assert len(self.lgs) == 2

    def pads(self, h, w):
        ph = math.ceil(h / self.lgs[0]) * self.lgs[0] - h
        pw = math.ceil(w / self.lgs[1]) * self.lgs[1] - w
        return ph // 2, ph - ph // 2, pw // 2, pw - pw // 2

    def padded_size(self, h, w):
        t, b, l, r = self.pads(h, w)
        return h + t + b, w + l + r


class LocalPermuteModule(object):
    """(n, qh*ph, qw*pw, c) <-> (ph*pw, n*qh*qw, c) grouping; window id and slot of a padded pixel."""

    def __init__(self, local_group_size=7):
        self.lgs = local_group_size if isinstance(local_group_size, (tuple, list)) else (local_group_size,) * 2

    def window_of(self, n, u, v, hp, wp):
        qh, qw = hp // self.lgs[0], wp // self.lgs[1]
        return (n * qh + u // self.lgs[0]) * qw + v // self.lgs[1], (u % self.lgs[0]) * self.lgs[1] + v % self.lgs[1]
