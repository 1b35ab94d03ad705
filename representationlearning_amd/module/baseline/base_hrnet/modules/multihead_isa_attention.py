"""Window geometry helpers.  In the reference (modules/multihead_isa_attention.py:364-426) PadBlock /
LocalPermuteModule materialise padded and permuted copies of the feature map; here the fused HIP kernel does
that index math in-kernel, so on the hot path these classes only describe the geometry.  The reference's four methods are kept for
callers that use the classes on their own: pure layout (pad / slice / reshape / permute), any device, no arithmetic."""
import math

import torch.nn.functional as F


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

    def pad_if_needed(self, x, size):
        """x [n, h, w, c] -> zero-padded to multiples of the group size, floor(P/2) before (reference :373-382)."""
        _, h, w, _ = size
        t, b, l, r = self.pads(h, w)
        return F.pad(x, (0, 0, l, r, t, b)) if (t or b or l or r) else x

    def depad_if_needed(self, x, size):
        """inverse of pad_if_needed: the original h x w region (reference :384-390)."""
        _, h, w, _ = size
        t, b, l, r = self.pads(h, w)
        return x[:, t:t + h, l:l + w, :] if (t or b or l or r) else x


class LocalPermuteModule(object):
    """(n, qh*ph, qw*pw, c) <-> (ph*pw, n*qh*qw, c) grouping; window id and slot of a padded pixel."""

    def __init__(self, local_group_size=7):
        self.lgs = local_group_size if isinstance(local_group_size, (tuple, list)) else (local_group_size,) * 2

    def permute(self, x, size):
        """[n, qh*ph, qw*pw, c] -> [ph*pw, n*qh*qw, c] (slot-major, window id = (n, qh, qw) row-major; reference :402-413).
        `size` = (n, h, w, c) of the PADDED map."""
        n, h, w, c = size
        ph, pw = self.lgs
        qh, qw = h // ph, w // pw
        return x.reshape(n, qh, ph, qw, pw, c).permute(2, 4, 0, 1, 3, 5).reshape(ph * pw, n * qh * qw, c)

    def rev_permute(self, x, size):
        """inverse of permute (reference :415-426)."""
        n, h, w, c = size
        ph, pw = self.lgs
        qh, qw = h // ph, w // pw
        return x.reshape(ph, pw, n, qh, qw, c).permute(2, 3, 0, 4, 1, 5).reshape(n, h, w, c)

    def window_of(self, n, u, v, hp, wp):
        qh, qw = hp // self.lgs[0], wp // self.lgs[1]
        return (n * qh + u // self.lgs[0]) * qw + v // self.lgs[1], (u % self.lgs[0]) * self.lgs[1] + v % self.lgs[1]
