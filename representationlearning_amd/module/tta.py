"""Test-time augmentation (reference: module/tta.py:12-46 `tta` / `TestTimeAugmentation`, :52-136 the transforms).

`er.MultiTransform` of the un-vendored `ever` package is restated from its call sites in the reference's `tta()`:
`transform(image)` yields one image per transform, `inv_transform(outs)` maps every output back with ITS transform, the
results are averaged.  `Scale` (bilinear, align_corners=True, both ways - the transform eval.py:57-64 uses) runs on the
librssf upsampling kernel; the flips / rot90 / transpose are index permutations (torch views + one copy)."""
import torch
import torch.nn as nn

from .. import nnf

__all__ = ["tta", "TestTimeAugmentation", "MultiTransform", "Identity", "Rotate90k", "HorizontalFlip", "VerticalFlip", "Transpose",
           "Scale"]


class Transform:
    def transform(self, inputs):
        raise NotImplementedError

    def inv_transform(self, transformed_inputs):
        raise NotImplementedError


class MultiTransform:
    def __init__(self, *transforms):
        self.transforms = list(transforms)

    def transform(self, inputs):
        return [t.transform(inputs) for t in self.transforms]

    def inv_transform(self, outs):
        return [t.inv_transform(o) for t, o in zip(self.transforms, outs)]


def tta(model, image, tta_config):
    trans = MultiTransform(*tta_config)
    images = trans.transform(image)
    with torch.no_grad():
        outs = [model(im) for im in images]
    outs = trans.inv_transform(outs)
    return sum(outs) / len(outs)


class TestTimeAugmentation(nn.Module):
    def __init__(self, module, tta_config):
        super().__init__()
        self.module = module
        self.trans = MultiTransform(*tta_config)

    @torch.no_grad()
    def forward(self, image):
        outs = self.trans.inv_transform([self.module(im) for im in self.trans.transform(image)])
        return sum(outs) / len(outs)


class Identity(Transform):
    def transform(self, inputs):
        return inputs

    def inv_transform(self, transformed_inputs):
        return transformed_inputs


class Rotate90k(Transform):
    def __init__(self, k=1):
        assert k in [1, 2, 3]
        self.k = k

    def transform(self, inputs):
        return torch.rot90(inputs, self.k, [2, 3])

    def inv_transform(self, transformed_inputs):
        return torch.rot90(transformed_inputs, 4 - self.k, [2, 3])


class HorizontalFlip(Transform):
    def transform(self, inputs):
        return torch.flip(inputs, [3])

    def inv_transform(self, transformed_inputs):
        return torch.flip(transformed_inputs, [3])


class VerticalFlip(Transform):
    def transform(self, inputs):
        return torch.flip(inputs, [2])

    def inv_transform(self, transformed_inputs):
        return torch.flip(transformed_inputs, [2])


class Transpose(Transform):
    def transform(self, inputs):
        return torch.transpose(inputs, 2, 3)

    def inv_transform(self, transformed_inputs):
        return torch.transpose(transformed_inputs, 2, 3)


class Scale(Transform):
    def __init__(self, size=None, scale_factor=None):
        self.size = size
        self.scale_factor = scale_factor
        self.input_shape = None

    def _target(self, h, w):
        if self.size is not None:
            return tuple(self.size) if isinstance(self.size, (tuple, list)) else (int(self.size), int(self.size))
        sf = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor, self.scale_factor)
        return int(h * float(sf[0])), int(w * float(sf[1]))          # F.interpolate: floor(input * scale_factor)

    def transform(self, inputs):
        self.input_shape = inputs.shape
        return nnf.upsample_bilinear(inputs, self._target(inputs.shape[2], inputs.shape[3]))

    def inv_transform(self, transformed_inputs):
        return nnf.upsample_bilinear(transformed_inputs, (self.input_shape[2], self.input_shape[3]))
