"""LoveDA training transform chain on the GPU (mirror of the reference's configs/base/loveda.py:18-36 + data/loveda.py:82-91).

The reference decodes a tile on a CPU worker, runs albumentations `Compose([RandomCrop(512, 512), OneOf([HorizontalFlip,
VerticalFlip, RandomRotate90], p=0.75), ShiftScaleRotate(p=0.2), Normalize(mean, std, max_pixel_value=1), ToTensor])` and
shifts the mask by -1.  Here the decoded uint8 tiles live in HBM (a 1024x1024x3 tile is 3 MB: the whole LoveDA training
split, 2 522 tiles, is 7.9 GB of the 288) and ONE launch of `rssf_input_pipeline` produces the normalised channels-last batch
and the int64 labels; the host only draws the per-image integers.  ShiftScaleRotate is not implemented.
The random stream is this class's own (a seeded numpy Generator): albumentations' stream cannot be reproduced without the
library, and the reference seeds nothing here either.
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L

LOVEDA_MEAN = (123.675, 116.28, 103.53)       # configs/base/loveda.py:31-33
LOVEDA_STD = (58.395, 57.12, 57.375)
AUG_NONE, AUG_HFLIP, AUG_VFLIP, AUG_ROT90 = 0, 1, 2, 3        # include/rssf.h RSSF_AUG_*


class DeviceAugment:
    """images: uint8 [N,H,W,3], masks: uint8 [N,H,W] (raw LoveDA label ids, 0 = no-data) device tensors."""

    def __init__(self, images, masks=None, crop=512, p_oneof=0.75, mean=LOVEDA_MEAN, std=LOVEDA_STD, max_pixel_value=1.0,
                 dtype=torch.float32, seed=None):
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_contiguous():
            raise ValueError("DeviceAugment: images must be a contiguous uint8 [N,H,W,3] tensor")
        if masks is not None and (masks.dtype != torch.uint8 or tuple(masks.shape) != tuple(images.shape[:3]) or not masks.is_contiguous()):
            raise ValueError("DeviceAugment: masks must be a contiguous uint8 [N,H,W] tensor matching the images")
        if crop > images.shape[1] or crop > images.shape[2]:
            raise ValueError("DeviceAugment: crop larger than the tiles")         # albumentations raises here as well
        L.require_gpu(images)
        self.images, self.masks, self.crop, self.p_oneof = images, masks, int(crop), float(p_oneof)
        self.mean = (ctypes.c_float * 3)(*mean)
        self.std = (ctypes.c_float * 3)(*std)
        self.max_pixel_value, self.dtype = float(max_pixel_value), dtype
        self.rng = np.random.default_rng(seed)

    def draw(self, batch):
        """[batch][4] int32 {source tile, crop y0, crop x0, op}: RandomCrop's uniform offsets; OneOf with p: one of the three
        transforms with equal probability (each is built with p=True, i.e. always applies once chosen); RandomRotate90 draws
        its factor from {0,1,2,3}."""
        n, H, W = self.images.shape[:3]
        out = np.zeros((batch, 4), dtype=np.int32)
        out[:, 0] = self.rng.integers(0, n, batch)
        out[:, 1] = self.rng.integers(0, H - self.crop + 1, batch)
        out[:, 2] = self.rng.integers(0, W - self.crop + 1, batch)
        for b in range(batch):
            if self.rng.random() < self.p_oneof:
                which = int(self.rng.integers(0, 3))
                out[b, 3] = AUG_HFLIP if which == 0 else AUG_VFLIP if which == 1 else AUG_ROT90 + int(self.rng.integers(0, 4))
        return out

    def apply(self, params):
        """params: int32 [B,4] (numpy or tensor).  Returns (img logical [B,3,crop,crop] channels-last of self.dtype,
        labels int64 [B,crop,crop] or None)."""
        p = torch.as_tensor(np.asarray(params, dtype=np.int32)).to(self.images.device).contiguous()
        B, S = p.shape[0], self.crop
        n, H, W = self.images.shape[:3]
        pc = np.asarray(params)
        if (pc[:, 0] < 0).any() or (pc[:, 0] >= n).any() or (pc[:, 1] < 0).any() or (pc[:, 1] + S > H).any() or (pc[:, 2] < 0).any() \
                or (pc[:, 2] + S > W).any() or (pc[:, 3] < 0).any() or (pc[:, 3] > AUG_ROT90 + 3).any():
            raise ValueError("DeviceAugment: parameters out of range")
        img = torch.empty(B, S, S, 3, device=self.images.device, dtype=self.dtype)
        lab = torch.empty(B, S, S, device=self.images.device, dtype=torch.int64) if self.masks is not None else None
        L.check(L.load().rssf_input_pipeline(L.ptr(self.images), L.ptr(self.masks), L.ptr(p), L.ptr(img), L.ptr(lab), B, n, H, W, S, S,
                                             self.mean, self.std, self.max_pixel_value, L.dtype_code(img), L.stream()),
                "rssf_input_pipeline")
        return img.permute(0, 3, 1, 2), lab

    def __call__(self, batch):
        return self.apply(self.draw(batch))
