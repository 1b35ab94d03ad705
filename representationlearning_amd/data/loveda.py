"""LoveDA training transform chain on the GPU (mirror of the reference's configs/base/loveda.py:18-36 + data/loveda.py:82-91).

The reference decodes a tile on a CPU worker, runs albumentations `Compose([RandomCrop(512, 512), OneOf([HorizontalFlip,
VerticalFlip, RandomRotate90], p=0.75), ShiftScaleRotate(p=0.2), Normalize(mean, std, max_pixel_value=1), ToTensor])` and
shifts the mask by -1.  Here the decoded uint8 tiles live in HBM (a 1024x1024x3 tile is 3 MB: the whole LoveDA training
split, 2 522 tiles, is 7.9 GB of the 288) and ONE launch of `rssf_input_pipeline` produces the normalised channels-last batch
and the int64 labels; the host only draws the per-image numbers (crop offsets, the OneOf choice, the affine matrix of
ShiftScaleRotate).
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


def ssr_inverse_matrix(width, height, angle, scale, dx, dy):
    """The matrix cv2.warpAffine actually uses for albumentations' shift_scale_rotate: M = getRotationMatrix2D((w/2, h/2), angle,
    scale) (alpha = scale*cos, beta = scale*sin, angle in degrees counter-clockwise), translation += (dx*w, dy*h), then
    cv::invertAffineTransform - all in float64, operation for operation."""
    import math
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = width / 2, height / 2
    m = [alpha, beta, (1 - alpha) * cx - beta * cy + dx * width, -beta, alpha, beta * cx + (1 - alpha) * cy + dy * height]
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    i0, i1, i3, i4 = a11, m[1] * (-d), m[3] * (-d), a22
    b1 = -i0 * m[2] - i1 * m[5]
    b2 = -i3 * m[2] - i4 * m[5]
    return np.array([i0, i1, b1, i3, i4, b2], dtype=np.float64)


class DeviceAugment:
    """images: uint8 [N,H,W,3], masks: uint8 [N,H,W] (raw LoveDA label ids, 0 = no-data) device tensors."""

    def __init__(self, images, masks=None, crop=512, p_oneof=0.75, mean=LOVEDA_MEAN, std=LOVEDA_STD, max_pixel_value=1.0,
                 dtype=torch.float32, seed=None, shift_scale_rotate=None):
        # shift_scale_rotate: dict(shift_limit, scale_limit, rotate_limit, p) as configs/base/loveda.py:31 passes to albumentations
        # (ShiftScaleRotate(shift_limit=0.0625, scale_limit=0.2, rotate_limit=45, p=0.2)), or None
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
        self.ssr = dict(shift_scale_rotate) if shift_scale_rotate else None

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

    def draw_affine(self, batch):
        """[batch][6] float64: per image the INVERSE warp matrix of ShiftScaleRotate, NaN row = not applied (probability 1 - p).
        albumentations draws angle ~ U(-rotate_limit, rotate_limit), scale ~ U(1 - scale_limit, 1 + scale_limit), dx, dy ~
        U(-shift_limit, shift_limit); the matrix is cv2.getRotationMatrix2D((w/2, h/2), angle, scale) with (dx*w, dy*h) added to the
        translation, inverted as cv2.warpAffine inverts it (see `ssr_inverse_matrix`)."""
        out = np.full((batch, 6), np.nan, dtype=np.float64)
        if self.ssr is None:
            return out
        sl, cl, rl, p = (self.ssr[k] for k in ("shift_limit", "scale_limit", "rotate_limit", "p"))
        for b in range(batch):
            if self.rng.random() < p:
                angle, scale = self.rng.uniform(-rl, rl), self.rng.uniform(1.0 - cl, 1.0 + cl)
                dx, dy = self.rng.uniform(-sl, sl), self.rng.uniform(-sl, sl)
                out[b] = ssr_inverse_matrix(self.crop, self.crop, angle, scale, dx, dy)
        return out

    def apply(self, params, affine=None):
        """params: int32 [B,4] (numpy or tensor); affine: float64 [B,6] inverse warp matrices (NaN row = none) or None.
        Returns (img logical [B,3,crop,crop] channels-last of self.dtype, labels int64 [B,crop,crop] or None)."""
        p = torch.as_tensor(np.asarray(params, dtype=np.int32)).to(self.images.device).contiguous()
        B, S = p.shape[0], self.crop
        n, H, W = self.images.shape[:3]
        pc = np.asarray(params)
        if (pc[:, 0] < 0).any() or (pc[:, 0] >= n).any() or (pc[:, 1] < 0).any() or (pc[:, 1] + S > H).any() or (pc[:, 2] < 0).any() \
                or (pc[:, 2] + S > W).any() or (pc[:, 3] < 0).any() or (pc[:, 3] > AUG_ROT90 + 3).any():
            raise ValueError("DeviceAugment: parameters out of range")
        img = torch.empty(B, S, S, 3, device=self.images.device, dtype=self.dtype)
        lab = torch.empty(B, S, S, device=self.images.device, dtype=torch.int64) if self.masks is not None else None
        aff = None
        if affine is not None:
            a = np.ascontiguousarray(np.asarray(affine, dtype=np.float64))
            if a.shape != (B, 6):
                raise ValueError("DeviceAugment: affine must be [B,6]")
            if np.isfinite(a[:, 0]).any():
                aff = torch.from_numpy(a).to(self.images.device)
        L.check(L.load().rssf_input_pipeline(L.ptr(self.images), L.ptr(self.masks), L.ptr(p), L.ptr(aff), L.ptr(img), L.ptr(lab), B, n, H, W, S, S,
                                             self.mean, self.std, self.max_pixel_value, L.dtype_code(img), L.stream()),
                "rssf_input_pipeline")
        return img.permute(0, 3, 1, 2), lab

    def __call__(self, batch):
        return self.apply(self.draw(batch), self.draw_affine(batch))


# ---- file side (reference data/loveda.py:53-91, 94-121): the folders of LoveDA tiles and the loader that feeds the model ----
class LoveDA:
    """`LoveDA(image_dir, mask_dir)` of the reference: *.tif / *.png tiles of one or several folders, masks with the same file
    names (optional).  `__getitem__` returns the DECODED tile - uint8 [H,W,3] - and `dict(cls=mask - 1 as int64 | None, fname)`;
    the transforms run on the GPU (DeviceAugment / DeviceLoader), not per item on a CPU worker.  Files are read with PIL (the
    reference uses skimage.io.imread; same arrays for 8-bit RGB / single-channel PNG and TIFF)."""

    def __init__(self, image_dir, mask_dir=None):
        import glob
        import os
        self.rgb_filepath_list, self.cls_filepath_list = [], []
        if isinstance(image_dir, (list, tuple)):
            mask_dirs = mask_dir if isinstance(mask_dir, (list, tuple)) else [mask_dir] * len(image_dir)
            pairs = list(zip(image_dir, mask_dirs))
        else:
            pairs = [(image_dir, mask_dir)]
        for idir, mdir in pairs:
            files = glob.glob(os.path.join(idir, "*.tif")) + glob.glob(os.path.join(idir, "*.png"))       # data/loveda.py:73-74
            self.rgb_filepath_list += files
            if mdir is not None:
                self.cls_filepath_list += [os.path.join(mdir, os.path.split(f)[-1]) for f in files]

    def __len__(self):
        return len(self.rgb_filepath_list)

    def __getitem__(self, idx):
        import os
        from PIL import Image
        image = np.asarray(Image.open(self.rgb_filepath_list[idx]).convert("RGB"), dtype=np.uint8)
        mask = None
        if self.cls_filepath_list:
            mask = np.asarray(Image.open(self.cls_filepath_list[idx]), dtype=np.uint8)                     # raw ids, 0 = no-data
        return image, dict(cls=None if mask is None else mask.astype(np.int64) - 1, raw_mask=mask,
                           fname=os.path.basename(self.rgb_filepath_list[idx]))


class DeviceLoader:
    """Evaluation / prediction loader (reference `LoveDALoader` with `training=False`: SequentialSampler, batch_size 4,
    transforms = Normalize + ToTensor, data/loveda.py:94-121 + configs/base/loveda.py:47-66).  Decoded tiles are uploaded once per
    batch as uint8 and normalised by `rssf_input_pipeline` (whole tile, no augmentation); yields (img [B,3,H,W] channels-last,
    dict(cls=int64 labels | None, fname=[...]))."""

    def __init__(self, dataset, batch_size=4, device="cuda", dtype=torch.float32, drop_last=False):
        self.ds, self.bs, self.device, self.dtype, self.drop_last = dataset, int(batch_size), device, dtype, drop_last

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        for start in range(0, len(self.ds), self.bs):
            items = [self.ds[i] for i in range(start, min(start + self.bs, len(self.ds)))]
            if self.drop_last and len(items) < self.bs:
                return
            imgs = torch.from_numpy(np.stack([it[0] for it in items])).to(self.device)
            raw = [it[1]["raw_mask"] for it in items]
            masks = None if raw[0] is None else torch.from_numpy(np.stack(raw)).to(self.device)
            aug = DeviceAugment(imgs.contiguous(), None if masks is None else masks.contiguous(), crop=imgs.shape[1], p_oneof=0.0,
                                dtype=self.dtype) if imgs.shape[1] == imgs.shape[2] else None
            if aug is None:
                raise ValueError("DeviceLoader: square tiles expected (LoveDA tiles are 1024 x 1024)")
            params = np.zeros((len(items), 4), dtype=np.int32)
            params[:, 0] = np.arange(len(items))
            img, lab = aug.apply(params)
            yield img, dict(cls=lab, fname=[it[1]["fname"] for it in items])
