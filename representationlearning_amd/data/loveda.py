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


class SyntheticTiles:
    """Stand-in for a LoveDA folder when no dataset is on disk (there is none in this build's environment): `n` procedurally drawn
    uint8 tiles with blocky raw label ids 0..classes (0 = no-data, like LoveDA's masks), same item format as `LoveDA`.  The tiles go
    through exactly the path real tiles take: upload as uint8, HBM-resident, augmented by `rssf_input_pipeline`."""

    def __init__(self, n=32, size=1024, classes=6, seed=2333, block=16):
        self.n, self.size, self.classes, self.seed, self.block = int(n), int(size), int(classes), int(seed), int(block)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        rng = np.random.default_rng(self.seed * 1000003 + idx)
        S, b = self.size, self.block
        image = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
        m = rng.integers(0, self.classes + 1, ((S + b - 1) // b, (S + b - 1) // b), dtype=np.uint8)
        mask = np.ascontiguousarray(np.repeat(np.repeat(m, b, 0), b, 1)[:S, :S])
        return image, dict(cls=mask.astype(np.int64) - 1, raw_mask=mask, fname="synthetic%05d.png" % idx)


def epoch_shard(n_tiles, epoch, rank, world, batch_size, seed=2333):
    """Tile order of one rank for one epoch: ONE permutation per epoch (same seed on every rank), rank r takes every world-th tile,
    truncated to whole batches (drop_last) - the ranks' shares are disjoint and of equal length."""
    perm = np.random.default_rng(seed + epoch).permutation(n_tiles)
    return perm[rank::world][:(n_tiles // world // batch_size) * batch_size]


class LoveDALoader:
    """Training loader (reference `LoveDALoader` with `training=True`, data/loveda.py:97-117: StepDistributedSampler - every rank walks
    its own 1/world share of one permutation per epoch - batch_size per rank, drop_last; transforms configs/base/loveda.py:18-36).

    MI355X-first: decoded tiles are HBM-RESIDENT uint8 (a 1024 x 1024 x 3 tile is 3 MB; the whole LoveDA training split is 7.9 GB
    of the 288), uploaded the first time the sampler asks for them (a thread pool decodes the files of a batch) and never again; the
    whole transform chain of a batch is ONE `rssf_input_pipeline` launch (`DeviceAugment`).  The reference decodes and augments every
    tile on 2 CPU workers per GPU in every epoch.  Iterating yields (img [B,3,crop,crop] channels-last, dict(cls=int64 labels)) for
    one epoch; `epoch` advances by itself, so `for ep in range(E): for img, tgt in loader:` reshuffles like the reference."""

    def __init__(self, dataset, batch_size=8, rank=0, world=1, crop=512, p_oneof=0.75, shift_scale_rotate=None, dtype=torch.float32,
                 device="cuda", seed=2333, mean=LOVEDA_MEAN, std=LOVEDA_STD, max_pixel_value=1.0, decode_threads=8):
        if len(dataset) < batch_size * world:
            raise ValueError("LoveDALoader: %d tiles cannot fill one batch of %d on each of %d ranks" % (len(dataset), batch_size, world))
        self.ds, self.bs, self.rank, self.world, self.seed = dataset, int(batch_size), int(rank), int(world), int(seed)
        self.device, self.epoch, self.threads = device, 0, int(decode_threads)
        img0, t0 = dataset[0]
        H, W = img0.shape[:2]
        self.store = torch.empty(len(dataset), H, W, 3, device=device, dtype=torch.uint8)
        self.mstore = torch.empty(len(dataset), H, W, device=device, dtype=torch.uint8)
        self.have = np.zeros(len(dataset), dtype=bool)
        self._put(0, img0, t0)
        self.aug = DeviceAugment(self.store, self.mstore, crop=crop, p_oneof=p_oneof, mean=mean, std=std, max_pixel_value=max_pixel_value,
                                 dtype=dtype, seed=seed + 7919 * rank, shift_scale_rotate=shift_scale_rotate)

    def _put(self, i, image, tgt):
        if tuple(image.shape) != tuple(self.store.shape[1:]) or tgt["raw_mask"] is None:
            raise ValueError("LoveDALoader: tile %d is %s (labelled: %s); the resident store holds labelled %s tiles"
                             % (i, tuple(image.shape), tgt["raw_mask"] is not None, tuple(self.store.shape[1:])))
        self.store[i].copy_(torch.from_numpy(np.ascontiguousarray(image)))
        self.mstore[i].copy_(torch.from_numpy(np.ascontiguousarray(tgt["raw_mask"])))
        self.have[i] = True

    def __len__(self):
        return len(self.ds) // self.world // self.bs

    def indices(self, epoch):
        """This rank's tile order for one epoch: ranks share ONE permutation (seeded by the epoch) and take every world-th tile."""
        return epoch_shard(len(self.ds), epoch, self.rank, self.world, self.bs, self.seed)

    def resident_fraction(self):
        return float(self.have.mean())

    def __iter__(self):
        idx = self.indices(self.epoch)
        self.epoch += 1
        for b in range(len(self)):
            tiles = idx[b * self.bs:(b + 1) * self.bs]
            missing = [int(i) for i in tiles if not self.have[i]]
            if missing:
                if self.threads > 1 and len(missing) > 1:
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(min(self.threads, len(missing))) as ex:
                        items = list(ex.map(self.ds.__getitem__, missing))
                else:
                    items = [self.ds[i] for i in missing]
                for i, (image, tgt) in zip(missing, items):
                    self._put(i, image, tgt)
            params = self.aug.draw(self.bs)
            params[:, 0] = tiles                        # the sampler, not RandomCrop's stream, picks the tiles
            img, lab = self.aug.apply(params, self.aug.draw_affine(self.bs))
            yield img, dict(cls=lab)
