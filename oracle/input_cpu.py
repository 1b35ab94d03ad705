"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

CPU restatement of the reference's LoveDA training transform chain for the ops the device pipeline covers
(configs/base/loveda.py:18-36, data/loveda.py:82-91).  **PARITY UNPINNED**: the chain is albumentations code, an un-vendored,
unpinned third-party dependency that is not installed in this image (no import of the reference's loader is possible, and the
reference holds no fixtures for it).  What follows restates albumentations' published algorithms:
  RandomCrop.apply            img[y1:y1+h, x1:x1+w]                                   (functional.random_crop)
  HorizontalFlip / VerticalFlip   img[:, ::-1] / img[::-1]                           (functional.hflip / vflip)
  RandomRotate90.apply        np.rot90(img, factor), factor in {0,1,2,3}             (functional.rot90)
  Normalize.apply             mean, std float32 * max_pixel_value; img = float32(img); img -= mean; img *= reciprocal(std)
  data/loveda.py:84           mask = imread(mask).astype(long) - 1
ShiftScaleRotate (cv2.warpAffine) is not restated.
"""
import numpy as np

AUG_NONE, AUG_HFLIP, AUG_VFLIP, AUG_ROT90 = 0, 1, 2, 3


def geometric(a, op):
    if op == AUG_HFLIP:
        return a[:, ::-1]
    if op == AUG_VFLIP:
        return a[::-1]
    if op >= AUG_ROT90:
        return np.rot90(a, op - AUG_ROT90)
    return a


def normalize(img_u8, mean, std, max_pixel_value=1.0):
    mean = np.array(mean, dtype=np.float32) * np.float32(max_pixel_value)
    std = np.array(std, dtype=np.float32) * np.float32(max_pixel_value)
    den = np.reciprocal(std, dtype=np.float32)
    out = img_u8.astype(np.float32)
    out -= mean
    out *= den
    return out


def pipeline(images, masks, params, crop, mean, std, max_pixel_value=1.0):
    """images uint8 [N,H,W,3], masks uint8 [N,H,W] or None, params int [B,4] -> (float32 [B,crop,crop,3], int64 [B,crop,crop])."""
    imgs, labs = [], []
    for src, y0, x0, op in np.asarray(params):
        im = geometric(images[src, y0:y0 + crop, x0:x0 + crop], op)
        imgs.append(normalize(np.ascontiguousarray(im), mean, std, max_pixel_value))
        if masks is not None:
            labs.append(np.ascontiguousarray(geometric(masks[src, y0:y0 + crop, x0:x0 + crop], op)).astype(np.int64) - 1)
    return np.stack(imgs), (np.stack(labs) if masks is not None else None)
