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
  ShiftScaleRotate.apply      cv2.warpAffine(img, M, (w, h), INTER_LINEAR, BORDER_REFLECT_101), mask: INTER_NEAREST - restated
                              from OpenCV's published imgwarp.cpp (WarpAffineInvoker, remapBilinear / remapNearest for 8-bit
                              data); cv2 is not installed either: PARITY UNPINNED as well.
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


def _reflect101(p, n):
    """cv::borderInterpolate(p, n, BORDER_REFLECT_101), vectorised."""
    if n == 1:
        return np.zeros_like(p)
    period = 2 * (n - 1)
    q = np.mod(p, period)
    return np.where(q >= n, period - q, q)


def warp_affine_u8(img, inv_m, nearest=False):
    """cv2.warpAffine on an 8-bit image [H,W(,C)] with the INVERSE matrix `inv_m` (6 float64), dsize = (W, H), BORDER_REFLECT_101.
    Fixed point as OpenCV: AB_BITS = 10, INTER_BITS = 5; cvRound = round half to even (np.rint)."""
    H, W = img.shape[:2]
    m = np.asarray(inv_m, dtype=np.float64)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    adelta = np.rint(m[0] * x * 1024.0).astype(np.int64)
    bdelta = np.rint(m[3] * x * 1024.0).astype(np.int64)
    X0 = np.rint((m[1] * y + m[2]) * 1024.0).astype(np.int64)
    Y0 = np.rint((m[4] * y + m[5]) * 1024.0).astype(np.int64)
    if nearest:
        sx = _reflect101((X0[:, None] + 512 + adelta[None, :]) >> 10, W)
        sy = _reflect101((Y0[:, None] + 512 + bdelta[None, :]) >> 10, H)
        return img[sy, sx]
    X = (X0[:, None] + 16 + adelta[None, :]) >> 5
    Y = (Y0[:, None] + 16 + bdelta[None, :]) >> 5
    sx, sy, ax, ay = X >> 5, Y >> 5, X & 31, Y & 31
    acc = np.full(img.shape[:2] + img.shape[2:], 1 << 14, dtype=np.int64)
    for t, w in enumerate(((32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32)):
        ty, tx = _reflect101(sy + (t >> 1), H), _reflect101(sx + (t & 1), W)
        tap = img[ty, tx].astype(np.int64)
        acc = acc + tap * (w[..., None] if img.ndim == 3 else w)
    return (acc >> 15).astype(np.uint8)


def pipeline(images, masks, params, crop, mean, std, max_pixel_value=1.0, affine=None):
    """images uint8 [N,H,W,3], masks uint8 [N,H,W] or None, params int [B,4], affine float64 [B,6] (NaN row: no warp) or None
    -> (float32 [B,crop,crop,3], int64 [B,crop,crop])."""
    imgs, labs = [], []
    for b, (src, y0, x0, op) in enumerate(np.asarray(params)):
        warp = affine is not None and np.isfinite(affine[b][0])
        im = np.ascontiguousarray(geometric(images[src, y0:y0 + crop, x0:x0 + crop], op))
        if warp:
            im = warp_affine_u8(im, affine[b])
        imgs.append(normalize(im, mean, std, max_pixel_value))
        if masks is not None:
            mk = np.ascontiguousarray(geometric(masks[src, y0:y0 + crop, x0:x0 + crop], op))
            if warp:
                mk = warp_affine_u8(mk, affine[b], nearest=True)
            labs.append(mk.astype(np.int64) - 1)
    return np.stack(imgs), (np.stack(labs) if masks is not None else None)
