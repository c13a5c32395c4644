"""CPU restatement of the image side of the reference's data pipeline.  TEST INFRASTRUCTURE ONLY.

The reference resizes with torchvision's `F.resize` on a PIL image (maskrcnn_benchmark/data/transforms/transforms.py:
58-62), i.e. Pillow's `Image.resize(size, BILINEAR)`: a separable triangle filter whose support grows with the
down-scaling factor, evaluated in 8-bit fixed point (Pillow src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; not vendored in /root/reference — third
party, Pillow 12.2 in this image).  `pil_bilinear_resize` restates that integer arithmetic; tests pin it against
Pillow itself where Pillow is importable.  Then ToTensor -> [2,1,0]*255 -> Normalize(mean, std)
(transforms.py:77-97, `to_bgr255`)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """-> (bounds int32 [out,2] = (first input index, tap count), coefficients int32 [out, ksize] in 2^-22 units)"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale            # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        n = xmax - xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(n):
            w = (x + xmin - center + 0.5) * ss
            w = -w if w < 0 else w
            w = 1.0 - w if w < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:n] /= ww
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            coeffs[xx, x] = int(v - 0.5) if k[x] < 0 else int(v + 0.5)
        bounds[xx] = (xmin, n)
    return bounds, coeffs


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_bilinear_resize(img, out_h, out_w):
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C]; horizontal pass first, then vertical, like ImagingResample"""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        bounds, coeffs = resample_coeffs(W, out_w)
        out = np.empty((H, out_w, C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = (src[:, x0:x0 + n, :] * coeffs[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx, :] = _clip8(acc)
        cur = out
    if out_h != H:
        bounds, coeffs = resample_coeffs(H, out_h)
        out = np.empty((out_h, cur.shape[1], C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = (src[y0:y0 + n, :, :] * coeffs[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur


def preprocess(img_rgb_u8, out_h, out_w, flip, pixel_mean, pixel_std, to_bgr255=True):
    """Resize -> RandomHorizontalFlip (decision given) -> ToTensor -> Normalize of transforms.py:32-97.
    -> float32 [3, out_h, out_w]"""
    r = pil_bilinear_resize(img_rgb_u8, out_h, out_w)
    if flip:
        r = r[:, ::-1, :]
    t = r.astype(np.float32) / np.float32(255.0)            # ToTensor
    t = np.transpose(t, (2, 0, 1))
    if to_bgr255:
        t = t[[2, 1, 0]] * np.float32(255.0)
    mean = np.asarray(pixel_mean, dtype=np.float32).reshape(3, 1, 1)
    std = np.asarray(pixel_std, dtype=np.float32).reshape(3, 1, 1)
    return ((t - mean) / std).astype(np.float32)
