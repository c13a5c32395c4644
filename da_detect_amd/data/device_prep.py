"""Batch preparation on the GPU: decoded uint8 RGB images in, the model's padded fp32 batch out.

One pair of kernel launches per image (csrc/image.hip) replaces the reference's per-sample host chain Resize ->
RandomHorizontalFlip -> ToTensor -> Normalize (data/transforms/transforms.py:32-97) and the collator's zero padding
(data/collate_batch.py:46-55, structures/image_list.py:49-91).  The resize is Pillow's bilinear resample reproduced bit
for bit (integer arithmetic), so the batch equals what the host transforms produce; ground-truth boxes are resized /
mirrored by the same BoxList methods the reference calls.  Random decisions (training scale, flip) are drawn from
Python's `random` in the reference's per-sample order: get_size, then the flip toss."""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from ..structures.image_list import ImageList
from .transforms import RandomHorizontalFlip, Resize, transform_params

_PRECISION_BITS = 32 - 8 - 2
_TABLES = {}


def resample_tables(in_size, out_size):
    """Pillow's per-axis tables (Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear filter): bounds
    int32 [out,2] = (first input index, tap count), coefficients int32 [out,ksize] in 2^-22 units."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C truncation of a value > -1
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(w < 1.0, 1.0 - w, 0.0)
    w = np.where(taps < n[:, None], w, 0.0)
    # Pillow sums the weights left to right in double precision; cumsum reproduces that order
    ww = np.cumsum(w, axis=1)[:, -1:]
    k = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    coeffs = (k * float(1 << _PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)   # weights are >= 0
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return bounds, coeffs


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _TABLES:
        b, c = resample_tables(in_size, out_size)
        if len(_TABLES) > 64:
            _TABLES.clear()
        _TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), c.shape[1])
    return _TABLES[key]


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def resize_normalize_into(image_u8, out_hw, flip, mean, std, to_bgr255, out_slot):
    """image_u8 [H,W,3] uint8 on the device -> writes fp32 into out_slot, a [3, Hp, Wp] channels_last view (physically
    [Hp][Wp][3]) with Hp >= oh, Wp >= ow; everything outside [oh, ow] is left untouched (zero padding)"""
    if not image_u8.is_cuda:
        raise _lib.DadetError("resize_normalize_into: the image must live on the HIP device")
    assert image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.shape[2] == 3
    image_u8 = image_u8.contiguous()
    H, W = int(image_u8.shape[0]), int(image_u8.shape[1])
    oh, ow = out_hw
    dev = image_u8.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cur = image_u8
    if ow != W:
        bw, cw, kw = _device_tables(W, ow, dev)
        tmp = torch.empty((H, ow, 3), dtype=torch.uint8, device=dev)
        _lib.call("dadet_image_resample_h", _p(cur), H, W, _p(bw), _p(cw), kw, ow, _p(tmp), stream)
        cur = tmp
    bh = ch = None
    kh = 0
    if oh != H:
        bh, ch, kh = _device_tables(H, oh, dev)
    assert out_slot.stride(0) == 1 and out_slot.stride(2) == 3, "out_slot must be a channels_last [3,Hp,Wp] view"
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.call("dadet_image_resample_v_normalize", _p(cur), H, ow, _p(bh), _p(ch), kh, oh, int(bool(flip)),
              int(bool(to_bgr255)), m, s, _p(out_slot), int(out_slot.stride(1) // 3), stream)


class DeviceBatchPreparer(object):
    """callable(images_u8, targets=None) -> (ImageList, targets): the GPU counterpart of `build_transforms(cfg)`
    followed by `BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY)`."""

    def __init__(self, cfg, is_train=True):
        min_size, max_size, flip_prob = transform_params(cfg, is_train)
        self.resize = Resize(min_size, max_size)
        self.flip = RandomHorizontalFlip(flip_prob)
        self.mean, self.std = cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD
        self.to_bgr255 = cfg.INPUT.TO_BGR255
        self.size_divisible = cfg.DATALOADER.SIZE_DIVISIBILITY

    def __call__(self, images_u8, targets=None, decisions=None):
        """images_u8: list of uint8 [H,W,3] RGB device tensors.  decisions (optional, for tests): list of
        ((oh, ow), flip) to use instead of drawing them."""
        n = len(images_u8)
        if decisions is None:
            decisions = []
            for img in images_u8:      # per sample: Resize.get_size first, then the flip toss (transforms/build.py:19-26)
                size = self.resize.get_size((int(img.shape[1]), int(img.shape[0])))
                decisions.append((size, self.flip.toss()))
        sizes = [d[0] for d in decisions]
        hp, wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if self.size_divisible > 0:
            d = self.size_divisible
            hp, wp = int(math.ceil(hp / d) * d), int(math.ceil(wp / d) * d)
        dev = images_u8[0].device
        batch = torch.empty((n, 3, hp, wp), dtype=torch.float32, device=dev,
                            memory_format=torch.channels_last).zero_()
        out_targets = None if targets is None else []
        for i, (img, ((oh, ow), flip)) in enumerate(zip(images_u8, decisions)):
            resize_normalize_into(img, (oh, ow), flip, self.mean, self.std, self.to_bgr255, batch[i])
            if targets is not None:
                t = targets[i].resize((ow, oh))
                if flip:
                    t = t.transpose(0)
                out_targets.append(t)
        return ImageList(batch, [(oh, ow) for (oh, ow) in sizes]), out_targets
