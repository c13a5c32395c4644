"""Sample transforms — host-side API of the reference (maskrcnn_benchmark/data/transforms/transforms.py:17-97,
transforms/build.py:5-28) without torchvision: the image operations go to Pillow directly (which is what torchvision's
functional ops call for PIL images) and `ToTensor` / `Normalize` are a few lines of torch.  The training path does not
need these per-sample host transforms: `data/device_prep.py` applies the same Resize -> flip -> ToTensor -> Normalize
chain on the GPU, bit for bit; both draw their random decisions through `Resize.get_size` / `RandomHorizontalFlip.toss`
in the same order, so they are interchangeable."""
import random

import numpy as np
import torch


class Compose(object):
    """chain of (image, target) -> (image, target) callables"""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, image, target):
        for step in self.transforms:
            image, target = step(image, target)
        return image, target

    def __repr__(self):
        return "Compose(%s)" % ", ".join(type(t).__name__ for t in self.transforms)


class Resize(object):
    """shorter side to one of `min_size` (drawn per sample), longer side capped at `max_size`"""

    def __init__(self, min_size, max_size):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size

    def get_size(self, image_size):
        """(w, h) -> (out_h, out_w), the arithmetic of transforms.py:41-62 (float ratio first, truncation of the
        longer side, rounding only when the cap applies) written on (short, long) instead of per orientation"""
        w, h = image_size
        short, long_ = (w, h) if w <= h else (h, w)
        goal = random.choice(self.min_size)
        if self.max_size is not None and float(long_) / float(short) * goal > self.max_size:
            goal = int(round(self.max_size * float(short) / float(long_)))
        if short == goal:
            return (h, w)
        stretched = int(goal * long_ / short)
        return (stretched, goal) if w < h else (goal, stretched)

    def __call__(self, image, target):
        from PIL import Image

        out_h, out_w = self.get_size(image.size)
        image = image.resize((out_w, out_h), Image.BILINEAR)   # what torchvision's F.resize does for a PIL image
        return image, (target.resize(image.size) if target is not None else None)


class RandomHorizontalFlip(object):
    def __init__(self, prob=0.5):
        self.prob = prob

    def toss(self):
        """one draw per sample from the global `random` stream (the device path calls this too, in the same order)"""
        return random.random() < self.prob

    def __call__(self, image, target):
        if not self.toss():
            return image, target
        from PIL import Image

        flipped = image.transpose(Image.FLIP_LEFT_RIGHT)
        return flipped, (target.transpose(0) if target is not None else None)


class ToTensor(object):
    """uint8 HWC image -> float CHW in [0, 1]"""

    def __call__(self, image, target):
        pixels = np.array(image, dtype=np.uint8)    # a writable copy (torch.from_numpy warns on read-only views)
        pixels = pixels[:, :, None] if pixels.ndim == 2 else pixels
        chw = torch.from_numpy(np.ascontiguousarray(pixels)).permute(2, 0, 1)
        return chw.to(torch.float32).div(255), target


class Normalize(object):
    """optionally RGB [0,1] -> BGR [0,255], then (x - mean) / std per channel"""

    def __init__(self, mean, std, to_bgr255=True):
        self.mean, self.std, self.to_bgr255 = mean, std, to_bgr255

    def __call__(self, image, target):
        x = image[[2, 1, 0]] * 255 if self.to_bgr255 else image
        shift = torch.as_tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
        scale = torch.as_tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
        return (x - shift) / scale, target


def transform_params(cfg, is_train=True):
    """(min_size, max_size, flip_prob) of transforms/build.py:5-14"""
    if is_train:
        return cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, 0.5
    return cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, 0


def build_transforms(cfg, is_train=True):
    min_size, max_size, flip_prob = transform_params(cfg, is_train)
    return Compose([Resize(min_size, max_size), RandomHorizontalFlip(flip_prob), ToTensor(),
                    Normalize(mean=cfg.INPUT.PIXEL_MEAN, std=cfg.INPUT.PIXEL_STD, to_bgr255=cfg.INPUT.TO_BGR255)])
