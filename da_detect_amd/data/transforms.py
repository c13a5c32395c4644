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
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join("\n    {0}".format(t) for t in self.transforms) + "\n)"


class Resize(object):
    def __init__(self, min_size, max_size):
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size = min_size
        self.max_size = max_size

    def get_size(self, image_size):
        """(w, h) -> (oh, ow): shorter side to a randomly chosen min_size unless that pushes the longer side past
        max_size (transforms.py:41-62)"""
        w, h = image_size
        size = random.choice(self.min_size)
        max_size = self.max_size
        if max_size is not None:
            min_original_size = float(min((w, h)))
            max_original_size = float(max((w, h)))
            if max_original_size / min_original_size * size > max_size:
                size = int(round(max_size * min_original_size / max_original_size))
        if (w <= h and w == size) or (h <= w and h == size):
            return (h, w)
        if w < h:
            return (int(size * h / w), size)
        return (size, int(size * w / h))

    def __call__(self, image, target):
        from PIL import Image

        oh, ow = self.get_size(image.size)
        image = image.resize((ow, oh), Image.BILINEAR)      # == torchvision F.resize(image, (oh, ow))
        if target is not None:
            target = target.resize(image.size)
        return image, target


class RandomHorizontalFlip(object):
    def __init__(self, prob=0.5):
        self.prob = prob

    def toss(self):
        return random.random() < self.prob

    def __call__(self, image, target):
        from PIL import Image

        if self.toss():
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
            if target is not None:
                target = target.transpose(0)
        return image, target


class ToTensor(object):
    def __call__(self, image, target):
        arr = np.array(image, dtype=np.uint8)       # a writable copy (torch.from_numpy warns on read-only views)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).to(torch.float32).div(255)
        return t, target


class Normalize(object):
    def __init__(self, mean, std, to_bgr255=True):
        self.mean = mean
        self.std = std
        self.to_bgr255 = to_bgr255

    def __call__(self, image, target):
        if self.to_bgr255:
            image = image[[2, 1, 0]] * 255
        mean = torch.as_tensor(self.mean, dtype=image.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=image.dtype).view(-1, 1, 1)
        return (image - mean) / std, target


def transform_params(cfg, is_train=True):
    """(min_size, max_size, flip_prob) of transforms/build.py:5-14"""
    if is_train:
        return cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, 0.5
    return cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, 0


def build_transforms(cfg, is_train=True):
    min_size, max_size, flip_prob = transform_params(cfg, is_train)
    return Compose([Resize(min_size, max_size), RandomHorizontalFlip(flip_prob), ToTensor(),
                    Normalize(mean=cfg.INPUT.PIXEL_MEAN, std=cfg.INPUT.PIXEL_STD, to_bgr255=cfg.INPUT.TO_BGR255)])
