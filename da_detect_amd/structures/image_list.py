"""ImageList — a zero-padded batch of images with their true sizes
(reference: maskrcnn_benchmark/structures/image_list.py:15-91)."""
import math

import torch


class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes  # list of (height, width)

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)

    def __add__(self, other):
        """batch concatenation with re-padding to the larger canvas (image_list.py:36-47); this is how the
        trainer joins source / target / auxiliary batches (engine/trainer.py:215,223)."""
        a, b = self.tensors, other.tensors
        shape = [a.shape[0] + b.shape[0]] + [max(x, y) for x, y in zip(a.shape[1:], b.shape[1:])]
        out = a.new_zeros(shape)
        out[: a.shape[0], : a.shape[1], : a.shape[2], : a.shape[3]].copy_(a)
        out[a.shape[0]:, : b.shape[1], : b.shape[2], : b.shape[3]].copy_(b)
        return ImageList(out, list(self.image_sizes) + list(other.image_sizes))


def to_image_list(tensors, size_divisible=0):
    """ImageList | 4-D tensor | list of 3-D tensors -> ImageList, padding to a multiple of
    `size_divisible` (image_list.py:49-91)."""
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors]
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        assert tensors.dim() == 4
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        max_size = [max(s) for s in zip(*[img.shape for img in tensors])]
        if size_divisible > 0:
            max_size[1] = int(math.ceil(max_size[1] / size_divisible) * size_divisible)
            max_size[2] = int(math.ceil(max_size[2] / size_divisible) * size_divisible)
        batched = tensors[0].new_zeros([len(tensors)] + max_size)
        for img, pad in zip(tensors, batched):
            pad[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        return ImageList(batched, [im.shape[-2:] for im in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
