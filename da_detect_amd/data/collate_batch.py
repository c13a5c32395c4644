"""Batch collators (reference: maskrcnn_benchmark/data/collate_batch.py:14-55)."""
from ..structures.image_list import to_image_list


class BatchCollator(object):
    """[(image, target, idx), ...] -> (ImageList padded to size_divisible, targets, idxs)"""

    def __init__(self, size_divisible=0):
        self.size_divisible = size_divisible

    def __call__(self, batch):
        transposed = list(zip(*batch))
        return to_image_list(transposed[0], self.size_divisible), transposed[1], transposed[2]


class BatchCollator_triplet(object):
    """samples of TripletDataset: (img_s, t_s, img_p, t_p, img_n, t_n, i1, i2, i3) -> the same nine fields batched"""

    def __init__(self, size_divisible=0):
        self.size_divisible = size_divisible

    def __call__(self, batch):
        t = list(zip(*batch))
        return (to_image_list(t[0], self.size_divisible), t[1], to_image_list(t[2], self.size_divisible), t[3],
                to_image_list(t[4], self.size_divisible), t[5], t[6], t[7], t[8])


class RawBatchCollator(object):
    """for the device-side pipeline: keeps decoded uint8 images and untransformed targets as lists"""

    def __call__(self, batch):
        transposed = list(zip(*batch))
        return list(transposed[0]), list(transposed[1]), list(transposed[2])
