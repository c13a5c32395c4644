"""Synthetic Cityscapes-shaped batches: images = rand*255 - PIXEL_MEAN (BGR, std 1),
8-20 seeded boxes per image with min side 16 px, labels uniform in 1..NUM_CLASSES-1, `is_source` all-True for
the first image and all-False for the others (target / auxiliary domains).  Everything is generated on the CPU
from an explicit seed so that any device (and the CPU baseline) sees identical inputs."""
import torch

from ..structures.bounding_box import BoxList
from ..structures.image_list import to_image_list


def make_targets(num_images, height, width, num_classes, seed):
    g = torch.Generator().manual_seed(seed)
    targets = []
    for i in range(num_images):
        n = int(torch.randint(8, 21, (1,), generator=g))
        x1 = torch.rand(n, generator=g) * (width - 17)
        y1 = torch.rand(n, generator=g) * (height - 17)
        w = 16 + torch.rand(n, generator=g) * (width * 0.4)
        h = 16 + torch.rand(n, generator=g) * (height * 0.4)
        boxes = torch.stack([x1, y1, torch.minimum(x1 + w, torch.tensor(width - 1.0)),
                             torch.minimum(y1 + h, torch.tensor(height - 1.0))], dim=1)
        t = BoxList(boxes, (width, height), mode="xyxy")
        t.add_field("labels", torch.randint(1, num_classes, (n,), generator=g))
        t.add_field("is_source", torch.full((n,), i == 0, dtype=torch.bool))
        targets.append(t)
    return targets


def make_batch(cfg, num_images, height, width, seed, device):
    """-> (ImageList on `device`, list[BoxList] on `device`)"""
    g = torch.Generator().manual_seed(seed + 7919)
    mean = torch.tensor(cfg.INPUT.PIXEL_MEAN, dtype=torch.float32).view(3, 1, 1)
    images = [torch.rand((3, height, width), generator=g) * 255.0 - mean for _ in range(num_images)]
    image_list = to_image_list(images, cfg.DATALOADER.SIZE_DIVISIBILITY).to(device)
    targets = [t.to(device) for t in make_targets(num_images, height, width, cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES, seed)]
    return image_list, targets
