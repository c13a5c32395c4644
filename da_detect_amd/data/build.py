"""Data-loader assembly (reference: maskrcnn_benchmark/data/build.py:127-419): aspect-ratio grouping, the
distributed / grouped / iteration-based sampler stack, and the source / target / auxiliary pairing of the DA
trainers.  Dataset locations come from the caller (`dataset_specs`) instead of the reference's path catalog."""
import bisect

import torch

from ..utils.comm import get_world_size
from . import samplers
from .collate_batch import BatchCollator, BatchCollator_triplet
from .datasets import COCODataset, TripletDataset
from .transforms import build_transforms


def _quantize(x, bins):
    bins = sorted(bins)
    return [bisect.bisect_right(bins, v) for v in x]


def _compute_aspect_ratios(dataset):
    out = []
    for i in range(len(dataset)):
        info = dataset.get_img_info(i)
        out.append(float(info["height"]) / float(info["width"]))
    return out


def make_data_sampler(dataset, shuffle, distributed):
    if distributed:
        return samplers.DistributedSampler(dataset, shuffle=shuffle)
    if shuffle:
        return torch.utils.data.sampler.RandomSampler(dataset)
    return torch.utils.data.sampler.SequentialSampler(dataset)


def make_batch_data_sampler(dataset, sampler, aspect_grouping, images_per_batch, num_iters=None, start_iter=0):
    """build.py:176-196"""
    if aspect_grouping:
        if not isinstance(aspect_grouping, (list, tuple)):
            aspect_grouping = [aspect_grouping]
        group_ids = _quantize(_compute_aspect_ratios(dataset), aspect_grouping)
        batch_sampler = samplers.GroupedBatchSampler(sampler, group_ids, images_per_batch, drop_uneven=False)
    else:
        batch_sampler = torch.utils.data.sampler.BatchSampler(sampler, images_per_batch, drop_last=False)
    if num_iters is not None:
        batch_sampler = samplers.IterationBasedBatchSampler(batch_sampler, num_iters, start_iter)
    return batch_sampler


def images_per_gpu(cfg, is_train, domain_share=1):
    """build.py:232-246: IMS_PER_BATCH // num_gpus, and with DOMAIN_ADAPTATION_ON every loader (source, target,
    auxiliary, or the triplet loader) takes IMS_PER_BATCH // (2 * num_gpus) samples per step (`domain_share` = 2)"""
    total = cfg.SOLVER.IMS_PER_BATCH if is_train else cfg.TEST.IMS_PER_BATCH
    world = get_world_size()
    assert total % (world * domain_share) == 0, \
        "IMS_PER_BATCH ({}) must be divisible by the number of GPUs ({}) x domains ({})".format(total, world, domain_share)
    return total // (world * domain_share)


def make_data_loader(cfg, dataset, is_train=True, is_distributed=False, start_iter=0, domain_share=1, collator=None):
    per_gpu = images_per_gpu(cfg, is_train, domain_share)
    sampler = make_data_sampler(dataset, shuffle=is_train, distributed=is_distributed)
    batch_sampler = make_batch_data_sampler(dataset, sampler, [1] if cfg.DATALOADER.ASPECT_RATIO_GROUPING else [],
                                            per_gpu, cfg.SOLVER.MAX_ITER if is_train else None, start_iter)
    if collator is None:
        collator = BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY)
    return torch.utils.data.DataLoader(dataset, num_workers=cfg.DATALOADER.NUM_WORKERS, batch_sampler=batch_sampler,
                                       collate_fn=collator)


def make_da_data_loaders(cfg, dataset_specs, is_distributed=False, start_iter=0):
    """dataset_specs = {"source": (ann_file, root), "target": (...)[, "auxiliary": (...)]} -> the loaders
    do_da_train iterates jointly (one per domain, each carrying IMS_PER_BATCH // (2 * num_gpus) images per step)"""
    tf = build_transforms(cfg, True)
    names = [n for n in ("source", "target", "auxiliary") if n in dataset_specs]
    out = []
    for name in names:
        ann, root = dataset_specs[name]
        ds = COCODataset(ann, root, remove_images_without_annotations=True, transforms=tf,
                         is_source=(name == "source"))
        out.append(make_data_loader(cfg, ds, True, is_distributed, start_iter, domain_share=2))
    return out


def make_triplet_data_loader(cfg, dataset_specs, is_distributed=False, start_iter=0):
    """one loader over index-aligned (source, target, auxiliary) samples (build.py:23-63, 332-419)"""
    tf = build_transforms(cfg, True)
    ds = [COCODataset(*dataset_specs[n], remove_images_without_annotations=True, transforms=tf,
                      is_source=(n == "source")) for n in ("source", "target", "auxiliary")]
    triplet = TripletDataset(ds)
    return make_data_loader(cfg, triplet, True, is_distributed, start_iter, domain_share=2,
                            collator=BatchCollator_triplet(cfg.DATALOADER.SIZE_DIVISIBILITY))
