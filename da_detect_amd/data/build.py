"""Data-loader assembly (reference: maskrcnn_benchmark/data/build.py:67-419): dataset construction from the path
catalog, aspect-ratio grouping, the distributed / grouped / iteration-based sampler stack, and the source / target /
auxiliary pairing of the DA trainers.

Two entry families over the same machinery:
  * the reference's own — `make_data_loader(cfg, is_train, is_source, is_negative, is_distributed, is_for_period,
    start_iter)` and `make_data_loader_da(cfg, is_source=[...], ...)` — dataset NAMES from cfg.DATASETS resolved through
    `DatasetCatalog` of the file cfg.PATHS_CATALOG (what tools/train_net_triplet.py:108-170 calls);
  * explicit (annotation file, image root) pairs — `make_da_data_loaders`, `make_triplet_data_loader`
    (tools/train_net_da.py).
"""
import bisect
import logging

import torch

from ..utils.comm import get_world_size
from ..utils.imports import load_paths_catalog
from . import datasets as D
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


def _loader_for(cfg, dataset, per_gpu, shuffle, is_distributed, num_iters, start_iter, collator):
    sampler = make_data_sampler(dataset, shuffle, is_distributed)
    batch_sampler = make_batch_data_sampler(dataset, sampler, [1] if cfg.DATALOADER.ASPECT_RATIO_GROUPING else [],
                                            per_gpu, num_iters, start_iter)
    return torch.utils.data.DataLoader(dataset, num_workers=cfg.DATALOADER.NUM_WORKERS, batch_sampler=batch_sampler,
                                       collate_fn=collator)


# ------------------------------------------------------------------------------------ the reference's entry points
def _catalog(cfg):
    return load_paths_catalog(cfg.PATHS_CATALOG).DatasetCatalog


def _from_catalog(name, catalog, transforms, is_train, is_source):
    """one dataset from its catalog entry (build.py:84-103, 141-169)"""
    data = catalog.get(name)
    factory = getattr(D, data["factory"])
    args = dict(data["args"])
    if data["factory"] == "COCODataset":
        args["remove_images_without_annotations"] = is_train
    args["transforms"] = transforms
    args["is_source"] = is_source
    return factory(**args)


def build_dataset(dataset_list, transforms, dataset_catalog, is_train=True, is_source=True):
    """training: ONE (possibly concatenated) dataset in a list; testing: one per name (build.py:124-181)"""
    if not isinstance(dataset_list, (list, tuple)):
        raise RuntimeError("dataset_list should be a list of strings, got {}".format(dataset_list))
    datasets = [_from_catalog(n, dataset_catalog, transforms, is_train, is_source) for n in dataset_list]
    if not is_train:
        return datasets
    return [datasets[0] if len(datasets) == 1 else D.ConcatDataset(datasets)]


def build_dataset_da(dataset_list, transforms, dataset_catalog, is_source, is_train=True):
    """`is_source` is a list parallel to `dataset_list` = [source, target, auxiliary]; training: the index-aligned
    triplet dataset in a list, testing: the three datasets (build.py:67-121)"""
    if not isinstance(dataset_list, (list, tuple)):
        raise RuntimeError("dataset_list should be a list of strings, got {}".format(dataset_list))
    datasets = [_from_catalog(n, dataset_catalog, transforms, is_train, src)
                for n, src in zip(dataset_list, is_source)]
    if not is_train:
        return datasets
    return [TripletDataset(datasets)]


def _batch_plan(cfg, is_train, is_distributed, start_iter):
    """-> (images per GPU and loader, shuffle, number of iterations, start_iter) — build.py:233-258"""
    if is_train:
        per_gpu = images_per_gpu(cfg, True, 2 if cfg.MODEL.DOMAIN_ADAPTATION_ON else 1)
        plan = per_gpu, True, cfg.SOLVER.MAX_ITER, start_iter
    else:
        plan = images_per_gpu(cfg, False), bool(is_distributed), None, 0
    if plan[0] > 1:
        logging.getLogger(__name__).warning(
            "More than one image per GPU and loader: memory grows with it; reduce SOLVER.IMS_PER_BATCH / "
            "TEST.IMS_PER_BATCH if needed (and rescale the learning rate and schedule for training).")
    return plan


def make_data_loader(cfg, is_train=True, is_source=True, is_negative=False, is_distributed=False, is_for_period=False,
                     start_iter=0):
    """reference signature (build.py:232-329).  Training: the loader of ONE domain — SOURCE_TRAIN, TARGET_TRAIN or
    TARGET_TRAIN_negative under DOMAIN_ADAPTATION_ON (each IMS_PER_BATCH // (2 * GPUs) images per step), TRAIN
    otherwise; testing: a list with one loader per cfg.DATASETS.TEST entry."""
    per_gpu, shuffle, num_iters, start_iter = _batch_plan(cfg, is_train, is_distributed, start_iter)
    if not is_train:
        names = cfg.DATASETS.TEST
    elif not cfg.MODEL.DOMAIN_ADAPTATION_ON:
        names = cfg.DATASETS.TRAIN
    elif is_source:
        names = cfg.DATASETS.SOURCE_TRAIN
    else:
        names = cfg.DATASETS.TARGET_TRAIN_negative if is_negative else cfg.DATASETS.TARGET_TRAIN
    datasets = build_dataset(list(names), build_transforms(cfg, is_train), _catalog(cfg), is_train, is_source)
    loaders = [_loader_for(cfg, ds, per_gpu, shuffle, is_distributed, num_iters, start_iter,
                           BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY)) for ds in datasets]
    if is_train:
        assert len(loaders) == 1
        return loaders[0]
    return loaders


def make_data_loader_da(cfg, is_source, is_train=True, is_negative=False, is_distributed=False, is_for_period=False,
                        start_iter=0):
    """reference signature (build.py:332-419): the ALIGNED triplet loader over (SOURCE_TRAIN[0], TARGET_TRAIN[0],
    TARGET_TRAIN_negative[0]); `is_source` = [True, False, False] flags the three domains.  Every batch is the nine
    fields of BatchCollator_triplet."""
    per_gpu, shuffle, num_iters, start_iter = _batch_plan(cfg, is_train, is_distributed, start_iter)
    if is_train:
        names = [cfg.DATASETS.SOURCE_TRAIN[0], cfg.DATASETS.TARGET_TRAIN[0], cfg.DATASETS.TARGET_TRAIN_negative[0]]
    else:
        names = list(cfg.DATASETS.TEST)
    single = is_train or is_for_period
    datasets = build_dataset_da(names, build_transforms(cfg, is_train), _catalog(cfg), is_source, single)
    loaders = [_loader_for(cfg, ds, per_gpu, shuffle, is_distributed, num_iters, start_iter,
                           BatchCollator_triplet(cfg.DATALOADER.SIZE_DIVISIBILITY)) for ds in datasets]
    if single:
        assert len(loaders) == 1
        return loaders[0]
    return loaders


# ------------------------------------------------------------------------- explicit (annotation file, image root) pairs
def _train_loader(cfg, dataset, is_distributed, start_iter, collator):
    return _loader_for(cfg, dataset, images_per_gpu(cfg, True, 2), True, is_distributed, cfg.SOLVER.MAX_ITER,
                       start_iter, collator)


def make_test_data_loader(cfg, dataset, is_distributed=False):
    """evaluation loader over an already constructed dataset (sequential, or sharded by rank when distributed)"""
    return _loader_for(cfg, dataset, images_per_gpu(cfg, False), bool(is_distributed), is_distributed, None, 0,
                       BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY))


def make_da_data_loaders(cfg, dataset_specs, is_distributed=False, start_iter=0):
    """dataset_specs = {"source": (ann_file, root), "target": (...)[, "auxiliary": (...)]} -> one loader per domain,
    iterated jointly by do_da_train (each carrying IMS_PER_BATCH // (2 * num_gpus) images per step)"""
    tf = build_transforms(cfg, True)
    out = []
    for name in [n for n in ("source", "target", "auxiliary") if n in dataset_specs]:
        ann, root = dataset_specs[name]
        ds = COCODataset(ann, root, remove_images_without_annotations=True, transforms=tf,
                         is_source=(name == "source"))
        out.append(_train_loader(cfg, ds, is_distributed, start_iter, BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY)))
    return out


def make_triplet_data_loader(cfg, dataset_specs, is_distributed=False, start_iter=0):
    """one loader over index-aligned (source, target, auxiliary) samples (build.py:23-63, 332-419)"""
    tf = build_transforms(cfg, True)
    ds = [COCODataset(*dataset_specs[n], remove_images_without_annotations=True, transforms=tf,
                      is_source=(n == "source")) for n in ("source", "target", "auxiliary")]
    return _train_loader(cfg, TripletDataset(ds), is_distributed, start_iter,
                         BatchCollator_triplet(cfg.DATALOADER.SIZE_DIVISIBILITY))
