"""Data side of the training step (reference: maskrcnn_benchmark/data/): synthetic batches for the benchmark, the
sample transforms / samplers / collators / COCO-style datasets of the reference, and the device-side batch preparation
(Pillow-exact resize + flip + normalisation on the GPU)."""
from .build import make_da_data_loaders, make_data_loader, make_data_loader_da, make_triplet_data_loader
from .collate_batch import BatchCollator
from .transforms import build_transforms

__all__ = ["make_data_loader", "make_data_loader_da", "make_da_data_loaders", "make_triplet_data_loader", "BatchCollator", "build_transforms"]
