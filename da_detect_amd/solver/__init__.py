from .build import make_cosine_lr_scheduler, make_lr_scheduler, make_optimizer
from .fused_sgd import FusedSGD
from .lr_scheduler import CosineLRScheduler, WarmupMultiStepLR

__all__ = ["make_optimizer", "make_lr_scheduler", "make_cosine_lr_scheduler", "FusedSGD", "WarmupMultiStepLR",
           "CosineLRScheduler"]
