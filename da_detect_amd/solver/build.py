"""Optimizer / scheduler builders (reference: maskrcnn_benchmark/solver/build.py:7-30)."""
from .fused_sgd import FusedSGD
from .lr_scheduler import CosineLRScheduler, WarmupMultiStepLR


def make_optimizer(cfg, model):
    """one parameter group per tensor; biases get BIAS_LR_FACTOR x LR and WEIGHT_DECAY_BIAS (build.py:7-20)"""
    params = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
            wd = cfg.SOLVER.WEIGHT_DECAY_BIAS
        params.append({"params": [value], "lr": lr, "weight_decay": wd})
    return FusedSGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


def make_lr_scheduler(cfg, optimizer):
    return WarmupMultiStepLR(optimizer, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, warmup_factor=cfg.SOLVER.WARMUP_FACTOR,
                             warmup_iters=cfg.SOLVER.WARMUP_ITERS, warmup_method=cfg.SOLVER.WARMUP_METHOD)


def make_cosine_lr_scheduler(cfg, optimizer):
    """the schedule tools/train_net_triplet.py:67-81 builds"""
    return CosineLRScheduler(optimizer, t_initial=cfg.SOLVER.MAX_ITER, lr_min=cfg.SOLVER.LR_MIN,
                             warmup_lr_init=cfg.SOLVER.WARMUP_LR, warmup_t=cfg.SOLVER.WARMUP_ITERS,
                             t_in_epochs=False)
