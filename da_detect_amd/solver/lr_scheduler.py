"""Learning-rate schedules of the reference trainers.

WarmupMultiStepLR: reference maskrcnn_benchmark/solver/lr_scheduler.py:10-52.
CosineLRScheduler: restatement of the timm scheduler the live trainer swaps in
(reference call site tools/train_net_triplet.py:67-81; timm is an un-vendored, unpinned dependency —
"parity unpinned", restated from the call site: linear warm-up from warmup_lr_init to the base LR over
warmup_t updates (warmup_prefix off), then cosine decay to lr_min at t_initial, driven per iteration by
`step_update(num_updates)`; `step(epoch)` is a no-op when t_in_epochs is False).
"""
import math
from bisect import bisect_right

import torch


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        if not list(milestones) == sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}", milestones)
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted got {}".format(warmup_method))
        self.milestones = milestones
        self.gamma = gamma
        self.warmup_factor = warmup_factor
        self.warmup_iters = warmup_iters
        self.warmup_method = warmup_method
        super(WarmupMultiStepLR, self).__init__(optimizer, last_epoch)

    def get_lr(self):
        warmup = 1
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                warmup = self.warmup_factor
            else:
                alpha = float(self.last_epoch) / self.warmup_iters
                warmup = self.warmup_factor * (1 - alpha) + alpha
        decay = self.gamma ** bisect_right(self.milestones, self.last_epoch)
        return [base_lr * warmup * decay for base_lr in self.base_lrs]


class CosineLRScheduler(object):
    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_lr_init=0.0, warmup_t=0, t_in_epochs=False,
                 **_unused):
        self.optimizer = optimizer
        self.t_initial = t_initial
        self.lr_min = lr_min
        self.warmup_lr_init = warmup_lr_init
        self.warmup_t = warmup_t
        self.t_in_epochs = t_in_epochs
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._apply([warmup_lr_init for _ in self.base_values])
        else:
            self.warmup_steps = [1 for _ in self.base_values]

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial))
                    for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def _apply(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def step(self, epoch, metric=None):
        if self.t_in_epochs:
            self._apply(self._get_lr(epoch))

    def step_update(self, num_updates, metric=None):
        if not self.t_in_epochs:
            self._apply(self._get_lr(num_updates))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)
