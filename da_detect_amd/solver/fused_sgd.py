"""SGD with momentum as ONE multi-tensor HIP launch per step.

Numerically torch.optim.SGD (dampening 0, no nesterov): d = g + wd * p; buf = d (first step) or
momentum * buf + d; p -= lr * buf — applied to the reference's one-param-group-per-tensor layout
(reference: maskrcnn_benchmark/solver/build.py:7-20, engine/trainer.py:237-239) by dadet_sgd_step, which reads a
device-resident table of (p, g, buf, numel, lr, weight_decay) entries.  It subclasses torch.optim.Optimizer so
LR schedulers and checkpoint code that walk `param_groups` / `state_dict()` keep working.
"""
import ctypes
import os

import torch

from .. import _lib
from .._lib import SgdEntry


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay)
        super(FusedSGD, self).__init__(params, defaults)
        self._table = None       # (key, device tensor, max_numel, n)
        self._steps = 0
        self.reducer = None      # optional parallel.reducer.BucketedGradReducer owning the .grad storage
        # one update launch per gradient bucket as its collective completes (several ranks); DADET_SGD_PER_BUCKET=0: one
        # launch behind all collectives
        self.per_bucket = os.environ.get("DADET_SGD_PER_BUCKET", "1") == "1"
        self._bucket_plan = None

    def attach_reducer(self, reducer):
        """gradients live in the reducer's flat buckets: zero_grad() clears them in place (stable pointers)
        and step() first completes the outstanding all-reduces"""
        self.reducer = reducer
        # with persistent zeroed .grad buffers the weight-gradient kernels may accumulate into them directly
        from ..utils import streams
        streams.enable_direct_wgrad(True)

    def zero_grad(self, set_to_none=True):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            super(FusedSGD, self).zero_grad(set_to_none=set_to_none)

    def _step_bucket(self, index, bucket, grad_scale):
        """the update of ONE gradient bucket's tensors: rows [first, first + n) of a table that holds every updated tensor,
        ordered by bucket (built once per set of (pointers, lr, weight decay); dadet_sgd_step takes the table by pointer)"""
        plan = self._bucket_table()
        if plan is None:
            return
        dev_table, ranges, device, momentum = plan
        first, n, max_numel = ranges[index]
        if n:
            _lib.call("dadet_sgd_step", ctypes.c_void_p(dev_table.data_ptr() + first * ctypes.sizeof(SgdEntry)), n,
                      ctypes.c_int64(max_numel), float(momentum), 0, float(grad_scale),
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def _bucket_table(self):
        red = self.reducer
        order = {id(p): i for i, b in enumerate(red.buckets) for p in b["params"]}
        entries = sorted(self._entries(), key=lambda e: order.get(id(e[0]), len(red.buckets)))
        if not entries:
            self._bucket_plan = None
            return None
        momentum = entries[0][3]
        assert all(e[3] == momentum for e in entries), "FusedSGD: one momentum for all groups"
        rows, owner = [], []
        for p, lr, wd, _ in entries:
            if not p.is_cuda:
                raise _lib.DadetError("FusedSGD runs on the HIP device only")
            if id(p) not in order:
                raise _lib.DadetError("FusedSGD: a parameter outside the reducer's buckets cannot be updated bucket by bucket")
            st = self.state[p]
            if "momentum_buffer" not in st:
                st["momentum_buffer"] = torch.empty_strided(p.size(), p.stride(), dtype=p.dtype, device=p.device).zero_()
            rows.append((p.data_ptr(), p.grad.data_ptr(), st["momentum_buffer"].data_ptr(), p.numel(), lr, wd))
            owner.append(order[id(p)])
        key = tuple(rows)
        if self._bucket_plan is None or self._bucket_plan[0] != key:
            arr = (SgdEntry * len(rows))()
            for i, r in enumerate(rows):
                arr[i].p, arr[i].g, arr[i].buf, arr[i].numel, arr[i].lr, arr[i].weight_decay = r
            device = entries[0][0].device
            # (the lr changes every step under a per-iteration schedule: two pinned staging buffers, as in step())
            nbytes = ctypes.sizeof(arr)
            stage = getattr(self, "_bstage", None)
            if stage is None or stage[0][0].numel() != nbytes:
                stage = self._bstage = [(torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                                         torch.empty(nbytes, dtype=torch.uint8, device=device),
                                         torch.cuda.Event()) for _ in range(2)]
            host, dev, done = stage[self._steps % 2]
            done.synchronize()
            ctypes.memmove(host.data_ptr(), ctypes.addressof(arr), nbytes)
            dev.copy_(host, non_blocking=True)
            done.record()
            ranges = []
            for b in range(len(red.buckets)):
                idx = [i for i, o in enumerate(owner) if o == b]
                ranges.append((idx[0], len(idx), max(rows[i][3] for i in idx)) if idx else (0, 0, 0))
            self._bucket_plan = (key, dev, device, ranges, momentum)
        _, dev, device, ranges, momentum = self._bucket_plan
        return dev, ranges, device, momentum

    def _entries(self):
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                # with the reducer every .grad is a persistent (zeroed) bucket view: a parameter that received no
                # gradient this step must still be SKIPPED, as torch.optim.SGD skips `grad is None` (no weight decay,
                # no momentum update) — e.g. the instance head when its loss weight is 0
                # (with several ranks: no gradient on ANY rank — reducer.update_ids)
                if self.reducer is not None and id(p) not in self._update_ids:
                    continue
                out.append((p, group["lr"], group["weight_decay"], group["momentum"]))
        return out

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        handed = None
        if self.reducer is not None:
            # the 1 / world of the gradient mean rides in the kernel's gradient read (g * grad_scale: exact for the
            # power-of-two worlds of one node) instead of one multiply launch per bucket
            if self.per_bucket and self.reducer.can_hand_over_buckets():
                # (round 6) one SGD launch per gradient bucket, issued as that bucket's all-reduce completes: the last
                # bucket's collective — the exposed tail of a ~6 ms backward — runs beside the earlier buckets' updates
                self._update_ids = self.reducer.update_ids()
                handed = []
                self.reducer.finalize(mean=False, per_bucket=lambda i, b: handed.append(i) or self._step_bucket(
                    i, b, grad_scale / self.reducer.world_size))
            else:
                self.reducer.finalize(mean=False)
            grad_scale = grad_scale * self.reducer.mean_scale
            self._update_ids = self.reducer.update_ids()
        if handed is not None:
            self._steps += 1
            from .. import _C
            if self._bucket_plan is not None:
                _C.bump_weight_epoch(self._bucket_plan[2], trained_only=True)
            return loss
        entries = self._entries()
        if not entries:
            return loss
        momentum = entries[0][3]
        assert all(e[3] == momentum for e in entries), "FusedSGD: one momentum for all groups"
        first = []
        rows = []
        for p, lr, wd, _ in entries:
            if not p.is_cuda:
                raise _lib.DadetError("FusedSGD runs on the HIP device only")
            g = p.grad
            if g.stride() != p.stride() or not _dense(p):
                # the kernel walks raw storage: gradient must share the parameter's physical layout
                g = _like_layout(g, p)
                p.grad = g
            st = self.state[p]
            if "momentum_buffer" not in st:
                # a zero buffer makes the general update `buf = momentum * buf + d` equal torch's first-step rule
                # `buf = d` exactly, also for a parameter that joins later (first gradient after some steps)
                st["momentum_buffer"] = torch.empty_strided(p.size(), p.stride(), dtype=p.dtype,
                                                            device=p.device).zero_()
            first.append(False)
            rows.append((p.data_ptr(), g.data_ptr(), st["momentum_buffer"].data_ptr(), p.numel(), lr, wd))
        key = tuple(rows)
        if self._table is None or self._table[0] != key:
            # The table changes every step under a per-iteration schedule (the cosine schedule of the reference's
            # trainer rewrites every group's lr): it is staged in one of two pinned host buffers and uploaded on the
            # compute stream without blocking the host (a pageable `.to(device)` here stalled the host once per step
            # until the stream had drained); the kernel of step t reads device copy t % 2.
            arr = (SgdEntry * len(rows))()
            for i, r in enumerate(rows):
                arr[i].p, arr[i].g, arr[i].buf, arr[i].numel, arr[i].lr, arr[i].weight_decay = r
            nbytes = ctypes.sizeof(arr)
            slot = self._steps % 2
            stage = getattr(self, "_stage", None)
            if stage is None or stage[0][0].numel() != nbytes:
                device = entries[0][0].device
                stage = self._stage = [(torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                                        torch.empty(nbytes, dtype=torch.uint8, device=device),
                                        torch.cuda.Event()) for _ in range(2)]
            host, dev, done = stage[slot]
            done.synchronize()            # the upload issued two steps ago from this buffer (long finished)
            ctypes.memmove(host.data_ptr(), ctypes.addressof(arr), nbytes)
            dev.copy_(host, non_blocking=True)
            done.record()
            self._table = (key, dev, max(r[3] for r in rows), len(rows))
        _, dev, max_numel, n = self._table
        _lib.call("dadet_sgd_step", ctypes.c_void_p(dev.data_ptr()), n, ctypes.c_int64(max_numel),
                  float(momentum), 1 if all(first) else 0, float(grad_scale),
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        self._steps += 1
        from .. import _C
        # parameters changed through raw pointers: the cached transposed weights are refreshed here, in one launch
        # (trained_only: frozen weights did not change — their maxima are not re-measured)
        _C.bump_weight_epoch(entries[0][0].device, trained_only=True)
        return loss


def _dense(t):
    return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)


def _like_layout(g, p):
    out = torch.empty_strided(p.size(), p.stride(), dtype=g.dtype, device=g.device)
    out.copy_(g)
    return out
