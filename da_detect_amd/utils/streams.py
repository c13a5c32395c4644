"""Side-stream sections for the host-synchronising bookkeeping of the training step.

Target assignment and proposal sampling (reference: rpn/loss.py:57-123, box_head/loss.py:55-163) are a few hundred
tiny launches with device->host round trips (nonzero, randperm sizes) in between.  On the compute stream every one of
those round trips waits for ALL queued work — backbone, RPN-head backward — and the GPU then idles while the host
issues the next tiny launch.  Issued on a side stream they wait for that stream alone, so the compute stream keeps
executing the large kernels queued in front of them.  Ordering is explicit: the section starts after an event of
the compute stream (its inputs exist), the compute stream waits for the section's end before it continues, and
every tensor handed across is registered with the caching allocator (record_stream)."""
import contextlib

import torch

import os

_SIDE = {}
_HIGH_PRIORITY = os.environ.get("DADET_SIDE_PRIORITY", "1") == "1"


def side_stream(device, which=0):
    key = (device.type, device.index, which)
    if key not in _SIDE:
        # streams 0 / 1 carry the latency-bound bookkeeping (single-workgroup NMS sweeps, sampling): high priority, so
        # their workgroups are dispatched ahead of the GEMM waves they run beside; the weight-gradient lane (2) is not
        prio = -1 if (which < 2 and _HIGH_PRIORITY) else 0
        _SIDE[key] = torch.cuda.Stream(device, priority=prio)
    return _SIDE[key]


def other_stream(device):
    """a stream different from the current one: side stream 0, or side stream 1 when already running on 0"""
    cur = torch.cuda.current_stream(device)
    s0 = side_stream(device, 0)
    return s0 if cur != s0 else side_stream(device, 1)


def record(obj, stream):
    """register every tensor reachable from obj (tensor / BoxList / dict / list) as used on `stream`"""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            record(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            record(v, stream)
    elif hasattr(obj, "bbox") and hasattr(obj, "extra_fields"):
        record(obj.bbox, stream)
        record(obj.extra_fields, stream)


@contextlib.contextmanager
def side_section(device, after=None):
    """with side_section(dev, after=event) as done: ... ; done(outputs) registers the hand-over.  On a CPU device it
    is a no-op.  `after`: event of the compute stream the section must wait for (None: everything queued so far)."""
    if device.type != "cuda":
        yield lambda *outs: None
        return
    main = torch.cuda.current_stream(device)
    side = side_stream(device)
    if after is not None:
        side.wait_event(after)
    else:
        side.wait_stream(main)
    handed = []
    with torch.cuda.stream(side):
        yield lambda *outs: handed.extend(outs)
    record(handed, main)
    main.wait_stream(side)


# ---------------------------------------------------------------------------------------------------------------
# weight gradients beside data gradients
import os  # noqa: E402

WGRAD_OVERLAP = os.environ.get("DADET_WGRAD_STREAM", "1") == "1"


class WgradLane(object):
    """Inside one backward node the weight-gradient GEMMs do not feed the data-gradient chain.  Issued on their own
    stream they run BESIDE the dgrad kernels, so the partial last waves and launch gaps of one fill with the other's
    workgroups.  `run(fn, *tensors)` queues fn() there once the tensors exist on the compute stream; `join()` makes
    the compute stream wait before the gradients are handed to autograd."""

    def __init__(self, device):
        self.on = WGRAD_OVERLAP and device.type == "cuda"
        if self.on:
            self.main = torch.cuda.current_stream(device)
            self.lane = side_stream(device, 2)
            self.out = []

    def run(self, fn, *inputs):
        if not self.on:
            return fn()
        self.lane.wait_event(self.main.record_event())
        with torch.cuda.stream(self.lane):
            res = fn()
        for t in inputs:
            t.record_stream(self.lane)
        self.out.append(res)
        return res

    def join(self):
        if self.on and self.out:
            record(self.out, self.main)
            self.main.wait_stream(self.lane)
