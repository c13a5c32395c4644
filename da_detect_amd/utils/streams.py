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
_HIGH_PRIORITY = True


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
    elif getattr(type(obj), "is_pending_proposals", False):
        record(obj.pending_tensors(), stream)      # (touching .bbox would materialise it: a host round trip)
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

# Two GEMM streams were worth 2 ms of a 37 ms step when they were introduced (partial last waves and launch gaps of one
# GEMM filled with the other's workgroups).  With the weight-gradient split plan, the direct accumulation and the tile
# rules of the later work the same switch costs 2 - 9% (img_only 64.0 vs 65.8, da 60.9 vs 63.1, R-101-FPN+DCN 19.3 vs
# 21.0 images/s on one box): two 1000-workgroup GEMMs sharing the CUs run slower than one after the other.  Off by
# default since then; DADET_WGRAD_STREAM=1 restores the lane.
WGRAD_OVERLAP = os.environ.get("DADET_WGRAD_STREAM", "0") == "1"
# GEMMs of at most this many rows (N * H * W of the gradient map) use the lane even when it is off in general
WGRAD_LANE_ROWS = int(os.environ.get("DADET_WGRAD_LANE_ROWS", "0"))


def lane_in_use():
    """some weight gradients may be accumulated on the lane stream"""
    return WGRAD_OVERLAP or WGRAD_LANE_ROWS > 0


class WgradLane(object):
    """Inside one backward node the weight-gradient GEMMs do not feed the data-gradient chain.  Issued on their own
    stream they run BESIDE the dgrad kernels, so the partial last waves and launch gaps of one fill with the other's
    workgroups.  `run(fn, *tensors)` queues fn() there once the tensors exist on the compute stream; `join()` makes
    the compute stream wait before the gradients are handed to autograd."""

    def __init__(self, device, rows=None):
        self.on = device.type == "cuda" and (WGRAD_OVERLAP or (rows is not None and rows <= WGRAD_LANE_ROWS))
        self.plain_used = False   # some weight gradient of this node is returned to autograd instead of accumulated
        if self.on:
            self.main = torch.cuda.current_stream(device)
            self.lane = side_stream(device, 2)
            self.out = []

    @staticmethod
    def _resolve_maxima(inputs):
        """Contraction mode 4: every GEMM operand carries its largest magnitude in a slot; an operand without one is
        MEASURED (a dadet_amax pass) by the first GEMM that asks.  That must happen here, on the compute stream, before the
        work moves to the lane: a pass issued on the lane would attach a slot that the compute stream's next GEMM on the
        same tensor reads with no ordering against it — e.g. the offset branch's gradient of a deformable block feeds a
        lane weight gradient AND the main stream's data gradient (ADVICE, round 4)."""
        from .. import _C, amax

        if not _C._mode4():
            return
        for t in inputs:
            # (dense tensors only: the GEMM wrappers copy anything else into a fresh NHWC tensor, which is measured where it
            # is made)
            if isinstance(t, torch.Tensor) and t.dtype == torch.float32 and (
                    t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
                amax.ptr(t)

    def run(self, fn, *inputs):
        if not self.on:
            return fn()
        self._resolve_maxima(inputs)
        self.lane.wait_event(self.main.record_event())
        with torch.cuda.stream(self.lane):
            res = fn()
        for t in inputs:
            t.record_stream(self.lane)
        self.out.append(res)
        return res

    def run_into(self, param, direct_fn, plain_fn, *inputs):
        """weight gradient of `param`: with direct accumulation enabled (enable_direct_wgrad) direct_fn(param.grad)
        ADDS it into the persistent gradient buffer on the lane and None is returned — autograd then has nothing to
        accumulate (its post-accumulate hooks still fire once every use of the parameter has run its backward), and
        the compute stream never waits for the lane until join_wgrad_lane().  Otherwise: run(plain_fn)."""
        tgt = direct_grad_target(param)
        if tgt is None:
            self.plain_used = True
            return self.run(plain_fn, *inputs)
        if not self.on:
            direct_fn(tgt)
            return None
        self._resolve_maxima(inputs)
        self.lane.wait_event(self.main.record_event())
        with torch.cuda.stream(self.lane):
            direct_fn(tgt)
        for t in inputs:
            t.record_stream(self.lane)
        return None

    def run_group(self, fn, *inputs):
        """fn() issues the weight gradients of several parameters straight into their persistent gradient buffers (one
        grouped launch, _C.conv_wgrad_group): on the lane when it is on — the same hand-over as run_into's direct path —
        else on the current stream"""
        if not self.on:
            fn()
            return
        self._resolve_maxima(inputs)
        self.lane.wait_event(self.main.record_event())
        with torch.cuda.stream(self.lane):
            fn()
        for t in inputs:
            t.record_stream(self.lane)

    def reduce_batch(self, batch, now=False):
        """the batched reduction pass of the weight gradients queued through this lane object (same stream as their
        GEMMs: the lane when it is on, else the current stream).  When every gradient of the batch is accumulated straight
        into its persistent buffer (run_into's direct path) nothing reads the result before the gradients' consumers
        (collectives, optimizer) ask for them through join_wgrad_lane / flush_wgrad_reductions: the pass is then
        DEFERRED and merged with the other blocks' into a few chip-filling launches at that point, instead of one
        10 - 40 us launch per block in the middle of the backward GEMM chain (rocprofv3 timeline of round 3: 13 launches,
        0.49 ms per step with nothing else running).  now=True: results handed back to autograd — reduce at once."""
        if not batch:
            return
        from .. import _C

        if DEFER_WGRAD_REDUCE and DIRECT_WGRAD and not now and not self.plain_used:
            # One merged launch sums every queued item with a plain read-modify-write of its dw (`dw = dw + sum of the
            # partials`, no atomics).  Two items with the SAME dw in one launch would race and lose a contribution: a
            # weight used twice in a graph (the res5 head run twice under DADET_NO_ROI_DEDUP=1, two backward() calls in
            # front of one step()).  The earlier item is then reduced first, in stream order.
            if any(it[0].dw in _PENDING_DW for it in batch):
                flush_wgrad_reductions(batch[0][1].device)
            for it in batch:
                if it[0].dw in _PENDING_DW:            # twice inside one node's own batch
                    flush_wgrad_reductions(it[1].device)
                _PENDING_DW.add(it[0].dw)
                _PENDING_REDUCES.append(it)
            del batch[:]
            return
        if self.on:
            with torch.cuda.stream(self.lane):
                _C.conv_wgrad_reduce_batch(batch)
        else:
            _C.conv_wgrad_reduce_batch(batch)

    def join(self):
        if self.on and self.out:
            record(self.out, self.main)
            self.main.wait_stream(self.lane)


# Direct accumulation of weight gradients into the parameters' persistent .grad buffers (the flat buckets of
# parallel.reducer): removes, per backward node, the compute stream's wait for the lane and autograd's `grad += dw`
# launches.  Switched on by FusedSGD.attach_reducer — with a reducer every .grad is a zeroed persistent view and every
# consumer of the gradients (collectives, optimizer) first calls join_wgrad_lane().  DADET_DIRECT_WGRAD=0 disables it.
DIRECT_WGRAD = False


def enable_direct_wgrad(flag=True):
    global DIRECT_WGRAD
    DIRECT_WGRAD = bool(flag) and os.environ.get("DADET_DIRECT_WGRAD", "1") == "1"


def direct_grad_target(param):
    if not DIRECT_WGRAD or param is None or not param.requires_grad or not param.is_leaf:
        return None
    g = param.grad
    if g is None or not g.is_cuda or g.dtype != torch.float32 or g.stride() != param.stride():
        return None
    if g.dim() != 4 or not g.is_contiguous(memory_format=torch.channels_last) or g.data_ptr() % 16:
        return None      # the kernels write [Cout][KH][KW][Cin]
    return g


def direct_bias_target(param):
    """the persistent gradient slot of a bias (1-D leaf) when direct accumulation is on — `_C.colsum(..., out=slot,
    accumulate=True)` then leaves nothing for autograd to add (its post-accumulate hooks still fire)"""
    if not DIRECT_WGRAD or param is None or not param.requires_grad or not param.is_leaf:
        return None
    g = param.grad
    if g is None or not g.is_cuda or g.dtype != torch.float32 or g.dim() != 1 or not g.is_contiguous():
        return None
    return g


def bias_grad(param, g2d, cols=None):
    """bias gradient = column sums of g2d: straight into the bias's gradient slot when there is one (returns None for
    autograd), else a fresh tensor"""
    from .. import _C

    tgt = direct_bias_target(param)
    if tgt is not None and tgt.numel() == (g2d.shape[1] if cols is None else cols):
        _C.colsum(g2d, out=tgt, accumulate=True, cols=cols)
        return None
    return _C.colsum(g2d, cols=cols)


# reduction passes of split weight gradients whose results nobody has asked for yet (WgradLane.reduce_batch)
_PENDING_REDUCES = []
DEFER_WGRAD_REDUCE = True
_PENDING_DW = set()          # dw pointers of the queued items (see WgradLane.reduce_batch)
# (measured in round 3 and removed: the same passes on a stream of their own beside the backward GEMMs — 19.29 / 19.53 ms
# per step against 19.22 / 19.23: the HBM traffic slows the power-limited GEMMs by what it saves at the end)


def discard_wgrad_reductions():
    """drop the queued passes without running them: their gradient buffers are about to be zeroed (a backward that was
    not followed by step(); BucketedGradReducer.zero_grad).  Also releases the partial-sum workspaces they pin."""
    del _PENDING_REDUCES[:]
    _PENDING_DW.clear()


def flush_wgrad_reductions(device):
    """every deferred reduction pass is complete for the CURRENT stream after this (callers: join_wgrad_lane, i.e.
    everything that reads gradients; the gradient reducer before it issues a bucket's collective)"""
    if not _PENDING_REDUCES:
        return
    items = list(_PENDING_REDUCES)
    del _PENDING_REDUCES[:]
    _PENDING_DW.clear()
    from .. import _C

    if device.type == "cuda":
        cur = torch.cuda.current_stream(device)
        if lane_in_use():
            cur.wait_stream(side_stream(device, 2))       # partial sums computed on the lane
        for it in items:
            it[1].record_stream(cur)                       # the partial-sum workspaces are read on this stream
    _C.conv_wgrad_reduce_batch(items)


def join_wgrad_lane(device):
    """the current stream waits for every weight gradient queued on the lane (call before reading .grad)"""
    if device.type == "cuda" and lane_in_use():
        torch.cuda.current_stream(device).wait_stream(side_stream(device, 2))
    flush_wgrad_reductions(device)
