"""Data-gradient weights kept across steps.

Every data-gradient GEMM reads its layer's weights as [Cin][KH'][KW'][Cout] (taps flipped, FrozenBN scale folded in),
produced by a transpose kernel.  In the reference-order schedule that is one small launch in front of each of the 44
data-gradient GEMMs of the backward pass — on the critical chain.  The weights only change in the optimizer step: with
the fused optimizer (solver.fused_sgd.FusedSGD) the transposed copies live in persistent buffers that are refreshed
right after the SGD kernel, on a side stream, while the next step's forward pass (which does not read them) is already
running; the backward pass then finds them ready.  146 MB of extra HBM for R-50-C4.

Validity: an entry carries (weight._version, scale._version, optimizer epoch); any in-place torch update of the
weight (load_state_dict, another optimizer) changes the stamp and the copy is recomputed at its next use."""
import weakref

import torch

from .. import _C
from .streams import side_stream

ENABLED = False           # set by FusedSGD.step: from then on weights change only there (or bump their _version)
ALLOWED = __import__("os").environ.get("DADET_WT_CACHE", "1") == "1"   # A/B switch (tests flip it)
_ENTRIES = {}
_EPOCH = 0
_REFRESH = None           # event: the refresh of the current epoch has completed
_SYNCED = set()           # streams that already wait for it


class _Entry(object):
    __slots__ = ("w", "scale", "wt", "stamp")


def _stamp(w, scale):
    return (w._version, scale._version if scale is not None else -1, _EPOCH)


def transposed(w, scale=None):
    """_C.conv_weight_transpose(w, scale), from the cache when the fused optimizer maintains it"""
    if not (ENABLED and ALLOWED and w.is_cuda and w.is_leaf and w.requires_grad):
        return _C.conv_weight_transpose(w, scale)
    key = (id(w), id(scale) if scale is not None else 0)
    e = _ENTRIES.get(key)
    if e is None or e.w() is not w or (scale is not None and (e.scale is None or e.scale() is not scale)):
        e = _Entry()
        e.w = weakref.ref(w, lambda _r, k=key: _ENTRIES.pop(k, None))
        e.scale = weakref.ref(scale) if scale is not None else None
        e.wt = _C.conv_weight_transpose(w, scale)
        e.stamp = _stamp(w, scale)
        _ENTRIES[key] = e
        return e.wt
    if e.stamp != _stamp(w, scale):          # modified outside the optimizer, or never refreshed: recompute in place
        _C.conv_weight_transpose(w, scale, out=e.wt)
        e.stamp = _stamp(w, scale)
        return e.wt
    if _REFRESH is not None:
        cur = torch.cuda.current_stream(w.device)
        if cur.cuda_stream not in _SYNCED:
            cur.wait_event(_REFRESH)
            _SYNCED.add(cur.cuda_stream)
    return e.wt


def refresh_all(device):
    """called by the fused optimizer right after its SGD launch (current stream): recompute every cached copy on a
    side stream, behind that launch"""
    global ENABLED, _EPOCH, _REFRESH
    ENABLED = True
    if device.type != "cuda" or not ALLOWED:
        return
    _EPOCH += 1
    _SYNCED.clear()
    for k in [k for k, e in _ENTRIES.items() if e.w() is None or (e.scale is not None and e.scale() is None)]:
        del _ENTRIES[k]           # the weight or its FrozenBN scale tensor is gone
    live = list(_ENTRIES.values())
    if not live:
        _REFRESH = None
        return
    main = torch.cuda.current_stream(device)
    side = side_stream(device, 3)
    side.wait_event(main.record_event())
    with torch.cuda.stream(side):
        for e in live:
            w = e.w()
            scale = e.scale() if e.scale is not None else None
            _C.conv_weight_transpose(w, scale, out=e.wt)
            e.stamp = _stamp(w, scale)
        _REFRESH = side.record_event()


def clear():
    global ENABLED, _REFRESH
    _ENTRIES.clear()
    _SYNCED.clear()
    _REFRESH = None
    ENABLED = False
