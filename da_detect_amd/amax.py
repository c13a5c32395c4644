"""Largest magnitudes of GEMM operands for contraction mode 4 (two fp16 terms under a per-tensor power-of-two scale,
`csrc/conv_common.h`).

A *slot* holds max|t| (or an upper bound within a few binades) of a tensor t: eight device floats, `STRIDE` apart, whose
maximum is the value (writers merge into the shard of their XCD; include/dadet.h).  Slots are columns of zero-filled
[8][STRIDE] arrays.  The GEMM kernels read the operands' slots with scalar loads and derive the scales themselves; nothing
here ever brings a maximum to the host.  Where slots come from:

* the output of a GEMM: the kernel's own epilogue merges max|y| into a fresh slot (`dadet_conv_forward_scaled`), which
  `_C.conv_forward` attaches to the result as the Python attribute `_dadet_amax`.  The attribute belongs to the tensor
  OBJECT: a view, a slice or the result of any ATen operation is a new object without it, and an in-place ATen operation
  bumps `_version`, which invalidates it — a stale maximum cannot be picked up by construction;
* tensors written by other kernels of this library: `carry(dst, src)` where max|dst| <= max|src| by the operation's
  nature (max-pooling, ROIAlign, average pooling, gathers, ReLU gates) — a bound, not a measurement;
* anything else: `ptr(t)` measures it (`dadet_amax`, one pass over t) the first time a GEMM asks;
* parameters and the cached transposed weights of the data-gradient GEMMs: slots in a persistent per-device array, all
  re-measured by ONE launch per weight epoch (`WeightSlots.refresh`, called where the transposed weights are refreshed:
  behind the optimizer step).

`MEASURED` counts the one-pass measurements (tools / tests read it to see which producers still lack a fused maximum).
"""
import ctypes

import torch

from . import _lib

STRIDE = 1 << 16     # DADET_AMAX_STRIDE
_POOL_SLOTS = STRIDE
_pools = {}          # device index -> [tensor, next]
MEASURED = 0         # dadet_amax launches issued by ptr() for activations (not weights)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def new_slot(device):
    """(address, owner) of a zero-initialised slot nothing else uses; `owner` keeps its memory alive"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _pools.get(idx)
    if pool is None or pool[1] >= _POOL_SLOTS:
        t = torch.zeros((8, STRIDE), dtype=torch.float32, device=device)
        # the fill is ordered on THIS stream only, and slots are handed to kernels on any stream
        torch.cuda.current_stream(device).synchronize()
        pool = _pools[idx] = [t, 0]
    i = pool[1]
    pool[1] = i + 1
    return pool[0].data_ptr() + 4 * i, pool[0]


def attach(t, slot):
    """`slot` (address, owner) holds max|t| from now on — valid while t's version stays what it is"""
    t._dadet_amax = (slot[0], slot[1], t._version)
    return t


def carry(dst, src, *more):
    """max|dst| <= max over the sources: dst takes over a source's slot.  With several sources (an addition's operands do
    NOT qualify: |a + b| can exceed both) the caller guarantees the bound; only the first VALID source's slot is used when
    there is one source, otherwise nothing is attached and dst is measured on demand."""
    if more:
        return dst
    a = src.__dict__.get("_dadet_amax") if src is not None else None
    if a is not None and a[2] == src._version:
        dst._dadet_amax = (a[0], a[1], dst._version)
    return dst


def slot_of(t):
    """the valid slot attached to t, or None"""
    a = t.__dict__.get("_dadet_amax")
    if a is not None and a[2] == t._version:
        return a[0], a[1]
    return None


def value(t):
    """the number in t's slot, brought to the host (tests and probes only: a synchronising read)"""
    s = slot_of(t)
    if s is None:
        return None
    return float(s[1].view(8, STRIDE)[:, (s[0] - s[1].data_ptr()) // 4].max())


def measure(t):
    """one pass over t (any shape, dense storage of t.numel() floats) into a fresh slot, attached to t"""
    global MEASURED
    slot = new_slot(t.device)
    _lib.call("dadet_amax", ctypes.c_void_p(t.data_ptr()), ctypes.c_longlong(t.numel()), ctypes.c_void_p(slot[0]),
              _stream())
    MEASURED += 1
    return attach(t, slot)


def ptr(t):
    """address of the slot of t as a ctypes pointer; measures t when it carries none"""
    a = t.__dict__.get("_dadet_amax")
    if a is None or a[2] != t._version:
        measure(t)
        a = t.__dict__["_dadet_amax"]
    return ctypes.c_void_p(a[0])


class WeightSlots(object):
    """Slots of tensors that persist across steps and change once per step: parameters (updated in place by the fused
    optimizer through raw pointers — no version bump, hence the explicit epoch) and the transposed-weight buffers of the
    data-gradient GEMMs.  Keyed by (address, numel); an entry keeps its tensor alive."""

    CAP = 4096

    def __init__(self):
        self.by_dev = {}      # device index -> dict(slots=tensor, n=int, entries={key: entry}, table=None, epoch=int)

    def _dev(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        d = self.by_dev.get(idx)
        if d is None:
            slots = torch.zeros((8, STRIDE), dtype=torch.float32, device=device)
            torch.cuda.current_stream(device).synchronize()
            d = self.by_dev[idx] = dict(slots=slots, n=0, entries={}, table=None, epoch=-1)
        return d

    def _measure_one(self, d, e):
        # its own slot back to zero, then one pass: same stream, ordered
        d["slots"][:, e["i"]].zero_()
        _lib.call("dadet_amax", ctypes.c_void_p(e["t"].data_ptr()), ctypes.c_longlong(e["t"].numel()),
                  ctypes.c_void_p(d["slots"].data_ptr() + 4 * e["i"]), _stream())

    def ptr(self, t, epoch):
        d = self._dev(t.device)
        key = (t.data_ptr(), t.numel())
        e = d["entries"].get(key)
        if e is None:
            if d["n"] >= self.CAP:
                raise _lib.DadetError("amax: more than %d persistent GEMM operands registered" % self.CAP)
            e = d["entries"][key] = dict(t=t, i=d["n"], version=t._version, epoch=epoch, used=epoch)
            d["n"] += 1
            d["table"] = None
            self._measure_one(d, e)
        else:
            e["used"] = epoch
            if e["version"] != t._version or e["epoch"] != epoch:
                if e["version"] == t._version and d["epoch"] != epoch:
                    self.refresh(t.device, epoch, sync=True)
                if e["version"] != t._version or e["epoch"] != epoch:
                    e["t"], e["version"], e["epoch"] = t, t._version, epoch
                    self._measure_one(d, e)
        return ctypes.c_void_p(d["slots"].data_ptr() + 4 * e["i"])

    def invalidate(self, t):
        """t was rewritten on the current stream: its slot is measured again at the next request"""
        d = self._dev(t.device)
        e = d["entries"].get((t.data_ptr(), t.numel()))
        if e is not None:
            e["epoch"] = -1

    def refresh(self, device, epoch, sync=False):
        """every registered tensor of this device re-measured by one fill + one launch on the current stream.  sync: the
        call comes from the middle of a step (not from behind the optimizer): other streams may still read the slots"""
        if sync:
            torch.cuda.synchronize(device)
        d = self._dev(device)
        d["epoch"] = epoch
        # tensors nobody asked about during the last two epochs belong to a model that is gone
        stale = [k for k, e in d["entries"].items() if e["used"] < epoch - 2]
        if stale:
            # their slot indices are not reused (the array is large); just forget them
            for k in stale:
                del d["entries"][k]
            d["table"] = None
        live = [e for e in d["entries"].values() if e["version"] == e["t"]._version]
        if not live:
            return
        keys = [e["i"] for e in live]
        if d["table"] is None or d["table"][3] != keys:
            arr = (_lib.AmaxItem * len(live))()
            blocks = 0
            for k, e in enumerate(live):
                it = arr[k]
                n = e["t"].numel()
                it.x, it.slot, it.n = e["t"].data_ptr(), d["slots"].data_ptr() + 4 * e["i"], n
                it.first_block = blocks
                it.blocks = max(1, min(64, (n // 4 + 1023) // 1024))
                blocks += it.blocks
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            d["table"] = (host.to(device), blocks, len(live), keys)
        d["slots"][:, :d["n"]].zero_()
        dev_t, blocks, n, _ = d["table"]
        _lib.call("dadet_amax_batch", ctypes.c_void_p(dev_t.data_ptr()), n, blocks, _stream())
        for e in live:
            e["epoch"] = epoch


WEIGHTS = WeightSlots()
