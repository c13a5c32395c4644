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
* views of a tensor that carries a slot use it (`ptr`); anything else is measured (`dadet_amax`, one pass over t) the
  first time a GEMM asks;
* parameters and the cached transposed weights of the data-gradient GEMMs: slots in a persistent per-device array, all
  re-measured by ONE launch per weight epoch (`WeightSlots.refresh`, called where the transposed weights are refreshed:
  behind the optimizer step).

`MEASURED` counts the one-pass measurements (tools / tests read it to see which producers still lack a fused maximum).
"""
import ctypes
import weakref

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


def carry(dst, src):
    """max|dst| <= max|src| by the nature of the operation that made dst from src (a gate, a pooling, a gather, a view, a
    slice ...): dst takes over src's slot when src carries a valid one; otherwise nothing is attached and dst is measured
    when a GEMM first asks.  NOT for sums: |a + b| can exceed both."""
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
    # dadet_amax scans t.numel() floats from data_ptr(): right for dense storage in any dimension order (NCHW-contiguous,
    # channels_last), wrong for a column slice / strided view (it would scan other elements and could UNDER-estimate
    # max|t|, i.e. overflow fp16 in mode 4)
    if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        raise _lib.DadetError("amax.measure: a non-dense view (shape %s, strides %s) cannot be measured in place; make "
                              "it contiguous first" % (tuple(t.shape), tuple(t.stride())))
    slot = new_slot(t.device)
    _lib.call("dadet_amax", ctypes.c_void_p(t.data_ptr()), ctypes.c_longlong(t.numel()), ctypes.c_void_p(slot[0]),
              _stream())
    MEASURED += 1
    return attach(t, slot)


def ptr(t):
    """address of the slot of t as a ctypes pointer.  A view (reshape, slice, permute ...) of a tensor that carries a slot
    uses its base's — a subset of the values under the same bound, and view and base share one version counter; anything
    else without a valid slot is measured."""
    a = t.__dict__.get("_dadet_amax")
    if a is None or a[2] != t._version:
        base = t._base
        ab = base.__dict__.get("_dadet_amax") if base is not None else None
        if ab is not None and ab[2] == base._version:
            a = t._dadet_amax = (ab[0], ab[1], t._version)
        else:
            measure(t)
            a = t.__dict__["_dadet_amax"]
    return ctypes.c_void_p(a[0])


class WeightSlots(object):
    """Slots of tensors that persist across steps and change once per step: parameters (updated in place by the fused
    optimizer through raw pointers — no version bump, hence the explicit epoch) and the transposed-weight buffers of the
    data-gradient GEMMs.  Keyed by (address, numel).  An entry holds a WEAK reference to the tensor that owns the storage
    (the parameter itself for a view of one): when a model is gone its entries are swept and their slots reused."""

    CAP = 4096

    def __init__(self):
        self.by_dev = {}      # device index -> dict(slots=tensor, free=[...], entries={key: entry}, table=None, epoch=int)

    def _dev(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        d = self.by_dev.get(idx)
        if d is None:
            slots = torch.zeros((8, STRIDE), dtype=torch.float32, device=device)
            torch.cuda.current_stream(device).synchronize()
            d = self.by_dev[idx] = dict(slots=slots, free=list(range(self.CAP - 1, -1, -1)), entries={}, table=None,
                                        epoch=-1)
        return d

    @staticmethod
    def _covers(o, e):
        """does tensor o still own the recorded address range?  (`p.data = ...`, model.to() / .float() and re-flattening
        replace a parameter's storage while the Python object — and its version counter — live on)"""
        try:
            st = o.untyped_storage()
            base = st.data_ptr()
            return base <= e["ptr"] and e["ptr"] + 4 * e["n"] <= base + st.nbytes()
        except RuntimeError:
            return False

    @staticmethod
    def _alive(e):
        o = e["ref"]()
        return o is not None and o._version == e["version"] and WeightSlots._covers(o, e)

    def _sweep(self, d, epoch):
        """forget entries whose owner is gone, or that nobody asked about during the last two epochs"""
        dead = [k for k, e in d["entries"].items()
                if e["ref"]() is None or e["used"] < epoch - 2 or not self._covers(e["ref"](), e)]
        for k in dead:
            d["free"].append(d["entries"].pop(k)["i"])
        if dead:
            d["table"] = d["table_trained"] = None

    def _measure_one(self, d, e):
        # its own slot back to zero, then one pass: same stream, ordered
        d["slots"][:, e["i"]].zero_()
        _lib.call("dadet_amax", ctypes.c_void_p(e["ptr"]), ctypes.c_longlong(e["n"]),
                  ctypes.c_void_p(d["slots"].data_ptr() + 4 * e["i"]), _stream())

    def ptr(self, t, epoch):
        d = self._dev(t.device)
        key = (t.data_ptr(), t.numel())
        e = d["entries"].get(key)
        owner = t._base if t._base is not None else t
        if e is not None and e["ref"]() is not owner:        # the address went to another tensor
            d["free"].append(d["entries"].pop(key)["i"])
            d["table"] = d["table_trained"] = None
            e = None
        if e is None:
            if not d["free"]:
                self._sweep(d, epoch)
            if not d["free"]:
                raise _lib.DadetError("amax: more than %d persistent GEMM operands alive" % self.CAP)
            e = d["entries"][key] = dict(ref=weakref.ref(owner), ptr=key[0], n=key[1], i=d["free"].pop(),
                                         version=owner._version, epoch=epoch, used=epoch)
            d["table"] = d["table_trained"] = None
            self._measure_one(d, e)
        else:
            e["used"] = epoch
            if e["version"] != owner._version or e["epoch"] != epoch:
                if e["version"] == owner._version and d["epoch"] != epoch:
                    self.refresh(t.device, epoch, sync=True)
                if e["version"] != owner._version or e["epoch"] != epoch:
                    e["version"], e["epoch"] = owner._version, epoch
                    self._measure_one(d, e)
        return ctypes.c_void_p(d["slots"].data_ptr() + 4 * e["i"])

    def invalidate(self, t):
        """t was rewritten on the current stream: its slot is measured again at the next request"""
        d = self._dev(t.device)
        e = d["entries"].get((t.data_ptr(), t.numel()))
        if e is not None:
            e["epoch"] = -1

    @staticmethod
    def _frozen(e):
        """the entry's storage belongs to a parameter no optimizer changes (a frozen stage's weight)"""
        o = e["ref"]()
        return isinstance(o, torch.nn.Parameter) and not o.requires_grad

    def refresh(self, device, epoch, sync=False, trained_only=False):
        """every registered tensor of this device re-measured by one fill + one launch on the current stream.  sync: the
        call comes from the middle of a step (not from behind the optimizer): other streams may still read the slots.
        trained_only (the optimizer's call): frozen parameters keep their slots UNTOUCHED — their values did not change, and
        the next step's frozen prefix may be reading them on the compute stream while this runs on the optimizer lane
        (utils.streams: a slot is zero between the fill and the launch that re-measures it)."""
        if sync:
            torch.cuda.synchronize(device)
        d = self._dev(device)
        d["epoch"] = epoch
        self._sweep(d, epoch)
        live = [e for e in d["entries"].values() if self._alive(e)]
        if trained_only:
            for e in live:
                if self._frozen(e):
                    e["epoch"] = epoch
            live = [e for e in live if not self._frozen(e)]
        if not live:
            return
        keys = [(e["i"], e["ptr"], e["n"]) for e in live]
        which = "table_trained" if trained_only else "table"
        if d.get(which) is None or d[which][3] != keys:
            arr = (_lib.AmaxItem * len(live))()
            blocks = 0
            for k, e in enumerate(live):
                it = arr[k]
                n = e["n"]
                it.x, it.slot, it.n = e["ptr"], d["slots"].data_ptr() + 4 * e["i"], n
                it.first_block = blocks
                # sixteen float4 per thread (four rounds of four loads in flight): ~200 MB of weights and cached transposes
                # in ~3000 workgroups — at two float4 per thread the launch was bound by its 25 000 workgroups' dispatch
                # (149 us per step, 1.4 TB/s; profiles/r05_step_timeline_img_only.txt)
                it.blocks = max(1, min(512, (n // 4 + 4095) // 4096))
                blocks += it.blocks
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            idx = torch.tensor([e["i"] for e in live], dtype=torch.int64).to(device) if trained_only else None
            d[which] = (host.to(device), blocks, len(live), keys, idx)
        dev_t, blocks, n, _, idx = d[which]
        if idx is None:
            d["slots"][:, :self.CAP].zero_()
        else:
            d["slots"].index_fill_(1, idx, 0.0)
        _lib.call("dadet_amax_batch", ctypes.c_void_p(dev_t.data_ptr()), n, blocks, _stream())
        for e in live:
            e["epoch"] = epoch


WEIGHTS = WeightSlots()
