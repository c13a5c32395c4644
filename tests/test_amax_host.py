"""Host-side bookkeeping of the largest magnitudes that contraction mode 4 needs (da_detect_amd/amax.py): which tensor
OBJECT carries a valid slot, when it stops being valid, what a view inherits.  No kernel runs here: `measure` is replaced by
a recorder (the GPU side is covered by tests/test_ops_gpu.py::test_fp16_split_* and the model-level parity tests)."""
import torch

from da_detect_amd import amax


class _Owner(object):
    """stands for the pool tensor that keeps a slot's memory alive"""


def _recording_measure(monkeypatch):
    calls = []

    def measure(t):
        calls.append(t)
        return amax.attach(t, (0x1000 + 4 * len(calls), _Owner()))

    monkeypatch.setattr(amax, "measure", measure)
    return calls


def test_a_slot_belongs_to_the_object_and_dies_with_an_in_place_write(monkeypatch):
    calls = _recording_measure(monkeypatch)
    t = torch.ones(4, 8)
    amax.attach(t, (0x2000, _Owner()))
    assert amax.slot_of(t)[0] == 0x2000 and amax.ptr(t).value == 0x2000 and not calls
    u = t + 0                                   # a new object: nothing attached, measured on demand
    assert amax.slot_of(u) is None
    assert amax.ptr(u).value == 0x1004 and calls == [u]
    t.mul_(3.0)                                 # an in-place write through ATen bumps the version: the bound is void
    assert amax.slot_of(t) is None
    amax.ptr(t)
    assert len(calls) == 2 and calls[1] is t


def test_views_use_their_base_and_share_its_version(monkeypatch):
    calls = _recording_measure(monkeypatch)
    t = torch.ones(6, 8)
    amax.attach(t, (0x3000, _Owner()))
    v = t.view(3, 16)[1:]                       # a view of a view: `_base` is the root
    assert amax.ptr(v).value == 0x3000 and not calls
    assert amax.slot_of(v)[0] == 0x3000         # cached on the view
    v.add_(1.0)                                 # writes through the view void both
    assert amax.slot_of(t) is None and amax.slot_of(v) is None
    d = t.detach()                              # an alias, not a view: no base to inherit from
    assert d._base is None
    amax.ptr(d)
    assert calls == [d]


def test_carry_hands_a_valid_bound_on_and_nothing_else(monkeypatch):
    _recording_measure(monkeypatch)
    src, dst = torch.ones(4), torch.zeros(2)
    assert amax.slot_of(amax.carry(dst, src)) is None          # the source carries nothing
    amax.attach(src, (0x4000, _Owner()))
    assert amax.slot_of(amax.carry(dst, src))[0] == 0x4000
    src.add_(1.0)
    other = torch.zeros(2)
    assert amax.slot_of(amax.carry(other, src)) is None        # stale source
    assert amax.slot_of(dst)[0] == 0x4000                      # what was handed on before stays (dst was not written)


class _FakeLib(object):
    """stands in for the library's `call`: dadet_amax / dadet_amax_batch computed with torch on CPU tensors registered by
    address, written into the slot array the way the kernels do (shard 0 of the eight)"""

    def __init__(self, slots_of):
        self.tensors, self.calls, self.slots_of = {}, [], slots_of

    def know(self, *ts):
        for t in ts:
            self.tensors[t.data_ptr()] = t
        return ts[0] if len(ts) == 1 else ts

    def _merge(self, ptr, n, slot_addr):
        slots = self.slots_of()
        i = (slot_addr - slots.data_ptr()) // 4
        v = float(self.tensors[ptr].detach().reshape(-1)[:n].abs().max())
        slots[0, i] = max(float(slots[0, i]), v)

    def __call__(self, name, *args):
        self.calls.append(name)
        val = lambda a: a.value if hasattr(a, "value") else a    # noqa: E731
        if name == "dadet_amax":
            self._merge(val(args[0]), val(args[1]), val(args[2]))
        elif name == "dadet_amax_batch":
            import ctypes
            from da_detect_amd import _lib
            table = self.table_host
            items = (_lib.AmaxItem * val(args[1])).from_buffer_copy(bytes(table.numpy().tobytes()))
            for it in items:
                self._merge(it.x, it.n, it.slot)


def _weight_slots_on_cpu(monkeypatch):
    import types

    ws = amax.WeightSlots()
    fake = _FakeLib(lambda: ws.by_dev[0]["slots"])
    monkeypatch.setattr(amax._lib, "call", fake)
    monkeypatch.setattr(amax, "_stream", lambda: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: types.SimpleNamespace(synchronize=lambda: None))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda device=None: None)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    # the device table of the batched launch stays on the host here: remember it for the fake kernel
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        if self.dtype == torch.uint8 and self.dim() == 1:
            fake.table_host = self
        return orig_to(self, *a, **k)

    monkeypatch.setattr(torch.Tensor, "to", to)
    return ws, fake


def _held(ws, t):
    d = ws.by_dev[0]
    e = d["entries"][(t.data_ptr(), t.numel())]
    return float(d["slots"][:, e["i"]].max())


def test_weight_slots_follow_epochs_versions_and_owners(monkeypatch):
    """the persistent table of parameters' maxima (amax.WeightSlots): measured at registration, re-measured by ONE batched
    launch per weight epoch (raw in-place updates bump no version), singly after a versioned write, swept when the owner
    is gone, and never confused by an address that went to another tensor"""
    ws, fake = _weight_slots_on_cpu(monkeypatch)
    dev = torch.device("cpu")
    w1, w2 = fake.know(torch.nn.Parameter(torch.full((8, 4), 2.0)), torch.nn.Parameter(torch.full((16,), 0.5)))
    view = w1.view(4, 8)                                    # a view shares the parameter's address, size and entry
    ws.ptr(w1, 0), ws.ptr(w2, 0), ws.ptr(view, 0)
    assert fake.calls.count("dadet_amax") == 2 and (_held(ws, w1), _held(ws, w2)) == (2.0, 0.5)
    w1.data.mul_(3.0)                                       # the fused optimizer's kind of update: no version bump
    ws.ptr(w1, 0)
    assert _held(ws, w1) == 2.0                             # same epoch: still the recorded value (hence bump_weight_epoch)
    ws.refresh(dev, 1)                                      # the epoch bump: one batched launch for everything alive
    assert fake.calls.count("dadet_amax_batch") == 1 and (_held(ws, w1), _held(ws, w2)) == (6.0, 0.5)
    n = len(fake.calls)
    ws.ptr(w1, 1), ws.ptr(w2, 1)
    assert len(fake.calls) == n                             # valid for the epoch: nothing launched
    with torch.no_grad():
        w2.mul_(8.0)                                        # a versioned write: that one entry alone is measured again
    ws.ptr(w2, 1)
    assert fake.calls[n:] == ["dadet_amax"] and _held(ws, w2) == 4.0
    # the owner goes away: the entry is swept at the next refresh and its slot index is reused
    i2 = ws.by_dev[0]["entries"][(w2.data_ptr(), w2.numel())]["i"]
    del fake.tensors[w2.data_ptr()], w2
    ws.refresh(dev, 2)
    assert len(ws.by_dev[0]["entries"]) == 1 and i2 in ws.by_dev[0]["free"]
    # lazily, without a bump-time refresh: the first request of a new epoch refreshes everything
    w1.data.fill_(1.25)
    ws.ptr(w1, 3)
    assert _held(ws, w1) == 1.25


def test_weight_slots_drop_an_entry_whose_storage_was_replaced(monkeypatch):
    """`p.data = ...` (model.to(), re-flattening) gives the parameter OBJECT new storage: object and version live on, the
    recorded address is freed memory — the entry must not be measured again through it"""
    ws, fake = _weight_slots_on_cpu(monkeypatch)
    dev = torch.device("cpu")
    w = fake.know(torch.nn.Parameter(torch.full((8, 4), 2.0)))
    ws.ptr(w, 0)
    old_key = (w.data_ptr(), w.numel())
    assert old_key in ws.by_dev[0]["entries"]
    keep = w.data                                           # (keeps the old allocation alive so that its address is not reused)
    w.data = torch.full((8, 4), 5.0)
    fake.know(w)
    e = ws.by_dev[0]["entries"][old_key]
    assert not ws._alive(e), "the entry still claims storage the parameter no longer owns"
    n = len(fake.calls)
    ws.refresh(dev, 1)                                      # swept, not measured through the stale pointer
    assert old_key not in ws.by_dev[0]["entries"]
    assert "dadet_amax_batch" not in fake.calls[n:]
    ws.ptr(w, 1)
    assert _held(ws, w) == 5.0
    del keep


def test_measure_refuses_a_non_dense_view():
    """dadet_amax scans numel() floats from data_ptr(): a column slice would be measured over the wrong elements"""
    import pytest

    from da_detect_amd import _lib

    t = torch.zeros(4, 8)
    with pytest.raises(_lib.DadetError):
        amax.measure(t[:, :3])


def test_the_lane_resolves_operand_maxima_before_it_switches_streams(monkeypatch):
    """WgradLane.run / run_into (utils/streams.py): in mode 4 an operand without a slot is measured BEFORE the work moves to
    the lane stream, so the slot every later GEMM on the compute stream reads was written in that stream's order"""
    from da_detect_amd import _C
    from da_detect_amd.utils import streams

    calls = _recording_measure(monkeypatch)
    monkeypatch.setattr(_C, "_mode4", lambda: True)
    x, gy = torch.zeros(2, 8, 4, 4).contiguous(memory_format=torch.channels_last), torch.zeros(2, 8, 4, 4)
    has_slot = amax.attach(torch.zeros(3), (0x2000, _Owner()))
    strided = torch.zeros(4, 8)[:, :3]
    streams.WgradLane._resolve_maxima((x, gy, has_slot, strided, None, 3))
    assert [id(c) for c in calls] == [id(x), id(gy)]        # dense operands without a slot, each once; nothing else
    streams.WgradLane._resolve_maxima((x, gy))
    assert len(calls) == 2                                  # both carry valid slots now
