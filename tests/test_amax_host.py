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
