"""dadet_topk_sorted against torch.sort(descending=True, stable=True) — the ranking rule of RPNPostProcessor
(reference rpn/inference.py:93-95: objectness.topk(pre_nms_top_n, dim=1, sorted=True)); indices must be IDENTICAL."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(scores, k):
    from da_detect_amd import _C

    vals, idx = _C.topk_sorted(scores, k)
    want_v, want_i = torch.sort(scores, dim=1, descending=True, stable=True)
    assert idx.dtype == torch.int64 and torch.equal(idx, want_i[:, :k]), "indices differ from the stable sort"
    assert torch.equal(vals, want_v[:, :k])


@pytest.mark.parametrize("n,k", [(122880, 12000), (122880, 6000), (30720, 12000), (12000, 12000), (11520, 2000),
                                 (5, 3), (1, 1), (16384, 16384), (40000, 16384), (2049, 2048)])
def test_topk_matches_stable_sort(device, n, k):
    g = torch.Generator().manual_seed(n + k)
    logits = torch.randn((2, n), generator=g) * 3 - 3
    _check(logits.sigmoid().to(device), k)            # what the RPN feeds it
    _check((torch.randn((3, n), generator=g) * 100).to(device), k)     # negative values, wide range


def test_topk_ties(device):
    g = torch.Generator().manual_seed(0)
    n, k = 122880, 12000
    # heavy ties everywhere (64 distinct values): the threshold bucket holds ~1900 equal scores -> tie list path
    coarse = (torch.randint(0, 64, (2, n), generator=g).float() / 64).to(device)
    _check(coarse, k)
    # saturated scores: 40 000 exact ones -> more ties than the tie list holds -> ordered-compaction path
    sat = torch.rand((2, n), generator=g)
    sat[:, torch.randperm(n, generator=g)[:40000]] = 1.0
    _check(sat.to(device), k)
    # all equal: the first k indices
    _check(torch.full((1, n), 0.25, device=device), k)
    # threshold tie straddling the boundary with few ties
    few = torch.rand((1, 5000), generator=g)
    few[0, [10, 4000, 77, 3000]] = few[0].sort(descending=True)[0][999]
    _check(few.to(device), 1000)
    _check(few.to(device), 1001)


def test_rpn_selection_uses_the_kernel_and_matches_the_sort_path(device, monkeypatch):
    """forward_for_single_feature_map with the kernel == with torch.sort (same proposals, same order)"""
    from da_detect_amd.modeling.rpn import inference
    from da_detect_amd.modeling.rpn.anchor_generator import make_anchor_generator
    from da_detect_amd.structures.image_list import to_image_list
    from golden.cases import case_cfg

    c = case_cfg("da_plain")
    g = torch.Generator().manual_seed(3)
    H, W = 512, 1024
    obj = (torch.randn((2, 15, H // 16, W // 16), generator=g) * 2 - 3).to(device)
    reg = (torch.randn((2, 60, H // 16, W // 16), generator=g) * 0.2).to(device)
    images = to_image_list([torch.zeros(3, H, W), torch.zeros(3, H, W)], 32).to(device)
    anchors = make_anchor_generator(c).to(device)(images, [obj])
    sel = inference.make_rpn_postprocessor(c, None, is_train=True)
    out = []
    for flag in (True, False):
        monkeypatch.setattr(inference, "_TOPK_KERNEL", flag)
        out.append(sel.forward_for_single_feature_map([a[0] for a in anchors], obj, reg))
    for a, b in zip(*out):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("objectness"), b.get_field("objectness"))
