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


def _check_rows(rows, ks):
    from da_detect_amd import _C

    got = _C.topk_sorted_rows(rows, ks)
    for r, k, (vals, idx) in zip(rows, ks, got):
        want_v, want_i = torch.sort(r, descending=True, stable=True)
        assert idx.dtype == torch.int64 and torch.equal(idx, want_i[:k]), "row of %d scores, k = %d: indices differ" % (r.numel(), k)
        assert torch.equal(vals, want_v[:k])


def test_topk_rows_pyramid_shapes_match_stable_sort(device):
    """dadet_topk_sorted_rows on the ten rows of a five-level pyramid at 1024 x 2048 (3 anchors per cell, two images,
    pre_nms_top_n = 2000) — and the same call twice (the workspace is re-zeroed per call)"""
    g = torch.Generator().manual_seed(5)
    lens = [3 * 256 * 512, 3 * 128 * 256, 3 * 64 * 128, 3 * 32 * 64, 3 * 16 * 32]
    rows = [(torch.randn(n, generator=g) * 3 - 3).sigmoid().to(device) for n in lens for _ in range(2)]
    ks = [min(2000, n) for n in lens for _ in range(2)]
    assert ks[-1] == 1536            # the coarsest level: the whole row is taken (k == n)
    _check_rows(rows, ks)
    _check_rows(rows, ks)
    # other k, a single row, the largest k
    _check_rows(rows[:3], [12000, 700, 16384])
    _check_rows([rows[0]], [1])


def test_topk_rows_ties(device):
    g = torch.Generator().manual_seed(9)
    n = 98304
    coarse = (torch.randint(0, 64, (n,), generator=g).float() / 64).to(device)           # ~1500 ties at the threshold
    sat = torch.rand(n, generator=g)
    sat[torch.randperm(n, generator=g)[:30000]] = 1.0                                     # > tie list: one-workgroup redo
    flat = torch.full((n,), 0.25)
    few = torch.rand(5000, generator=g)
    few[[10, 4000, 77, 3000]] = few.sort(descending=True)[0][999]
    _check_rows([coarse, sat.to(device), flat.to(device), few.to(device), few.to(device)], [2000, 2000, 2000, 1000, 1001])
    neg = (torch.randn(40000, generator=g) * 100).to(device)                             # negative values, wide range
    _check_rows([neg, coarse], [16384, 12000])


def test_fpn_selection_with_the_batched_ranking_matches_the_per_level_sorts(device, monkeypatch):
    """_device_selection with one dadet_topk_sorted_rows call for all (level, image) rows == with one library sort per
    level: same boxes, scores, keep lists and counts"""
    from da_detect_amd.modeling.rpn import inference
    from da_detect_amd.modeling.rpn.anchor_generator import make_anchor_generator
    from da_detect_amd.structures.image_list import to_image_list
    from golden.cases import fpn_dcn_da_cfg

    c = fpn_dcn_da_cfg()
    g = torch.Generator().manual_seed(4)
    H, W = 512, 768
    strides = (4, 8, 16, 32, 64)
    A = len(c.MODEL.RPN.ASPECT_RATIOS)
    obj = [(torch.randn((2, A, H // s, W // s), generator=g) * 2 - 3).to(device) for s in strides]
    reg = [(torch.randn((2, 4 * A, H // s, W // s), generator=g) * 0.2).to(device) for s in strides]
    images = to_image_list([torch.zeros(3, H, W), torch.zeros(3, H, W)], 32).to(device)
    anchors = make_anchor_generator(c).to(device)(images, obj)
    sel = inference.make_rpn_postprocessor(c, None, is_train=True)
    out = []
    for flag in (True, False):
        monkeypatch.setattr(inference, "_ROWS_TOPK", flag)
        out.append(sel._device_selection(anchors, obj, reg))
    torch.cuda.synchronize()
    for a, b in zip(*out):
        assert int(a[3]) == int(b[3]) > 100 and a[4] == b[4]
        n = int(a[3])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2][:n], b[2][:n])
