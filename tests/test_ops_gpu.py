"""Parity of the HIP operators (through the C-ABI) against the CPU oracle / a torch fp32 reference.

Bars: integer / index results bit-exact; ROIAlign forward bit-exact (same operation order, contraction off);
atomically-accumulated and MFMA results within the fp32 tolerances stated per test.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CL = torch.channels_last


def _rand_boxes(rng, n, W=2048, H=1024, max_side=400):
    xy = np.stack([rng.uniform(0, W - 2, n), rng.uniform(0, H - 2, n)], 1)
    wh = np.stack([rng.uniform(1, max_side, n), rng.uniform(1, max_side, n)], 1)
    b = np.concatenate([xy, np.minimum(xy + wh, [W - 1, H - 1])], 1).astype(np.float32)
    return b


# ------------------------------------------------------------------------------------------- NMS
@pytest.mark.parametrize("n,thr", [(1, 0.5), (5, 0.5), (63, 0.3), (64, 0.7), (65, 0.7), (1000, 0.3),
                                   (6000, 0.7), (12000, 0.7), (20000, 0.5)])
@pytest.mark.parametrize("tie_rule", [0, 1])
def test_nms_matches_oracle(device, n, thr, tie_rule):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(n * 7 + tie_rule)
    boxes = _rand_boxes(rng, n)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    # duplicate scores to exercise the (score desc, index asc) tie order
    if n > 10:
        scores[rng.integers(0, n, n // 10)] = scores[0]
    want = O.nms(boxes, scores, thr, tie_rule)
    keep, count = _C.nms_with_count(torch.from_numpy(boxes).to(device), torch.from_numpy(scores).to(device),
                                    thr, tie_rule=tie_rule)
    got = keep[: int(count.item())].cpu().numpy()
    assert got.dtype == np.int64
    assert np.array_equal(got, want)


def test_nms_exact_threshold_tie_rules(device):
    """IoU exactly 0.5: suppressed under the CPU rule (>=), kept under the CUDA rule (>)."""
    from da_detect_amd import _C

    boxes = torch.tensor([[0, 0, 9, 9], [0, 0, 9, 19]], dtype=torch.float32, device=device)  # 100 / 200
    scores = torch.tensor([0.9, 0.8], device=device)
    k0, c0 = _C.nms_with_count(boxes, scores, 0.5, tie_rule=0)
    k1, c1 = _C.nms_with_count(boxes, scores, 0.5, tie_rule=1)
    assert k0[: int(c0)].tolist() == [0]
    assert k1[: int(c1)].tolist() == [0, 1]


def test_nms_max_keep_on_sorted_input(device):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(3)
    n = 12000
    boxes = _rand_boxes(rng, n, max_side=150)
    scores = np.sort(rng.uniform(0, 1, n).astype(np.float32))[::-1].copy()
    want = O.nms(boxes, scores, 0.7, 0)[:2000]
    keep, count = _C.nms_with_count(torch.from_numpy(boxes).to(device), torch.from_numpy(scores).to(device),
                                    0.7, max_keep=2000, tie_rule=0)
    got = keep[: int(count.item())].cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,max_side,max_keep", [(12000, 150, 2000), (12000, 150, -1), (12000, 600, 2000), (5001, 80, 700),
                                                 (16384, 60, -1), (300, 200, 50), (257, 400, -1), (70, 300, 3)])
@pytest.mark.parametrize("tie_rule", [0, 1])
def test_nms_three_sweeps_agree_with_the_oracle(device, monkeypatch, n, max_side, max_keep, tie_rule):
    """DADET_NMS_SWEEP = 0 (plain), 1 (chunk-pipelined, round 3), default (256-box blocks resolved by four waves, round 6):
    the same kept indices, equal to the CPU oracle's (nms_cpu.cpp:6-75 restated) — crowded and sparse box sets (long and
    short suppression chains inside a block), counts that are not multiples of 64 / 256, quotas that end inside a block, an
    odd number of blocks, the 16 384-box limit of the one-workgroup sweeps"""
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(n + 3 * tie_rule + max_side)
    boxes = _rand_boxes(rng, n, max_side=max_side)
    if n > 1000:      # a crowded region: many boxes of one block suppress each other in chains
        boxes[: n // 3, :2] = boxes[: n // 3, :2] % 300
        boxes[: n // 3, 2:] = boxes[: n // 3, :2] + rng.uniform(20, 90, (n // 3, 2)).astype(np.float32)
    scores = np.sort(rng.uniform(0, 1, n).astype(np.float32))[::-1].copy()
    want = O.nms(boxes, scores, 0.7, tie_rule)
    if max_keep > 0:
        want = want[:max_keep]
    b, sc = torch.from_numpy(boxes).to(device), torch.from_numpy(scores).to(device)
    for kind in ("0", "1", "2"):
        monkeypatch.setenv("DADET_NMS_SWEEP", kind)
        for presorted in (False, True):
            keep, count = _C.nms_with_count(b, None if presorted else sc, 0.7, max_keep=max_keep, tie_rule=tie_rule)
            got = keep[: int(count.item())].cpu().numpy()
            assert np.array_equal(got, want), "sweep %s (presorted %s): %d kept, oracle %d" % (kind, presorted, len(got), len(want))


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_nms_near_threshold_pairs_and_degenerate_boxes_decide_as_the_oracle(device, tie_rule):
    """the mask kernel compares inter with thresh * union and divides only inside a band of 2^-19 around the threshold
    (csrc/nms.hip: iou_suppresses): pairs whose IoU lies within a few ulps of 0.7 — found by search, one pair per well
    separated cell so that each decision shows in the kept set — and degenerate boxes (zero / negative extent: the
    reference's quotient is then 0, negative or NaN) must come out as from the reference's division"""
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(5 + tie_rule)
    thr = np.float32(0.7)
    pairs = []
    # a box [0, 0, w, h] against itself shifted by dx: IoU sweeps through 0.7 as dx grows; keep the shifts that land near it
    while len(pairs) < 600:
        w, h = rng.uniform(20, 120, 2).astype(np.float32)
        a = np.array([0, 0, w, h], np.float32)
        dx0 = (w + 1) * 0.3 / 1.7
        for dx in (np.float32(dx0) + np.arange(-40, 41, dtype=np.float32) * np.float32(2.0 ** -14) * np.float32(w)):
            b = a + np.array([dx, 0, dx, 0], np.float32)
            inter = np.float32(max(np.float32(0), min(a[2], b[2]) - max(a[0], b[0]) + np.float32(1))) * np.float32(h + 1)
            aa = np.float32(a[2] - a[0] + 1) * np.float32(a[3] - a[1] + 1)
            ab = np.float32(b[2] - b[0] + 1) * np.float32(b[3] - b[1] + 1)
            ovr = inter / np.float32(np.float32(aa + ab) - inter)
            if abs(float(ovr) - 0.7) < 3e-6:
                pairs.append((a.copy(), b.copy()))
    boxes = []
    for i, (a, b) in enumerate(pairs):
        ox, oy = np.float32(300 * (i % 40)), np.float32(200 * (i // 40))
        boxes += [a + [ox, oy, ox, oy], b + [ox, oy, ox, oy]]
    boxes = np.asarray(boxes, np.float32)
    # degenerate boxes far away from everything else, overlapping each other
    deg = np.array([[20000, 20000, 19990, 20010], [20000, 20000, 19995, 20005], [20001, 20001, 20001, 20001],
                    [20001, 20001, 20001, 20001], [20000, 20005, 20010, 20000]], np.float32)
    boxes = np.concatenate([boxes, deg])
    scores = np.linspace(1, 0, len(boxes)).astype(np.float32)
    want = O.nms(boxes, scores, float(thr), tie_rule)
    kept_seconds = sum(1 for k in want if k % 2 == 1 and k < 2 * len(pairs))
    assert 50 < kept_seconds < len(pairs) - 50, "the pairs do not straddle the threshold (%d of %d kept)" % (kept_seconds, len(pairs))
    keep, count = _C.nms_with_count(torch.from_numpy(boxes).to(device), torch.from_numpy(scores).to(device), float(thr),
                                    tie_rule=tie_rule)
    assert np.array_equal(keep[: int(count)].cpu().numpy(), want)


def test_nms_empty(device):
    from da_detect_amd import _C

    out = _C.nms(torch.zeros((0, 4), device=device), torch.zeros((0,), device=device), 0.5)
    assert out.numel() == 0 and out.dtype == torch.int64


# -------------------------------------------------------------------------------------- ROIAlign
def _rois(rng, R, B, W=2048, H=1024):
    b = _rand_boxes(rng, R, W, H, max_side=900)
    # edge cases: degenerate, out-of-image, whole image
    b[0] = [10, 10, 10, 10]
    if R > 3:
        b[1] = [-50, -30, 40, 60]
        b[2] = [0, 0, W - 1, H - 1]
        b[3] = [W - 5, H - 5, W + 40, H + 40]
    idx = rng.integers(0, B, (R, 1)).astype(np.float32)
    return np.concatenate([idx, b], 1).astype(np.float32)


@pytest.mark.parametrize("C,H,W,R,ph,sr", [(8, 20, 32, 16, 7, 0), (64, 38, 76, 40, 14, 0),
                                            (256, 64, 128, 64, 14, 0), (12, 25, 31, 9, 7, 2), (3, 16, 16, 5, 2, 0)])
def test_roi_align_forward_bit_exact(device, C, H, W, R, ph, sr):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(C + R)
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    rois = _rois(rng, R, 2, W * 16, H * 16)
    want = O.roi_align_forward(x, rois, 1 / 16.0, ph, ph, sr)
    got = _C.roi_align_forward(torch.from_numpy(x).to(device), torch.from_numpy(rois).to(device), 1 / 16.0,
                               ph, ph, sr)
    assert got.shape == (R, C, ph, ph)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("atomic", [False, True])
@pytest.mark.parametrize("C,H,W,R,ph,sr", [(8, 20, 32, 16, 7, 0), (64, 38, 76, 40, 14, 0), (12, 25, 31, 9, 7, 2),
                                            (3, 9, 7, 300, 7, 0)])
def test_roi_align_backward(device, C, H, W, R, ph, sr, atomic):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(C * 3 + R)
    g = rng.standard_normal((R, C, ph, ph)).astype(np.float32)
    rois = _rois(rng, R, 2, W * 16, H * 16)
    want = O.roi_align_backward(g, rois, 1 / 16.0, ph, ph, 2, C, H, W, sr)
    got = _C.roi_align_backward(torch.from_numpy(g).to(device), torch.from_numpy(rois).to(device), 1 / 16.0,
                                ph, ph, 2, C, H, W, sr, atomic=atomic).cpu().numpy()
    # gather form (default) / atomic scatter form: summation order differs from the oracle -> fp32 tolerance
    tol = 1e-4 if atomic else 1e-5  # the atomic form sums hundreds of contributions per pixel in arrival order
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)


def test_roi_align_empty(device):
    from da_detect_amd import _C

    x = torch.zeros((1, 8, 4, 4), device=device)
    out = _C.roi_align_forward(x, torch.zeros((0, 5), device=device), 0.25, 7, 7, 2)
    assert out.shape == (0, 8, 7, 7)


# --------------------------------------------------------------------------------- ROIPool
@pytest.mark.parametrize("C,H,W,R,ph", [(8, 20, 32, 16, 7), (64, 38, 76, 40, 14), (12, 25, 31, 9, 7), (3, 9, 7, 50, 2)])
def test_roi_pool_forward_backward_bit_exact(device, C, H, W, R, ph):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(C + R)
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    rois = _rois(rng, R, 2, W * 16, H * 16)
    rois[0, 1:] = [-40, -40, -30, -30]            # fully outside -> empty bins (0, argmax -1)
    rois[1, 1:] = [50, 60, 20, 10]                # malformed -> forced to 1x1
    want, warg = O.roi_pool_forward(x, rois, 1 / 16.0, ph, ph)
    xd, rd = torch.from_numpy(x).to(device), torch.from_numpy(rois).to(device)
    got, garg = _C.roi_pool_forward(xd, rd, 1 / 16.0, ph, ph)
    assert garg.dtype == torch.int32 and (warg == -1).any()
    assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(garg.cpu().numpy(), warg)
    # backward: a power-of-two gradient makes the atomic accumulation order-independent -> bit exact
    g = (2.0 ** rng.integers(-3, 3, size=want.shape)).astype(np.float32)
    gwant = O.roi_pool_backward(g, warg, rois, 2, C, H, W)
    ggot = _C.roi_pool_backward(torch.from_numpy(g).to(device), xd, rd, garg, 1 / 16.0, ph, ph, 2, C, H, W)
    assert np.array_equal(ggot.cpu().numpy(), gwant)


def test_roi_pool_layer_autograd_and_empty(device):
    from da_detect_amd.layers import ROIPool

    torch.manual_seed(0)
    x = torch.randn(2, 16, 12, 20, device=device, requires_grad=True)
    rois = torch.tensor([[0, 0, 0, 19 * 16, 11 * 16], [1, 32, 32, 95, 95]], dtype=torch.float32, device=device)
    pool = ROIPool((2, 2), 1 / 16.0)
    y = pool(x, rois)
    # ROI 0 covers the whole map: max over the 2x2 quadrant grid equals adaptive max pooling of image 0
    torch.testing.assert_close(y[0], torch.nn.functional.adaptive_max_pool2d(x[0].detach(), 2))
    y.sum().backward()
    assert x.grad.sum().item() == y.numel()   # every bin routes its unit gradient to exactly one cell
    assert pool(x, rois[:0]).shape == (0, 16, 2, 2)


# --------------------------------------------------------------------------------- convolution
def _conv_case(rng, N, Cin, H, W, Cout, k, stride, pad):
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(w)


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 24, 40, 64, 1, 1, 0),
    (2, 64, 24, 40, 256, 1, 1, 0),
    (1, 256, 30, 34, 128, 1, 2, 0),
    (2, 128, 19, 23, 128, 3, 1, 1),
    (1, 1024, 16, 24, 1024, 3, 1, 1),
    (3, 512, 7, 7, 2048, 1, 1, 0),
    (5, 1024, 14, 14, 512, 1, 2, 0),
    (1, 1024, 9, 13, 76, 1, 1, 0),   # fused RPN cls(15)+bbox(60) padded to 76
    (300, 2048, 1, 1, 48, 1, 1, 0),  # predictor as 1x1 conv on [R,C,1,1]
    (2, 32, 11, 9, 40, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_epilogues(device, case):
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    rng = np.random.default_rng(sum(case))
    x, w = _conv_case(rng, *case)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32))
    ref = F.conv2d(x, w, None, stride, pad)
    res = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    xd, wd = x.to(device).contiguous(memory_format=CL), w.to(device).contiguous(memory_format=CL)
    tol = dict(rtol=2e-5, atol=2e-5)
    got = _C.conv_forward(xd, wd, stride=stride, pad=pad).cpu()
    torch.testing.assert_close(got, ref, **tol)
    got = _C.conv_forward(xd, wd, scale.to(device), bias.to(device), stride=stride, pad=pad, relu_mode=1).cpu()
    torch.testing.assert_close(got, F.relu(ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)), **tol)
    got = _C.conv_forward(xd, wd, scale.to(device), bias.to(device), addend=res.to(device), stride=stride,
                          pad=pad, relu_mode=1).cpu()
    torch.testing.assert_close(got, F.relu(ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res), **tol)
    got = _C.conv_forward(xd, wd, addend=res.to(device), mask_ref=res.to(device), stride=stride, pad=pad,
                          relu_mode=2).cpu()
    torch.testing.assert_close(got, (ref + res) * (res > 0), **tol)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_wgrad(device, case):
    """data / weight gradients against torch autograd on the CPU (fp32)."""
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    rng = np.random.default_rng(sum(case) + 1)
    x, w = _conv_case(rng, *case)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(gy)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    xd = x.detach().to(device).contiguous(memory_format=CL)
    wd = w.detach().to(device).contiguous(memory_format=CL)
    gyd = gy.to(device).contiguous(memory_format=CL)
    # wgrad (with the FrozenBN scale folded in as out_scale)
    dw = _C.conv_wgrad(xd, gyd, tuple(w.shape), stride, pad, out_scale=scale.to(device)).cpu()
    torch.testing.assert_close(dw, w.grad * scale.view(-1, 1, 1, 1), rtol=1e-4, atol=1e-4)
    dw2 = _C.conv_wgrad(xd, gyd, tuple(w.shape), stride, pad, dw=dw.to(device).contiguous(memory_format=CL),
                        accumulate=True).cpu()
    torch.testing.assert_close(dw2, dw + w.grad, rtol=1e-4, atol=1e-4)
    # dgrad = forward kernel on gy with transposed / flipped weights
    wt = _C.conv_weight_transpose(wd)
    if stride == 1:
        dx = _C.conv_forward(gyd, wt, stride=1, pad=k - 1 - pad).cpu()
    else:
        assert k == 1
        dx = _C.conv_forward(gyd, wt, stride=1, pad=0, out_spatial_stride=stride, out_hw=(H, W)).cpu()
    torch.testing.assert_close(dx, x.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode,tol", [(4, 2e-5), (3, 2e-5), (2, 2e-3)])
@pytest.mark.parametrize("case", [CONV_CASES[1], CONV_CASES[3], CONV_CASES[4], CONV_CASES[6], CONV_CASES[9]])
def test_conv_forward_split_bf16_modes(device, case, mode, tol):
    """split contractions: mode 4 (two fp16 terms, 3 MFMAs / K=16, per-tensor scales) and mode 3 (three bf16 terms,
    6 MFMAs / K=16) must hold the exact-fp32 tolerance, mode 2 is ~2^-16"""
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    rng = np.random.default_rng(sum(case) + 5)
    x, w = _conv_case(rng, *case)
    ref64 = F.conv2d(x.double(), w.double(), None, stride, pad)
    xd, wd = x.to(device).contiguous(memory_format=CL), w.to(device).contiguous(memory_format=CL)
    prev = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(0)
        exact = _C.conv_forward(xd, wd, stride=stride, pad=pad).cpu()
        _C.set_gemm_mode(mode)
        got = _C.conv_forward(xd, wd, stride=stride, pad=pad).cpu()
    finally:
        _C.set_gemm_mode(prev)
    scale = float(ref64.abs().mean())
    err_exact = float((exact.double() - ref64).abs().max()) / scale
    err_split = float((got.double() - ref64).abs().max()) / scale
    print("case %s mode %d: max err / mean|y|  exact-fp32 %.2e  split %.2e" % (case, mode, err_exact, err_split))
    assert err_split < tol * 10
    if mode >= 3:  # fp32 class: no worse than a few times the exact-fp32 kernel's own rounding
        assert err_split < max(4 * err_exact, 1e-6)


# the two longest reductions of the BASELINE step: res5 3x3 512 -> 512 on the 7 x 7 maps of 256 ROIs (M = 12544, K = 4608)
# and the RPN 3x3 1024 -> 1024 on a 64 x 128 map (M = 8192, K = 9216)
PRODUCTION_K = [(256, 512, 7, 7, 512, 3, 1, 1), (1, 1024, 64, 128, 1024, 3, 1, 1)]


@pytest.mark.parametrize("case", PRODUCTION_K)
def test_split_bf16_accuracy_at_production_k(device, case):
    """the claim "the split contraction's error is not larger than the exact-fp32 MFMA kernel's", at the production
    reduction lengths: forward outputs and weight gradients of both kernels against float64 dot products of the same
    operands, on 96 sampled output pixels x all channels (forward) and 8 sampled taps x all channel pairs (weight
    gradient: float64 GEMM over all M rows on the device).  Asserted on the RMS error (the maximum over 5e4 samples of
    two error populations of the same size is a coin flip) with 10% slack for that comparison's own noise, and on the
    maximum within 1.5x."""
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    rng = np.random.default_rng(sum(case) + 77)
    x, w = _conv_case(rng, *case)
    xd, wd = x.to(device).contiguous(memory_format=CL), w.to(device).contiguous(memory_format=CL)
    gy = torch.from_numpy(rng.standard_normal((N, Cout, H, W)).astype(np.float32))
    gyd = gy.to(device).contiguous(memory_format=CL)
    prev = _C.get_gemm_mode()
    out = {}
    try:
        for mode in (0, 3, 4):
            _C.set_gemm_mode(mode)
            out[mode] = (_C.conv_forward(xd, wd, stride=stride, pad=pad), _C.conv_wgrad(xd, gyd, tuple(w.shape), stride, pad))
    finally:
        _C.set_gemm_mode(prev)
    # forward: sampled pixels (corners and edges included: padding taps), float64 patches x float64 weights
    pix = [(0, 0, 0), (N - 1, H - 1, W - 1), (0, 0, W - 1), (N - 1, H - 1, 0)]
    pix += [(int(rng.integers(N)), int(rng.integers(H)), int(rng.integers(W))) for _ in range(92)]
    xp = F.pad(x.double(), (pad, pad, pad, pad))
    patches = torch.stack([xp[n, :, h:h + k, w_:w_ + k].reshape(-1) for n, h, w_ in pix])        # [96, Cin*k*k]
    ref = patches @ w.double().reshape(Cout, -1).t()                                               # [96, Cout]
    stats = {}
    for mode in (0, 3, 4):
        y = out[mode][0].cpu().double()
        got = torch.stack([y[n, :, h, w_] for n, h, w_ in pix])
        e = (got - ref) / ref.abs().mean()
        stats[mode] = (float(e.pow(2).mean().sqrt()), float(e.abs().max()))
    print("forward %s: rel. error rms / max  exact-fp32 %.3e / %.3e   bf16x3 %.3e / %.3e   fp16x2 %.3e / %.3e"
          % ((case,) + stats[0] + stats[3] + stats[4]))
    for m in (3, 4):
        assert stats[m][0] <= 1.1 * stats[0][0] and stats[m][1] <= 1.5 * stats[0][1], stats
    # weight gradient: taps (r, s) sampled, dW[:, :, r, s] = gy^T [Cout, M] @ x_shifted [M, Cin] in float64 on the device
    taps = [(0, 0), (1, 1), (2, 2), (0, 2), (2, 0), (1, 0), (0, 1), (2, 1)]
    xpd = F.pad(xd.double(), (pad, pad, pad, pad))
    g2 = gyd.double().permute(1, 0, 2, 3).reshape(Cout, -1)
    stats = {0: [0.0, 0.0, 0], 3: [0.0, 0.0, 0], 4: [0.0, 0.0, 0]}
    for r, s_ in taps:
        xs = xpd[:, :, r:r + H, s_:s_ + W].permute(0, 2, 3, 1).reshape(-1, Cin)
        refw = (g2 @ xs).cpu()
        for mode in (0, 3, 4):
            e = (out[mode][1][:, :, r, s_].cpu().double() - refw) / refw.abs().mean()
            stats[mode][0] += float(e.pow(2).sum())
            stats[mode][1] = max(stats[mode][1], float(e.abs().max()))
            stats[mode][2] += e.numel()
    rms = {m: (stats[m][0] / stats[m][2]) ** 0.5 for m in stats}
    print("wgrad   %s: rel. error rms / max  exact-fp32 %.3e / %.3e   bf16x3 %.3e / %.3e   fp16x2 %.3e / %.3e"
          % (case, rms[0], stats[0][1], rms[3], stats[3][1], rms[4], stats[4][1]))
    for m in (3, 4):
        assert rms[m] <= 1.1 * rms[0] and stats[m][1] <= 1.5 * stats[0][1], (rms, stats)


@pytest.mark.parametrize("sx,sw", [(1.0, 1.0), (3e-12, 7e9), (2.5e7, 1e-3), (1e-30, 1e-5), (6e4, 6e4)])
def test_fp16_split_is_scale_free(device, sx, sw):
    """mode 4 under the per-tensor scales: operands 40 binades below or 25 above one (far outside fp16's range, gradients
    and un-normalised activations look like that) give the relative error of the O(1) case — powers of two in, powers of
    two out.  Also: the slot the epilogue fills holds exactly max|y|."""
    from da_detect_amd import _C, amax

    case = (2, 256, 24, 40, 192, 3, 1, 1)
    rng = np.random.default_rng(11)
    x, w = _conv_case(rng, *case)
    x, w = x * sx, w * sw
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    prev = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(4)
        y = _C.conv_forward(x.to(device).contiguous(memory_format=CL), w.to(device).contiguous(memory_format=CL), pad=1)
        held = amax.value(y)
    finally:
        _C.set_gemm_mode(prev)
    assert held == float(y.abs().max())
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().mean())
    print("scales %g x %g: max err / mean|y| %.2e" % (sx, sw, err))
    assert err < 3e-6


def test_fused_maxima_equal_the_tensors_maximum(device, monkeypatch):
    """mode 4 reads every operand's largest magnitude from a slot its PRODUCER filled.  Each producing kernel form — tiled
    epilogue (16-byte and scalar), stream-K tail, split-K reduction pass, weight-stationary 1x1, the ReLU / FrozenBN
    backward, the image-level DA backward — leaves exactly max|y| there; bounds handed on by hand stay bounds; the batched
    pass over persistent weights agrees with torch."""
    from da_detect_amd import _C, amax

    g = torch.Generator().manual_seed(5)

    def rnd(*shape):
        return torch.randn(shape, generator=g).to(device)

    prev = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(4)
        cases = {
            "tiled 128x128 + residual + relu": dict(x=(2, 256, 40, 64), w=(256, 256, 3, 3), kw=dict(pad=1, relu_mode=1), add=True),
            "stream-K tail": dict(x=(2, 1024, 52, 128), w=(1024, 1024, 1, 1), kw={}),
            "split-K reduce (M = 512 linear)": dict(x=(512, 2048, 1, 1), w=(1024, 2048, 1, 1), kw=dict(relu_mode=1)),
            "weight-stationary 1x1": dict(x=(2, 256, 64, 128), w=(1024, 256, 1, 1), kw={}, add=True),
            "scalar epilogue (strided scatter)": dict(x=(2, 64, 20, 24), w=(128, 64, 1, 1),
                                                      kw=dict(out_spatial_stride=2, out_hw=(40, 48))),
            "ragged 128x64": dict(x=(2, 128, 19, 23), w=(72, 128, 3, 3), kw=dict(pad=1)),
        }
        for name, c in cases.items():
            x = rnd(*c["x"]).contiguous(memory_format=CL)
            w = (rnd(*c["w"]) * 0.05).contiguous(memory_format=CL)
            kw = dict(c["kw"])
            y_shape = None
            if c.get("add"):
                ho, wo = _C.conv_out_size(c["x"][2], c["x"][3], c["w"][2], c["w"][3], 1, kw.get("pad", 0))
                kw["addend"] = rnd(c["x"][0], c["w"][0], ho, wo).contiguous(memory_format=CL)
            y = _C.conv_forward(x, w, **kw)
            assert amax.value(y) == float(y.abs().max()), name
            assert amax.value(x) == float(x.abs().max()), name            # measured on demand (dadet_amax)
        # elementwise producers
        gy, yy = rnd(2, 256, 24, 40).contiguous(memory_format=CL), rnd(2, 256, 24, 40).contiguous(memory_format=CL)
        sc = torch.rand(256, generator=g).to(device) + 0.5
        g_out, g_sc = _C.relu_bn_backward(gy, yy, sc, want_unscaled=True)
        assert amax.value(g_out) == float(g_out.abs().max()) and amax.value(g_sc) == float(g_sc.abs().max())
        amax.measure(gy)
        g_out2, _ = _C.relu_bn_backward(gy, yy, sc, want_unscaled=True)        # the gate's input carries one: a bound
        assert amax.value(g_out2) == float(gy.abs().max()) >= float(g_out2.abs().max())
        pooled = _C.maxpool3x3s2(amax.measure(rnd(2, 64, 32, 48).contiguous(memory_format=CL)))
        assert amax.value(pooled) >= float(pooled.abs().max())
        # image-level DA backward
        N, H, W, C1 = 2, 12, 20, 512
        t = rnd(N, C1, H, W).clamp_min(0).contiguous(memory_format=CL)
        w2, b2 = rnd(C1) * 0.05, rnd(1)
        labels = torch.tensor([1.0, 0.0], device=device)
        logits, _ = _C.da_img_head_loss_forward(t, w2, b2, labels, N, H * W)
        gw, gx, _, _ = _C.da_img_head_loss_backward_g(t, w2, logits, labels, torch.ones(1, device=device),
                                                      torch.full((N,), 0.3, device=device), 0.7, 0.2, N, H * W)
        assert amax.value(gw) == float(gw.abs().max()) and amax.value(gx) == float(gx.abs().max())
        # persistent weights: one batched pass per weight epoch
        params = [torch.nn.Parameter((rnd(*s_) * k_).contiguous(memory_format=CL))
                  for s_, k_ in (((64, 64, 3, 3), 0.1), ((256, 64, 1, 1), 3.0), ((1024, 1024, 3, 3), 1e-3))]
        xs = rnd(2, 64, 16, 16).contiguous(memory_format=CL)
        with torch.no_grad():
            _C.conv_forward(xs, params[0], pad=1)
            _C.conv_forward(xs, params[1])
            _C.conv_forward(rnd(1, 1024, 8, 8).contiguous(memory_format=CL), params[2], pad=1)
            for p_ in params:
                p_.data.mul_(1.5)                 # a raw in-place update, as the fused optimizer makes it
        _C.bump_weight_epoch(xs.device)
        d = amax.WEIGHTS.by_dev[xs.device.index]
        for p_ in params:
            e = d["entries"][(p_.data_ptr(), p_.numel())]
            assert float(d["slots"][:, e["i"]].max()) == float(p_.detach().abs().max())
    finally:
        _C.set_gemm_mode(prev)


def test_fp16_split_wide_range_inside_one_tensor(device):
    """elements far below the tensor's maximum: above 2^-16 of it they keep fp32-class relative accuracy, below that an
    absolute error under 2^-38 of the maximum (the low term falls into fp16's subnormals) — rows whose values are all
    tiny are still good to ~1e-6 of THEIR OWN size down to 2^-20 of the maximum"""
    from da_detect_amd import _C

    rng = np.random.default_rng(12)
    M, K, Cout = 4096, 1024, 256
    x = rng.standard_normal((M, K)).astype(np.float32)
    rowscale = np.exp2(-rng.integers(0, 24, (M, 1)).astype(np.float32))      # rows between 1 and 2^-23 of the largest
    rowscale[0] = 1.0
    x = x * rowscale
    w = (rng.standard_normal((Cout, K)) / np.sqrt(K)).astype(np.float32)
    xt = torch.from_numpy(x).view(1, M, 1, K).permute(0, 3, 1, 2).contiguous(memory_format=CL)     # [1, K, M, 1]
    wt = torch.from_numpy(w).view(Cout, K, 1, 1).contiguous(memory_format=CL)
    ref = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t()
    prev = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(4)
        y = _C.conv_forward(xt.to(device), wt.to(device)).cpu().permute(0, 2, 3, 1).reshape(M, Cout).double()
        _C.set_gemm_mode(0)
        y0 = _C.conv_forward(xt.to(device), wt.to(device)).cpu().permute(0, 2, 3, 1).reshape(M, Cout).double()
    finally:
        _C.set_gemm_mode(prev)
    amax = float(np.abs(x).max())
    rs = torch.from_numpy(rowscale.astype(np.float64))
    e4 = (y - ref).abs().max(dim=1, keepdim=True).values
    e0 = (y0 - ref).abs().max(dim=1, keepdim=True).values
    big = rs[:, 0] >= 2.0 ** -14
    rel4, rel0 = float((e4 / rs)[big].max()), float((e0 / rs)[big].max())
    print("rows within 2^-14 of the maximum: max err / row scale  fp16x2 %.2e  exact-fp32 %.2e" % (rel4, rel0))
    assert rel4 < 4 * rel0
    print("all rows: max abs err / tensor max  %.2e" % (float(e4.max()) / amax))
    assert float(e4[~big].max()) < 2.0 ** -34 * amax * 4


@pytest.mark.parametrize("mode,tol", [(4, 1e-4), (3, 1e-4), (2, 2e-3)])
@pytest.mark.parametrize("case", [CONV_CASES[1], CONV_CASES[2], CONV_CASES[3], CONV_CASES[6], CONV_CASES[7], CONV_CASES[9]])
def test_conv_wgrad_split_bf16_modes(device, case, mode, tol):
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    rng = np.random.default_rng(sum(case) + 9)
    x, w = _conv_case(rng, *case)
    Ho, Wo = _C.conv_out_size(H, W, k, k, stride, pad)
    gy = torch.from_numpy(rng.standard_normal((N, Cout, Ho, Wo)).astype(np.float32))
    w64 = w.double().requires_grad_(True)
    F.conv2d(x.double(), w64, None, stride, pad).backward(gy.double())
    xd, gyd = x.to(device).contiguous(memory_format=CL), gy.to(device).contiguous(memory_format=CL)
    prev = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(0)
        exact = _C.conv_wgrad(xd, gyd, tuple(w.shape), stride, pad).cpu()
        _C.set_gemm_mode(mode)
        got = _C.conv_wgrad(xd, gyd, tuple(w.shape), stride, pad).cpu()
    finally:
        _C.set_gemm_mode(prev)
    scale = float(w64.grad.abs().mean())
    err_exact = float((exact.double() - w64.grad).abs().max()) / scale
    err_split = float((got.double() - w64.grad).abs().max()) / scale
    print("wgrad case %s mode %d: exact-fp32 %.2e  split %.2e" % (case, mode, err_exact, err_split))
    assert err_split < tol
    if mode >= 3:
        assert err_split < max(4 * err_exact, 1e-6)


SPLITK_CASES = [
    # small tile grids: the K range is cut over blockIdx.y and a reduce pass applies the epilogue (mode 3 only)
    (512, 2048, 1, 1, 1024, 1, 1, 0),    # instance-head fc1 on 512 ROIs
    (512, 1024, 1, 1, 4, 1, 1, 0),       # instance-head logit (Cout padded to 4)
    (300, 2048, 1, 1, 48, 1, 1, 0),      # box predictor
    (1, 64, 16, 16, 64, 3, 1, 1),        # M = 256, K = 576
    (2, 256, 8, 8, 256, 3, 1, 1),        # M = 128, K = 2304 (a coarse pyramid level)
    (1, 260, 9, 7, 36, 1, 1, 0),         # ragged everything: M = 63, K = 260, Cout = 36
]


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_conv_forward_split_k_epilogues(device, case):
    """same results as the unsplit kernel (DADET_SPLITK=0 is the library default off-switch; here the reference is
    torch fp64) for every epilogue, with the K reduction cut over several workgroups"""
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    rng = np.random.default_rng(sum(case) + 3)
    x, w = _conv_case(rng, *case)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32))
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    res = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    xd, wd = x.to(device).contiguous(memory_format=CL), w.to(device).contiguous(memory_format=CL)
    sd, bd, rd = scale.to(device), bias.to(device), res.to(device).contiguous(memory_format=CL)
    tol = dict(rtol=2e-5, atol=2e-5)
    prev = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(3)
        got = _C.conv_forward(xd, wd, stride=stride, pad=pad).cpu()
        torch.testing.assert_close(got.double(), ref, **tol)
        got = _C.conv_forward(xd, wd, sd, bd, stride=stride, pad=pad, relu_mode=1).cpu()
        want = F.relu(ref * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1))
        torch.testing.assert_close(got.double(), want, **tol)
        got = _C.conv_forward(xd, wd, sd, bd, addend=rd, stride=stride, pad=pad, relu_mode=1).cpu()
        torch.testing.assert_close(got.double(), F.relu(ref * scale.double().view(1, -1, 1, 1) +
                                                        bias.double().view(1, -1, 1, 1) + res.double()), **tol)
        got = _C.conv_forward(xd, wd, addend=rd, mask_ref=rd, stride=stride, pad=pad, relu_mode=2).cpu()
        torch.testing.assert_close(got.double(), (ref + res.double()) * (res > 0), **tol)
        out = rd.clone()                                 # in-place: the addend is also the output
        _C.conv_forward(xd, wd, addend=out, out=out, stride=stride, pad=pad)
        torch.testing.assert_close(out.cpu().double(), ref + res.double(), **tol)
    finally:
        _C.set_gemm_mode(prev)


def test_stem_conv7x7_as_padded_7x8(device):
    """BaseStem conv (3->64, 7x7, s2, p3) through NHWC4 staging and a 7x8 zero-padded kernel."""
    from da_detect_amd import _C

    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((2, 3, 45, 67)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((64, 3, 7, 7)) / 12).astype(np.float32))
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, None, 2, 3)), 3, 2, 1)
    x4 = _C.nchw3_to_nhwc4(x.to(device))
    w4 = torch.zeros((64, 4, 7, 8), dtype=torch.float32)
    w4[:, :3, :, :7] = w
    Ho, Wo = (45 + 6 - 7) // 2 + 1, (67 + 6 - 7) // 2 + 1
    y = _C.conv_forward(x4, w4.to(device).contiguous(memory_format=CL), stride=2, pad=3, relu_mode=1,
                        out_size=(Ho, Wo))
    got = _C.maxpool3x3s2(y).cpu()
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)


# --------------------------------------------------------------------------- elementwise helpers
def test_relu_bn_backward_colsum_avgpool(device):
    from da_detect_amd import _C

    rng = np.random.default_rng(5)
    g = torch.from_numpy(rng.standard_normal((3, 64, 9, 11)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((3, 64, 9, 11)).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 2, 64).astype(np.float32))
    gm, gs = _C.relu_bn_backward(g.to(device), y.to(device), scale.to(device), want_unscaled=True)
    torch.testing.assert_close(gm.cpu(), g * (y > 0))
    torch.testing.assert_close(gs.cpu(), g * (y > 0) * scale.view(1, -1, 1, 1))
    cs = _C.colsum(g.to(device).contiguous(memory_format=CL)).cpu()
    torch.testing.assert_close(cs, g.sum((0, 2, 3)), rtol=1e-4, atol=1e-4)
    x = torch.from_numpy(rng.standard_normal((10, 128, 7, 7)).astype(np.float32))
    ap = _C.avgpool_forward(x.to(device)).cpu()
    torch.testing.assert_close(ap, F.avg_pool2d(x, 7).flatten(1), rtol=1e-5, atol=1e-6)
    gx = _C.avgpool_backward(ap.to(device), 7, 7).cpu()
    torch.testing.assert_close(gx, (ap / 49).view(10, 128, 1, 1).expand(10, 128, 7, 7))
    ca = _C.channel_affine(g.to(device), scale.to(device), scale.to(device), relu=True).cpu()
    torch.testing.assert_close(ca, F.relu(g * scale.view(1, -1, 1, 1) + scale.view(1, -1, 1, 1)))


def test_sigmoid_focal_loss(device):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(9)
    logits = (rng.standard_normal((500, 80)) * 4).astype(np.float32)
    targets = rng.integers(-1, 81, 500).astype(np.int32)
    dl = rng.standard_normal((500, 80)).astype(np.float32)
    want = O.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25)
    got = _C.sigmoid_focalloss_forward(torch.from_numpy(logits).to(device), torch.from_numpy(targets).to(device),
                                       80, 2.0, 0.25).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
    wantb = O.sigmoid_focal_loss_backward(logits, targets, dl, 2.0, 0.25)
    gotb = _C.sigmoid_focalloss_backward(torch.from_numpy(logits).to(device), torch.from_numpy(targets).to(device),
                                         torch.from_numpy(dl).to(device), 80, 2.0, 0.25).cpu().numpy()
    np.testing.assert_allclose(gotb, wantb, rtol=2e-5, atol=1e-6)


def test_rpn_decode_clip(device):
    from da_detect_amd import _C
    from oracle import ops as O

    rng = np.random.default_rng(21)
    A = 4000
    anchors = _rand_boxes(rng, A, 1200, 600, 500)
    deltas = (rng.standard_normal((A, 4)) * 0.5).astype(np.float32)
    deltas[:5, 2:] = 9.0  # exercises the log(1000/16) clamp
    idx = rng.permutation(A)[:1500].astype(np.int64)
    clip = float(np.log(1000.0 / 16))
    want = O.decode_clip(deltas[idx], anchors[idx], (1.0, 1.0, 1.0, 1.0), clip, 1200, 600)
    got = _C.rpn_decode_clip(torch.from_numpy(deltas).to(device), torch.from_numpy(anchors).to(device),
                             torch.from_numpy(idx).to(device), (1.0, 1.0, 1.0, 1.0), clip, 1200, 600).cpu().numpy()
    # expf may differ by an ulp between libm and the device
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-3)


# ------------------------------------------------------------------------------------- DA heads
def test_da_img_head_loss_forward_backward(device):
    from da_detect_amd import _C

    rng = np.random.default_rng(33)
    B, HW, C1 = 2, 37 * 5, 512
    t = torch.from_numpy(np.maximum(rng.standard_normal((B * HW, C1)), 0).astype(np.float32)).requires_grad_(True)
    w2 = torch.from_numpy((rng.standard_normal(C1) * 0.05).astype(np.float32)).requires_grad_(True)
    b2 = torch.tensor([0.1], requires_grad=True)
    labels = torch.tensor([1.0, 0.0])
    logits = t @ w2 + b2
    lab_rows = labels.repeat_interleave(HW)
    bce = F.binary_cross_entropy_with_logits(logits, lab_rows)
    mean_sig = torch.sigmoid(logits).view(B, HW).mean(1)
    got_logits, sums = _C.da_img_head_loss_forward(t.detach().to(device), w2.detach().to(device),
                                                   b2.detach().to(device), labels.to(device), B, HW)
    torch.testing.assert_close(got_logits.cpu(), logits.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(sums[:, 0].sum().cpu() / (B * HW), bce.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sums[:, 1].cpu() / HW, mean_sig.detach(), rtol=1e-5, atol=1e-6)
    # backward: loss = 1.0*bce + sum_i k_i * mean_sig_i ; feature path uses reversal weights -0.1 / +0.1
    k = torch.tensor([0.3, -0.7])
    (bce + (k * mean_sig).sum()).backward()
    coef = torch.stack([torch.full((B,), 1.0 / (B * HW)), k / HW, torch.full((B,), -0.1 / (B * HW)),
                        0.1 * k / HW], 1).contiguous()
    g_t_w, g_t_x, g_w2, g_b2 = _C.da_img_head_loss_backward(t.detach().to(device), w2.detach().to(device),
                                                            got_logits, labels.to(device), coef.to(device), B, HW)
    mask = (t.detach() > 0).float()
    torch.testing.assert_close(g_t_w.cpu(), t.grad * mask, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(g_w2.cpu(), w2.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(g_b2.cpu(), b2.grad, rtol=1e-4, atol=1e-6)
    # g_t_x: same with the two paths reweighted
    t2 = t.detach().clone().requires_grad_(True)
    lg = t2 @ w2.detach() + b2.detach()
    (-0.1 * F.binary_cross_entropy_with_logits(lg, lab_rows) + 0.1 * (k * torch.sigmoid(lg).view(B, HW).mean(1)).sum()).backward()
    torch.testing.assert_close(g_t_x.cpu(), t2.grad * mask, rtol=1e-4, atol=1e-7)


def test_triplet_w_loss(device):
    from da_detect_amd import _C

    rng = np.random.default_rng(41)
    a, p, n = [torch.from_numpy(rng.standard_normal((1, 64, 9, 13)).astype(np.float32)).requires_grad_(True)
               for _ in range(3)]
    ref = torch.nn.TripletMarginLoss(margin=1.0, p=2)(a, p, n)
    ref.backward()
    loss, dist = _C.triplet_w_forward(a.detach().to(device), p.detach().to(device), n.detach().to(device), 1.0)
    torch.testing.assert_close(loss.cpu()[0], ref.detach(), rtol=1e-5, atol=1e-6)
    g = torch.tensor([1.0 / (64 * 9)], device=device)
    ga, gp, gn = _C.triplet_w_backward(a.detach().to(device), p.detach().to(device), n.detach().to(device), dist,
                                       g, 1.0)
    torch.testing.assert_close(ga.cpu(), a.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(gp.cpu(), p.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(gn.cpu(), n.grad, rtol=1e-4, atol=1e-7)


# --------------------------------------------------------------------------------- box-head target assignment
@pytest.mark.parametrize("P,G,high,low", [(2000, 12, 0.5, 0.5), (517, 1, 0.7, 0.3), (64, 37, 0.5, 0.0)])
def test_box_match_encode_matches_the_aten_chain(device, P, G, high, low):
    """one-launch IoU -> Matcher -> labels -> BoxCoder.encode vs the reference-order torch chain on the CPU
    (structures/boxlist_ops.py:56-91, modeling/matcher.py:42-92, box_head/loss.py:69-93, box_coder.py:22-50)"""
    from da_detect_amd import _C
    from oracle import model_ref as M

    rng = np.random.default_rng(P + G)
    props = torch.from_numpy(_rand_boxes(rng, P))
    gts = torch.from_numpy(_rand_boxes(rng, G))
    props[:G] = gts                                     # exact matches (IoU 1) like add_gt_proposals produces
    gl = torch.from_numpy(rng.integers(1, 9, G)).to(torch.int64)
    weights = (10.0, 10.0, 5.0, 5.0)
    iou = M.box_iou(gts, props)
    want_m = M.matcher(iou, high, low, False)
    want_lab = gl[want_m.clamp(min=0)].clone()
    want_lab[want_m == M.BELOW_LOW] = 0
    want_lab[want_m == M.BETWEEN] = -1
    want_reg = M.encode(gts[want_m.clamp(min=0)], props, weights)
    m, lab, reg = _C.box_match_encode(props.to(device), gts.to(device), gl.to(device), high, low, weights)
    assert m.dtype == torch.int64 and torch.equal(m.cpu(), want_m) and torch.equal(lab.cpu(), want_lab)
    torch.testing.assert_close(reg.cpu(), want_reg, rtol=1e-5, atol=1e-6)


# --------------------------------------------------------------------------------- fused detection losses
def test_fused_rpn_loss_matches_torch(device):
    import torch.nn.functional as F

    from da_detect_amd.layers.misc import rpn_loss_fused, smooth_l1_loss

    g = torch.Generator().manual_seed(0)
    N, A, H, W = 2, 15, 12, 20
    obj = (torch.randn(N, A, H, W, generator=g) * 3).to(device).contiguous(memory_format=torch.channels_last)
    reg = torch.randn(N, 4 * A, H, W, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    total = N * A * H * W
    perm = torch.randperm(total, generator=g)
    pos, neg = perm[:70].sort().values.to(device), perm[70:256].sort().values.to(device)
    sampled = torch.cat([pos, neg])
    labels = torch.cat([torch.ones(70), torch.zeros(186)]).to(device)
    tgt = (torch.randn(70, 4, generator=g) * 0.3).to(device)
    a, b = obj.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    l0, l1 = rpn_loss_fused(a, b, sampled, labels, pos, tgt, 1.0 / 9)
    (2.0 * l0 + 0.5 * l1).backward()
    c, d = obj.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    flat_o = c.permute(0, 2, 3, 1).reshape(-1)
    flat_r = d.view(N, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(-1, 4)
    w0 = F.binary_cross_entropy_with_logits(flat_o[sampled], labels)
    w1 = smooth_l1_loss(flat_r[pos], tgt, beta=1.0 / 9, size_average=False) / sampled.numel()
    (2.0 * w0 + 0.5 * w1).backward()
    torch.testing.assert_close(l0, w0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(l1, w1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, c.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b.grad, d.grad, rtol=1e-5, atol=1e-7)


def test_fused_fast_rcnn_loss_matches_torch(device):
    import torch.nn.functional as F

    from da_detect_amd.layers.misc import fast_rcnn_loss_fused, smooth_l1_loss

    g = torch.Generator().manual_seed(1)
    R, C = 512, 9
    logits = (torch.randn(R, C, generator=g) * 2).to(device)
    reg = torch.randn(R, 4 * C, generator=g).to(device)
    src = torch.arange(0, 256, device=device)                       # source-domain rows
    labels = torch.randint(0, C, (256,), generator=g).to(device)
    pos = torch.nonzero(labels > 0).squeeze(1)
    map_inds = 4 * labels[pos][:, None] + torch.tensor([0, 1, 2, 3], device=device)
    tgt = (torch.randn(pos.numel(), 4, generator=g) * 0.5).to(device)
    a, b = logits.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    l0, l1 = fast_rcnn_loss_fused(a, b, src, labels, src[pos], map_inds, tgt)
    (l0 + 3.0 * l1).backward()
    c, d = logits.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    w0 = F.cross_entropy(c.index_select(0, src), labels)
    w1 = smooth_l1_loss(d[src[pos][:, None], map_inds], tgt, size_average=False, beta=1) / labels.numel()
    (w0 + 3.0 * w1).backward()
    torch.testing.assert_close(l0, w0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(l1, w1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, c.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b.grad, d.grad, rtol=1e-5, atol=1e-7)
    assert float(a.grad[256:].abs().sum()) == 0.0                    # target-domain rows get no detection gradient


@pytest.mark.parametrize("agnostic", [False, True])
def test_fast_rcnn_loss_rows_matches_torch_and_the_index_kernel(device, agnostic):
    """per-row targets (label < 0 = target-domain row) vs box_head/loss.py:186-219 in torch, and vs the index-list
    kernel on the same sample"""
    import torch.nn.functional as F

    from da_detect_amd.layers.misc import fast_rcnn_loss_fused, fast_rcnn_loss_rows_fused, smooth_l1_loss

    g = torch.Generator().manual_seed(5)
    R, C = 500, 9
    reg_cols = 8 if agnostic else 4 * C
    logits = (torch.randn(R, C, generator=g) * 2).to(device)
    reg = torch.randn(R, reg_cols, generator=g).to(device)
    labels = torch.randint(0, C, (R,), generator=g)
    labels[torch.rand(R, generator=g) < 0.5] = 0
    source = torch.rand(R, generator=g) < 0.55                       # rows of source-domain images, interleaved
    loss_labels = torch.where(source, labels, torch.full_like(labels, -1)).to(device)
    tgt_rows = (torch.randn(R, 4, generator=g) * 0.5).to(device)
    a, b = logits.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    l0, l1 = fast_rcnn_loss_rows_fused(a, b, loss_labels, tgt_rows)
    (l0 + 3.0 * l1).backward()
    src = torch.nonzero(source).squeeze(1).to(device)
    labels_src = labels.to(device)[src]
    pos = torch.nonzero(labels_src > 0).squeeze(1)
    if agnostic:
        map_inds = torch.arange(4, 8, device=device).repeat(pos.numel(), 1)
    else:
        map_inds = 4 * labels_src[pos][:, None] + torch.arange(4, device=device)
    c, d = logits.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    w0 = F.cross_entropy(c.index_select(0, src), labels_src)
    w1 = smooth_l1_loss(d[src[pos][:, None], map_inds], tgt_rows[src[pos]], size_average=False, beta=1) / src.numel()
    (w0 + 3.0 * w1).backward()
    torch.testing.assert_close(l0, w0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(l1, w1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, c.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b.grad, d.grad, rtol=1e-5, atol=1e-7)
    assert float(a.grad[~source.to(device)].abs().sum()) == 0.0
    k0, k1 = fast_rcnn_loss_fused(logits, reg, src, labels_src, src[pos], map_inds, tgt_rows[src[pos]])
    torch.testing.assert_close(l0, k0, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(l1, k1, rtol=1e-6, atol=1e-7)
    # no source rows at all: both losses 0, no gradient
    z0, z1 = fast_rcnn_loss_rows_fused(logits, reg, torch.full((R,), -1, dtype=torch.int64, device=device), tgt_rows)
    assert float(z0) == 0.0 and float(z1) == 0.0


@pytest.mark.parametrize("n", [0, 1, 37, 256, 300, 2048, 2500, 4096])
def test_sample_rois_takes_a_balanced_sample_in_proposal_order(device, n):
    """dadet_sample_rois vs the rules of balanced_positive_negative_sampler.py:40-52 + box_head/loss.py:118-127:
    counts, classes, ignored rows never taken, ascending order, gathered fields, per-row loss labels"""
    from da_detect_amd import _C

    rng = np.random.default_rng(n)
    cap, max_pos = 256, 64
    boxes = torch.from_numpy(_rand_boxes(rng, max(n, 1))[:n]).to(device).reshape(n, 4)
    reg = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)).to(device)
    for frac_pos, frac_ign in ((0.1, 0.1), (0.5, 0.0), (0.0, 0.3), (1.0, 0.0)):
        u = rng.uniform(0, 1, n)
        lab = np.where(u < frac_pos, rng.integers(1, 9, n), 0)
        lab = np.where(rng.uniform(0, 1, n) < frac_ign, -1, lab).astype(np.int64)
        labels = torch.from_numpy(lab).to(device)
        n_pos, n_neg = int((lab >= 1).sum()), int((lab == 0).sum())
        want_pos = min(n_pos, max_pos)
        want_neg = min(n_neg, cap - want_pos)
        for is_source in (True, False):
            counts = torch.zeros(2, dtype=torch.int32, device=device)
            out = _C.sample_rois(boxes, labels, reg, cap, max_pos, 1234 + n, is_source, counts)
            k, p = counts.tolist()
            assert (k, p) == (want_pos + want_neg, want_pos)
            idx = out["idx"][:k].cpu()
            assert torch.equal(idx, idx.sort().values) and idx.unique().numel() == k
            assert k == 0 or (int(idx.min()) >= 0 and int(idx.max()) < n)
            got = labels.cpu()[idx]
            assert int((got >= 1).sum()) == want_pos and int((got == 0).sum()) == want_neg
            assert torch.equal(out["labels"][:k].cpu(), got)
            assert torch.equal(out["boxes"][:k].cpu(), boxes.cpu()[idx])
            assert torch.equal(out["regression_targets"][:k].cpu(), reg.cpu()[idx])
            assert torch.equal(out["loss_labels"][:k].cpu(), got if is_source else torch.full_like(got, -1))
            assert bool((out["domain"][:k] == is_source).all())
            assert torch.equal(out["idx"][k:].cpu(), torch.full((cap - k,), -1, dtype=torch.int64))
    # labels None: every proposal is a negative with zero targets
    counts = torch.zeros(2, dtype=torch.int32, device=device)
    out = _C.sample_rois(boxes, None, None, cap, max_pos, 7, False, counts)
    k, p = counts.tolist()
    assert (k, p) == (min(n, cap), 0) and float(out["regression_targets"].abs().sum()) == 0.0
    assert not bool(out["domain"].any()) and bool((out["loss_labels"] == -1).all())


@pytest.mark.parametrize("A,frac_pos,frac_ign", [(0, 0.0, 0.0), (5, 0.5, 0.0), (3000, 0.001, 0.3),
                                                   (122880, 0.0005, 0.4), (122880, 0.01, 0.2), (40000, 0.0, 0.1),
                                                   (523776, 0.0003, 0.5), (700, 1.0, 0.0)])
def test_sample_anchors_takes_a_balanced_sample_in_anchor_order(device, A, frac_pos, frac_ign):
    """dadet_sample_anchors vs the rules of balanced_positive_negative_sampler.py:40-52 on RPN-sized label vectors:
    counts, classes, ignored anchors never taken, ascending order, index offset, gathered positive targets"""
    from da_detect_amd import _C

    rng = np.random.default_rng(A + 1)
    cap, max_pos, offset = 256, 128, 1000000
    u = rng.uniform(0, 1, A)
    lab = np.where(u < frac_pos, 1.0, 0.0)
    lab = np.where(rng.uniform(0, 1, A) < frac_ign, -1.0, lab).astype(np.float32)
    labels = torch.from_numpy(lab).to(device)
    reg = torch.from_numpy(rng.standard_normal((A, 4)).astype(np.float32)).to(device)
    n_pos, n_neg = int((lab >= 1).sum()), int((lab == 0).sum())
    want_pos = min(n_pos, max_pos)
    want_neg = min(n_neg, cap - want_pos)
    counts = torch.zeros(2, dtype=torch.int32, device=device)
    out = _C.sample_anchors(labels, reg, cap, max_pos, 77 + A, offset, counts)
    p, q = counts.tolist()
    assert (p, q) == (want_pos, want_neg)
    pos, neg = out["pos"][:p].cpu() - offset, out["neg"][:q].cpu() - offset
    for idx, cls, k in ((pos, 1.0, p), (neg, 0.0, q)):
        assert torch.equal(idx, idx.sort().values) and idx.unique().numel() == k
        assert k == 0 or (int(idx.min()) >= 0 and int(idx.max()) < A)
        assert bool((labels.cpu()[idx] == cls).all())
    assert torch.equal(out["regression_targets_pos"][:p].cpu(), reg.cpu()[pos])
    assert bool((out["pos"][p:] == -1).all()) and bool((out["neg"][q:] == -1).all())
    again = _C.sample_anchors(labels, reg, cap, max_pos, 77 + A, offset, counts)
    assert torch.equal(again["neg"], out["neg"]) and torch.equal(again["pos"], out["pos"])
    if n_neg > cap:
        other = _C.sample_anchors(labels, reg, cap, max_pos, 78 + A, offset, counts)
        assert not torch.equal(other["neg"], out["neg"])


@pytest.mark.parametrize("A,frac_pos,frac_ign", [
    (122880, 0.0005, 0.4),      # C4 at 1024 x 2048: a few positives, ~70 000 negatives
    (523776, 0.0003, 0.5),      # five-level pyramid
    (122880, 0.01, 0.2),        # more positives than max_pos: the positives are thresholded too
    (122880, 0.08, 0.0),        # > 4096 positives: candidate list overflow -> one-workgroup algorithm
    (65536, 0.0, 0.999),        # ~65 negatives in all: every one is taken (fewer than wanted)
    (200000, 0.0, 0.9985),      # ~300 negatives, 256 wanted, ~3 listed: the lists do not hold the answer -> fallback
    (40000, 0.001, 0.1),
])
def test_chip_wide_anchor_scan_equals_the_one_workgroup_sampler(device, A, frac_pos, frac_ign):
    """dadet_sample_anchors for A >= 32768 scans the labels with one workgroup per 4096 anchors and finishes in one
    (csrc/sampling.hip, round 4); the result must be the one-workgroup kernel's (DADET_ANCHOR_SCAN=0) bit for bit — the k
    smallest keys per class, ties to the lower index — including when the candidate lists overflow or fall short and the
    finishing workgroup runs the old algorithm; and the per-stream scratch must be clean for the next call"""
    import os

    from da_detect_amd import _C

    rng = np.random.default_rng(A + int(frac_pos * 1e6))
    cap, max_pos, offset = 256, 128, 12345
    lab = np.where(rng.uniform(0, 1, A) < frac_pos, 1.0, 0.0)
    lab = np.where(rng.uniform(0, 1, A) < frac_ign, -1.0, lab).astype(np.float32)
    labels = torch.from_numpy(lab).to(device)
    reg = torch.from_numpy(rng.standard_normal((A, 4)).astype(np.float32)).to(device)
    for seed in (3, 0xDEADBEEFCAFE, 2 ** 63 + 11):
        got = {}
        for flag in ("0", "1", "1"):            # the scan twice in a row: counters left at zero by the first call
            os.environ["DADET_ANCHOR_SCAN"] = flag
            try:
                counts = torch.full((2,), -7, dtype=torch.int32, device=device)
                out = _C.sample_anchors(labels, reg, cap, max_pos, seed, offset, counts)
            finally:
                os.environ.pop("DADET_ANCHOR_SCAN")
            cur = (counts.tolist(), out["pos"].cpu(), out["neg"].cpu(), out["regression_targets_pos"].cpu())
            if flag in got:
                prev = got[flag]
                assert cur[0] == prev[0] and all(torch.equal(a, b) for a, b in zip(cur[1:], prev[1:]))
            got[flag] = cur
        assert got["0"][0] == got["1"][0], (got["0"][0], got["1"][0])
        for a, b in zip(got["0"][1:], got["1"][1:]):
            assert torch.equal(a, b)


def test_sample_anchors_is_uniform(device):
    """over many seeds every negative (and every positive when there are more than max_pos) is taken equally often"""
    from da_detect_amd import _C

    A, cap, max_pos = 400, 64, 16
    labels = torch.zeros(A, dtype=torch.float32, device=device)
    labels[:40] = 1.0
    labels[40:60] = -1.0
    reg = torch.zeros((A, 4), dtype=torch.float32, device=device)
    counts = torch.zeros(2, dtype=torch.int32, device=device)
    trials = 3000
    hits = torch.zeros(A, dtype=torch.int64)
    for seed in range(trials):
        out = _C.sample_anchors(labels, reg, cap, max_pos, seed * 2654435761 + 5, 0, counts)
        hits[out["pos"][:max_pos].cpu()] += 1
        hits[out["neg"][:cap - max_pos].cpu()] += 1
    assert int(hits[40:60].sum()) == 0
    for sl, prob in ((slice(0, 40), 16 / 40), (slice(60, A), 48 / 340)):
        sigma = (trials * prob * (1 - prob)) ** 0.5
        assert float((hits[sl].float() - trials * prob).abs().max()) < 5 * sigma


def test_sample_rois_is_seeded_and_uniform(device):
    """same seed -> same sample, other seed -> another one; over many seeds every negative is taken equally often"""
    from da_detect_amd import _C

    rng = np.random.default_rng(3)
    n, cap = 64, 16
    boxes = torch.from_numpy(_rand_boxes(rng, n)).to(device)
    labels = torch.zeros(n, dtype=torch.int64, device=device)
    labels[:8] = 3                                                   # 8 positives, 4 of them may be taken
    counts = torch.zeros(2, dtype=torch.int32, device=device)
    a = _C.sample_rois(boxes, labels, None, cap, 4, 99, True, counts)["idx"].clone()
    b = _C.sample_rois(boxes, labels, None, cap, 4, 99, True, counts)["idx"].clone()
    c = _C.sample_rois(boxes, labels, None, cap, 4, 100, True, counts)["idx"].clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    trials = 3000
    hits = torch.zeros(n, dtype=torch.int64)
    for seed in range(trials):
        idx = _C.sample_rois(boxes, labels, None, cap, 4, seed * 2654435761 + 17, True, counts)["idx"]
        hits[idx.cpu()] += 1
    p_pos, p_neg = 4 / 8, 12 / 56
    for sl, prob in ((slice(0, 8), p_pos), (slice(8, n), p_neg)):
        sigma = (trials * prob * (1 - prob)) ** 0.5
        assert float((hits[sl].float() - trials * prob).abs().max()) < 5 * sigma


@pytest.mark.parametrize("n,post_n,is_source,with_gt", [(3000, 2000, True, True), (3000, 2000, False, False),
                                                        (700, 2000, True, True), (50, 2000, True, True),
                                                        (1200, 300, True, False)])
def test_proposals_sample_equals_the_host_chain(device, n, post_n, is_source, with_gt):
    """dadet_proposals_sample (NMS result read on the device: kept boxes + appended ground truth + target assignment +
    sample, one launch) against the chain it replaces — kept-count read-back, gathers, concatenation,
    dadet_box_match_encode, dadet_sample_rois — on the same seed: every output identical, bit for bit; and
    PendingProposals materialises to the same list.  Cases: more / fewer kept boxes than post_nms_top_n, fewer proposals
    than the sample size, a target-domain image, matching against ground truth that is not appended (aligned passes)."""
    from da_detect_amd import _C
    from da_detect_amd.structures.bounding_box import BoxList, PendingProposals

    rng = np.random.default_rng(n + post_n)
    boxes = torch.from_numpy(_rand_boxes(rng, n)).to(device)
    scores = torch.from_numpy(np.sort(rng.uniform(0, 1, n).astype(np.float32))[::-1].copy()).to(device)
    keep, count = _C.nms_with_count(boxes, None, 0.7, max_keep=post_n)
    G = 12
    gt = torch.from_numpy(_rand_boxes(rng, G)).to(device)
    gt_labels = torch.from_numpy(rng.integers(1, 9, G).astype(np.int64)).to(device)
    pend = PendingProposals(boxes, scores, keep, count, post_n, (900, 600))
    if with_gt:
        pend.attach_ground_truth(BoxList(gt, (900, 600)))
    cap, max_pos, seed = 256, 64, 4242 + n
    counts_a = torch.zeros(2, dtype=torch.int32, device=device)
    out, prop_boxes, prop_scores, n_props = _C.proposals_sample(
        pend.pending, gt if is_source else None, gt_labels if is_source else None, 0.5, 0.5, (10.0, 10.0, 5.0, 5.0), cap,
        max_pos, seed, is_source, counts_a)
    # the host chain
    k = min(int(count), post_n)
    kept = keep[:k]
    pb, ps = boxes[kept], scores[kept]
    if with_gt:
        pb, ps = torch.cat([pb, gt]), torch.cat([ps, torch.ones(G, device=device)])
    lab = reg = None
    if is_source:
        _, lab, reg = _C.box_match_encode(pb, gt, gt_labels, 0.5, 0.5, (10.0, 10.0, 5.0, 5.0))
    counts_b = torch.zeros(2, dtype=torch.int32, device=device)
    ref = _C.sample_rois(pb, lab, reg, cap, max_pos, seed, is_source, counts_b)
    assert int(n_props) == pb.shape[0]
    assert torch.equal(prop_boxes[: pb.shape[0]], pb) and torch.equal(prop_scores[: pb.shape[0]], ps)
    assert counts_a.tolist() == counts_b.tolist()
    rows = counts_b.tolist()[0]
    assert rows == min(cap, pb.shape[0]) or is_source        # (source: ignored rows may leave the sample short)
    for name in ("idx", "boxes", "labels", "regression_targets", "loss_labels", "domain"):
        assert torch.equal(out[name], ref[name]), name
    assert torch.equal(out["objectness"][:rows], ps[ref["idx"][:rows]])
    # materialisation = the same list
    assert pend.pending is not None and pend.upper_bound() == post_n + (G if with_gt else 0)
    assert len(pend) == pb.shape[0] and pend.pending is None
    assert torch.equal(pend.bbox, pb) and torch.equal(pend.get_field("objectness"), ps)


def test_nms_presorted_equals_ranked(device):
    """scores=None: the caller's order is the ranking (RPN top-k output) — same kept set as the ranked call"""
    from da_detect_amd import _C

    rng = np.random.default_rng(11)
    boxes = torch.from_numpy(_rand_boxes(rng, 5000)).to(device)
    scores = torch.from_numpy(np.sort(rng.uniform(0, 1, 5000).astype(np.float32))[::-1].copy()).to(device)
    for max_keep in (-1, 300):
        k0, c0 = _C.nms_with_count(boxes, scores, 0.7, max_keep=max_keep)
        k1, c1 = _C.nms_with_count(boxes, None, 0.7, max_keep=max_keep)
        assert int(c0) == int(c1) and torch.equal(k0[: int(c0)], k1[: int(c1)])


@pytest.mark.parametrize("A,G", [(20000, 14), (3000, 1), (122880, 20)])
def test_rpn_anchor_targets_match_the_aten_chain(device, A, G):
    """two-launch anchor labelling vs the reference-order chain on the CPU (boxlist_iou, Matcher with low-quality
    matches, rpn/loss.py:78-96 label rules, BoxCoder(1,1,1,1).encode) — labels bit-exact, targets to 1e-5"""
    from da_detect_amd import _C
    from oracle import model_ref as M

    rng = np.random.default_rng(A + G)
    anchors = torch.from_numpy(_rand_boxes(rng, A, max_side=600))
    anchors[:, :2] -= 30.0                                # some anchors stick out of the image
    gts = torch.from_numpy(_rand_boxes(rng, G, max_side=500))
    anchors[5] = gts[0]                                   # one exact match
    vis = (anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] < 2048) & (anchors[:, 3] < 1024)
    iou = M.box_iou(gts, anchors)
    m = M.matcher(iou, 0.7, 0.3, True)
    want = (m >= 0).float()
    want[m == M.BELOW_LOW] = 0
    want[~vis] = -1
    want[m == M.BETWEEN] = -1
    want_reg = M.encode(gts[m.clamp(min=0)], anchors, (1.0, 1.0, 1.0, 1.0))
    lab, reg = _C.rpn_anchor_targets(anchors.to(device), vis.to(device), gts.to(device), 0.7, 0.3)
    assert torch.equal(lab.cpu(), want) and int((want == 1).sum()) >= G
    torch.testing.assert_close(reg.cpu(), want_reg, rtol=1e-5, atol=1e-6)


def test_roi_align_workspace_variants_are_bit_identical_at_full_size(device, monkeypatch):
    """dadet_roi_align_forward_ws (ROIs processed in Z-order of their centres) against the plain entry point at the
    BASELINE shape: 512 ROIs on a [2, 1024, 64, 128] map — same bits"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(9)
    R = 512
    xy = torch.rand((R, 2), generator=g) * torch.tensor([1800.0, 900.0])
    wh = torch.rand((R, 2), generator=g) ** 2 * torch.tensor([900.0, 600.0]) + 4
    rois = torch.cat([(torch.arange(R) >= 256).float().view(-1, 1), xy, torch.minimum(xy + wh, torch.tensor([2047.0, 1023.0]))],
                     1).to(device)
    rois[5] = torch.tensor([0.0, -40.0, -30.0, 2100.0, 1100.0])     # whole image and beyond
    rois[300] = torch.tensor([1.0, 100.0, 100.0, 100.5, 100.5])     # degenerate
    x = torch.randn((2, 1024, 64, 128), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    go = torch.randn((R, 1024, 14, 14), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    out, gin = [], []
    for flag in (False, True):
        monkeypatch.setattr(_C, "ROI_ALIGN_WORKSPACE", flag)
        out.append(_C.roi_align_forward(x, rois, 1 / 16.0, 14, 14, 0))
        gin.append(_C.roi_align_backward(go, rois, 1 / 16.0, 14, 14, 2, 1024, 64, 128, 0))
    assert torch.equal(out[0], out[1]), "spatially ordered forward differs from the plain one"
    assert torch.equal(gin[0], gin[1])
    # interleaved batch indices and R not a multiple of 32
    rois2 = rois[torch.randperm(R, generator=g)[:333].to(device)].contiguous()
    a, b = [], []
    for flag in (False, True):
        monkeypatch.setattr(_C, "ROI_ALIGN_WORKSPACE", flag)
        a.append(_C.roi_align_forward(x, rois2, 1 / 16.0, 7, 7, 2))
        b.append(_C.roi_align_backward(go[:333, :, :7, :7].contiguous(memory_format=torch.channels_last), rois2, 1 / 16.0,
                                       7, 7, 2, 1024, 64, 128, 2))
    assert torch.equal(a[0], a[1]) and torch.equal(b[0], b[1])


@pytest.mark.parametrize("shape", [
    # N, Cin, H, W, Cout, k, pad: tiles of 128 x 128 / K-tiles
    (2, 256, 40, 64, 256, 3, 1),        # 80 tiles, 72 K-tiles: every tile is cut into 6 - 7 parts
    (2, 256, 40, 60, 200, 3, 1),        # ragged M (4800 rows) and Cout
    (1, 512, 7, 7 * 96, 512, 3, 1),     # 37 x 4 = 148 tiles, 144 K-tiles (the res5 3x3 shape, fewer ROIs)
    (2, 1024, 52, 128, 1024, 1, 0),     # 104 x 8 = 832 tiles: 512 one per workgroup + 320 stream-K, 32 K-tiles
    (1, 512, 7, 7 * 256, 512, 3, 1),    # 98 x 4 = 392 tiles (res5 3x3 over 256 ROIs): between one and two workgroups per
                                        # CU, every tile a stream-K tile (DADET_STREAMK_SMALL: default in mode 3 only)
])
def test_stream_k_tail_matches_the_plain_grid(device, shape, monkeypatch):
    """conv_fwd_split_sk_kernel (stream-K tail: partial tiles parked in a workspace, the last arriver sums the parts in
    part order and runs the epilogue) against the one-tile-per-workgroup grid of the same kernel body: identical up to the
    association of the K sum (1e-5 relative), with every epilogue (FrozenBN scale / bias / residual / ReLU, and the
    dgrad gate), and bit-reproducible run to run"""
    import os

    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    CL = torch.channels_last
    x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(device).contiguous(memory_format=CL)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(device)
    bias = torch.randn(Cout, generator=g).to(device)
    addend = torch.randn((N, Cout, H, W), generator=g).to(device).contiguous(memory_format=CL)
    mask = torch.randn((N, Cout, H, W), generator=g).to(device).contiguous(memory_format=CL)
    cases = [dict(), dict(scale=scale, bias=bias, relu_mode=1), dict(scale=scale, bias=bias, addend=addend, relu_mode=1),
             dict(addend=addend, mask_ref=mask, relu_mode=2)]
    out = {}
    monkeypatch.setenv("DADET_STREAMK_SMALL", "1")
    for flag in ("0", "1"):
        monkeypatch.setenv("DADET_STREAMK", flag)
        os.environ["DADET_STREAMK"] = flag
        out[flag] = [_C.conv_forward(x, w, pad=pad, **kw) for kw in cases]
        if flag == "1":
            again = [_C.conv_forward(x, w, pad=pad, **kw) for kw in cases]
            for a, b in zip(out[flag], again):
                assert torch.equal(a, b), "stream-K result is not reproducible"
    for a, b, kw in zip(out["0"], out["1"], cases):
        scale_ = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale_, (sorted(kw), float((a - b).abs().max()), scale_)
    ref = F.conv2d(x.cpu(), w.cpu(), padding=pad)
    torch.testing.assert_close(out["1"][0].cpu(), ref, rtol=1e-4, atol=1e-4)


def test_hip_ops_against_the_reference_build_itself(device):
    """The reference's OWN compiled CPU operators (oracle/_ref/ref_C.so, built from maskrcnn_benchmark/csrc/{vision.cpp,
    cpu/*.cpp} by oracle/build_ref.py; the binary travels to the GPU box, the sources do not) called on the same inputs
    as the HIP operators: NMS kept indices identical (CPU tie rule), ROIAlign forward bit-exact — no restatement in
    between."""
    from da_detect_amd import _C
    from oracle import build_ref

    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref/ref_C.so was not built (needs /root/reference at build time)")
    g = torch.Generator().manual_seed(21)
    for n in (1, 37, 2000, 12000):
        xy = torch.rand((n, 2), generator=g) * torch.tensor([1900.0, 950.0])
        boxes = torch.cat([xy, xy + torch.rand((n, 2), generator=g) * 250 + 2], 1)
        scores = torch.rand(n, generator=g)
        want = ref.nms(boxes, scores, 0.7)
        keep, cnt = _C.nms_with_count(boxes.to(device), scores.to(device), 0.7, tie_rule=0)
        assert torch.equal(keep[: int(cnt)].cpu(), want), "n=%d: NMS indices differ from the reference build" % n
    for (C, H, W, R, ph, sr) in [(32, 24, 40, 50, 7, 2), (64, 64, 128, 128, 14, 0), (8, 13, 21, 20, 14, 0)]:
        x = torch.randn((2, C, H, W), generator=g)
        xy = torch.rand((R, 2), generator=g) * torch.tensor([W * 14.0, H * 14.0])
        rois = torch.cat([torch.randint(0, 2, (R, 1), generator=g).float(), xy,
                          xy + torch.rand((R, 2), generator=g) ** 2 * 400 + 1], 1)
        want = ref.roi_align_forward(x, rois, 1 / 16.0, ph, ph, sr)
        got = _C.roi_align_forward(x.to(device), rois.to(device), 1 / 16.0, ph, ph, sr)
        assert torch.equal(got.cpu(), want), "ROIAlign forward differs from the reference build"


@pytest.mark.parametrize("shape", [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 64, 96, 256, 1, 1, 0),        # 64x64 tiles (short K), res2-like
    (2, 128, 40, 60, 512, 1, 1, 0),       # ragged M
    (2, 256, 40, 64, 256, 3, 1, 1),       # 128x128 tiles
    (3, 96, 33, 47, 200, 3, 2, 1),        # stride 2, Cout % 128 != 0, 128x64 tiles
    (512, 512, 7, 7, 512, 3, 1, 1),       # stream-K launch form
])
def test_vectorised_epilogue_is_bit_identical(device, shape):
    """conv_epilogue_v4 (16 bytes per lane through an LDS transpose) against the 4-byte epilogue: every combination of
    FrozenBN scale / bias, residual addend, ReLU and the data-gradient gate — identical bits"""
    import os

    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    CL = torch.channels_last
    x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * 0.05).to(device).contiguous(memory_format=CL)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    scale = (torch.rand(Cout, generator=g) + 0.5).to(device)
    bias = torch.randn(Cout, generator=g).to(device)
    addend = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
    mask = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
    cases = [dict(), dict(bias=bias), dict(scale=scale, bias=bias, relu_mode=1),
             dict(scale=scale, bias=bias, addend=addend, relu_mode=1), dict(addend=addend, mask_ref=mask, relu_mode=2),
             dict(mask_ref=mask, relu_mode=2)]
    out = {}
    from da_detect_amd import _lib
    lib = _lib.load()
    big = lib.dadet_get_big_gemm()
    lib.dadet_set_big_gemm(0)          # the two epilogues of the 128 x 128 kernel (the 256 x 256 one has the 16-byte form only)
    try:
        for flag in ("0", "1"):
            os.environ["DADET_EPILOGUE_V4"] = flag
            try:
                out[flag] = [_C.conv_forward(x, w, stride=stride, pad=pad, **kw) for kw in cases]
            finally:
                os.environ.pop("DADET_EPILOGUE_V4")
    finally:
        lib.dadet_set_big_gemm(big)
    for a, b, kw in zip(out["0"], out["1"], cases):
        assert torch.equal(a, b), sorted(kw)


@pytest.mark.parametrize("sampling_ratio", [0, 2])
def test_roi_align_every_other_bin(device, sampling_ratio):
    """dadet_roi_align_forward_sub / _backward_sub (bin_stride 2): the compact 7 x 7 result IS the (2i, 2j) sub-grid of
    the full 14 x 14 one, bit for bit; the gradient equals the full backward fed with zeros in the other bins"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(77 + sampling_ratio)
    B, C, H, W, R = 2, 256, 38, 50, 300
    feat = torch.randn((B, C, H, W), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    xy = torch.rand((R, 2), generator=g) * torch.tensor([W * 16.0 - 40, H * 16.0 - 40])
    wh = torch.rand((R, 2), generator=g) * 400 + 4
    rois = torch.cat([torch.randint(0, B, (R, 1), generator=g).float(), xy, xy + wh], dim=1).to(device)
    rois[:5, 1:] = torch.tensor([-30.0, -20.0, 10.0, 12.0])       # partly outside the image
    full = _C.roi_align_forward(feat, rois, 1 / 16, 14, 14, sampling_ratio)
    sub = _C.roi_align_forward(feat, rois, 1 / 16, 14, 14, sampling_ratio, bin_stride=2)
    assert tuple(sub.shape) == (R, C, 7, 7)
    assert torch.equal(sub, full[:, :, ::2, ::2])
    go = torch.randn((R, C, 7, 7), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    spread = torch.zeros((R, C, 14, 14), device=device).contiguous(memory_format=torch.channels_last)
    spread[:, :, ::2, ::2] = go
    want = _C.roi_align_backward(spread, rois, 1 / 16, 14, 14, B, C, H, W, sampling_ratio)
    got = _C.roi_align_backward(go, rois, 1 / 16, 14, 14, B, C, H, W, sampling_ratio, bin_stride=2)
    assert torch.equal(got, want)
    # odd grids: ceil(5 / 2) = 3 bins per side
    full5 = _C.roi_align_forward(feat, rois, 1 / 16, 5, 5, sampling_ratio)
    assert torch.equal(_C.roi_align_forward(feat, rois, 1 / 16, 5, 5, sampling_ratio, bin_stride=2), full5[:, :, ::2, ::2])


def test_res5_head_on_the_sub_grid_is_the_same_head(device, monkeypatch):
    """ResNet50Conv5ROIFeatureExtractor: pooling only the bins its stride-2 1x1 convolutions read (default) against the
    reference's full 14 x 14 grid (DADET_ROI_SUBGRID=0): identical features, identical gradients"""
    from da_detect_amd.config import cfg as base
    from da_detect_amd.modeling.roi_heads.box_head import roi_box_feature_extractors as fe
    from da_detect_amd.structures.bounding_box import BoxList

    c = base.clone()
    c.merge_from_file(os.path.join(ROOT, "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_img_only.yaml"))
    torch.manual_seed(3)
    ext = fe.make_roi_box_feature_extractor(c).to(device)
    g = torch.Generator().manual_seed(5)
    feat = torch.randn((2, 1024, 24, 40), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    boxes = []
    for _ in range(2):
        xy = torch.rand((40, 2), generator=g) * torch.tensor([500.0, 300.0])
        boxes.append(BoxList(torch.cat([xy, xy + torch.rand((40, 2), generator=g) * 200 + 8], dim=1).to(device),
                             (640, 384), "xyxy"))
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(fe, "_SUBGRID", flag)
        f = feat.clone().requires_grad_(True)
        for p in ext.parameters():
            p.grad = None
        y = ext([f], boxes)
        (y * torch.linspace(-1, 1, y.numel(), device=device).view_as(y)).sum().backward()
        out[flag] = (y.detach(), f.grad, {n: p.grad.clone() for n, p in ext.named_parameters() if p.grad is not None})
    assert tuple(out[True][0].shape) == (80, 2048, 7, 7)
    assert torch.equal(out[True][0], out[False][0])
    torch.testing.assert_close(out[True][1], out[False][1], rtol=1e-5, atol=1e-6)
    for n, gfull in out[False][2].items():
        torch.testing.assert_close(out[True][2][n], gfull, rtol=1e-4, atol=1e-6, msg=lambda m, n=n: "%s: %s" % (n, m))


def test_batched_wgrad_reduction_is_bit_identical(device):
    """dadet_conv_wgrad_partials + ONE dadet_conv_wgrad_reduce_batch for several weight gradients against
    dadet_conv_wgrad's own reduction pass per tensor: identical bits, with / without FrozenBN scale and accumulation"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(9)
    shapes = [(2, 256, 40, 64, 256, 3, 1, 1), (2, 256, 40, 64, 1024, 1, 1, 0), (2, 1024, 40, 64, 256, 1, 1, 0),
              (2, 512, 80, 128, 256, 1, 2, 0), (64, 512, 7, 7, 512, 3, 1, 1)]
    batch = _C.WgradBatch()
    want, got = [], []
    for i, (N, Cin, H, W, Cout, k, stride, pad) in enumerate(shapes):
        x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        gy = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
        scale = (torch.rand(Cout, generator=g) + 0.5).to(device) if i % 2 == 0 else None
        base = torch.randn((Cout, Cin, k, k), generator=g).to(device).contiguous(memory_format=CL)
        acc = i % 3 == 0
        want.append(_C.conv_wgrad(x, gy, (Cout, Cin, k, k), stride, pad, out_scale=scale,
                                  dw=base.clone(memory_format=torch.preserve_format), accumulate=acc))
        got.append(_C.conv_wgrad(x, gy, (Cout, Cin, k, k), stride, pad, out_scale=scale,
                                 dw=base.clone(memory_format=torch.preserve_format), accumulate=acc, pending=batch))
    assert len(batch) >= 3, "the plan split fewer reductions than this test means to batch: %d" % len(batch)
    _C.conv_wgrad_reduce_batch(batch)
    assert len(batch) == 0
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_roi_align_backward_skips_images_without_rois(device):
    """live_images: the gather backward sweeps the leading images only and zero-fills the rest — same result as the full
    sweep when no ROI points at the other images"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(123)
    B, C, H, W, R = 3, 128, 30, 44, 90
    xy = torch.rand((R, 2), generator=g) * torch.tensor([W * 16.0 - 60, H * 16.0 - 60])
    rois = torch.cat([torch.zeros((R, 1)), xy, xy + torch.rand((R, 2), generator=g) * 300 + 6], dim=1).to(device)
    go = torch.randn((R, C, 7, 7), generator=g).to(device).contiguous(memory_format=torch.channels_last)
    for stride in (1, 2):
        gg = go if stride == 2 else torch.randn((R, C, 14, 14), generator=g).to(device).contiguous(memory_format=CL)
        full = _C.roi_align_backward(gg, rois, 1 / 16, 14, 14, B, C, H, W, 0, bin_stride=stride)
        lean = _C.roi_align_backward(gg, rois, 1 / 16, 14, 14, B, C, H, W, 0, bin_stride=stride, live_images=1)
        assert torch.equal(full, lean) and float(full[1:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [
    # N, Cin(K), H, W, Cout, stride, epilogue
    (2, 256, 64, 128, 1024, 1, "add_relu"),     # res4 conv3 forward (BASELINE size): K = 256, BN = 128 (halved epilogue slices)
    (2, 256, 64, 128, 1024, 1, "add_gate"),     # res4 conv1 data gradient: residual gradient + ReLU gate
    (2, 128, 128, 256, 512, 1, "add_relu"),     # res3 conv3: K = 128, BN = 128
    (2, 64, 256, 256, 256, 1, "add_relu"),      # res2 conv3: K = 64 (two slabs per loop iteration)
    (2, 64, 200, 164, 256, 1, "plain"),         # M = 65600: ragged last slab, odd slab count per workgroup
    (1, 256, 259, 262, 160, 2, "affine"),       # stride 2 (projection shortcut), Cout not a multiple of the panel
    (3, 128, 96, 100, 384, 1, "gate"),          # M = 28800, three panels
    (2, 256, 90, 100, 288, 1, "add_gate"),      # K = 256 with 128-column panels (Cout >= 256), ragged third panel, M = 18000
    (2, 256, 64, 128, 2304, 1, "plain"),        # the DCN blocks' data gradient: 18 panels, cut into two launches of 9
    (1, 128, 128, 128, 1184, 1, "affine"),      # 10 panels (the last ragged): 2 x 5
])
def test_weight_stationary_1x1_kernel(device, case):
    """conv1x1_ws_kernel (csrc/conv_ws.hip) against a float64 contraction and against the tiled split kernel it replaces
    on the same inputs: same products, another summation order -> fp32 rounding apart; error vs float64 not larger than the
    tiled kernel's (both within the bound of the accuracy test above)."""
    import os

    from da_detect_amd import _C

    N, K, H, W, Cout, stride, epi = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn((N, K, H, W), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((Cout, K, 1, 1), generator=g) * (2.0 / K) ** 0.5).to(device).contiguous(memory_format=CL)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    kw = {}
    if epi in ("add_relu", "affine", "add_gate"):
        kw["scale"] = (torch.rand(Cout, generator=g) + 0.5).to(device)
        kw["bias"] = torch.randn(Cout, generator=g).to(device)
    if epi in ("add_relu", "add_gate"):
        kw["addend"] = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
    if epi == "add_relu":
        kw["relu_mode"] = 1
    if epi in ("add_gate", "gate"):
        kw["relu_mode"] = 2
        kw["mask_ref"] = torch.randn((N, Cout, Ho, Wo), generator=g).clamp_min(0).to(device).contiguous(memory_format=CL)
    d = _C._desc(N, H, W, K, Cout, 1, 1, stride, 0, Ho, Wo)
    import ctypes
    from da_detect_amd import _lib
    assert _lib.load().dadet_conv_forward_variant(ctypes.byref(d)) == 3, "the case must take the weight-stationary kernel"
    out = {}
    for flag in ("0", "1"):
        os.environ["DADET_WS_1X1"] = flag
        try:
            out[flag] = _C.conv_forward(x, w, stride=stride, **kw)
            again = _C.conv_forward(x, w, stride=stride, **kw)
        finally:
            os.environ.pop("DADET_WS_1X1")
        assert torch.equal(out[flag], again)
    # float64 reference on the device
    xs = x[:, :, ::stride, ::stride].double().permute(0, 2, 3, 1).reshape(-1, K)
    ref = xs @ w.double().reshape(Cout, K).t()
    if "scale" in kw:
        ref = ref * kw["scale"].double() + kw["bias"].double()
    if "addend" in kw:
        ref = ref + kw["addend"].double().permute(0, 2, 3, 1).reshape(-1, Cout)
    if kw.get("relu_mode") == 1:
        ref = ref.clamp_min(0)
    if kw.get("relu_mode") == 2:
        ref = ref * (kw["mask_ref"].permute(0, 2, 3, 1).reshape(-1, Cout) > 0)
    errs = {}
    for flag in ("0", "1"):
        got = out[flag].permute(0, 2, 3, 1).reshape(-1, Cout).double()
        e = (got - ref).abs()
        errs[flag] = (float(e.pow(2).mean().sqrt()), float(e.max()))
    scale_ = float(ref.abs().mean()) + 1e-30
    assert errs["1"][1] <= 2e-5 * max(1.0, float(ref.abs().max())), errs
    assert errs["1"][0] <= 1.25 * errs["0"][0] + 1e-9 * scale_, errs
    # ReLU at an exact tie may fire on one side only; everything else agrees to fp32 rounding
    torch.testing.assert_close(out["1"], out["0"], rtol=2e-5, atol=2e-5 * max(1.0, float(ref.abs().max())))


def test_pyramid_roi_align_equals_the_per_level_split(device):
    """Pooler over four pyramid levels: one launch per level over all ROIs with the level filter in the kernel
    (dadet_roi_align_forward_level / _backward_level, no host round trip) against the reference's structure — per level
    nonzero, gather, ROIAlign, index_put (poolers.py:108-121): identical features, identical map gradients"""
    from da_detect_amd.modeling import poolers as P
    from da_detect_amd.structures.bounding_box import BoxList

    rng = np.random.default_rng(11)
    scales = (0.25, 0.125, 0.0625, 0.03125)
    H0, W0 = 640, 1024
    feats = [torch.from_numpy(rng.standard_normal((2, 64, int(H0 * s), int(W0 * s))).astype(np.float32)).to(device)
             .contiguous(memory_format=CL).requires_grad_(True) for s in scales]
    boxes = []
    for n in (150, 97):
        xy = rng.uniform(0, 300, (n, 2))
        wh = np.exp(rng.uniform(np.log(8), np.log(900), (n, 2)))          # every level is hit
        b = np.concatenate([xy, np.minimum(xy + wh, [W0 - 1, H0 - 1])], 1).astype(np.float32)
        boxes.append(BoxList(torch.from_numpy(b).to(device), (W0, H0), mode="xyxy"))
    pooler = P.Pooler((7, 7), scales, 2).to(device)
    levels = pooler.map_levels(boxes)
    assert sorted(int(v) for v in levels.unique().tolist()) == [0, 1, 2, 3]
    gy = torch.from_numpy(rng.standard_normal((247, 64, 7, 7)).astype(np.float32)).to(device)
    res = {}
    for flag in (False, True):
        P._PYRAMID_KERNELS = flag
        try:
            out = pooler(feats, boxes)
            grads = torch.autograd.grad(out, feats, gy)
        finally:
            P._PYRAMID_KERNELS = True
        res[flag] = (out.detach(), grads)
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
