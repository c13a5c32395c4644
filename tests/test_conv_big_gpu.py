"""conv_big_kernel (csrc/conv_big.hip): the 256 x 256-tile forward / data-gradient kernel of contraction mode 4.

Checked against a float64 convolution and against the 128 x 128 kernel it replaces on the same inputs (same products,
another summation order), over the paths the kernel has: 1x1 and 3x3 taps with padding and stride, ragged M / Cout, every
epilogue, reductions cut into 1 .. 4 parts (the parked-part hand-over), several tiles per launch.
Reference of the operation: ATen conv2d + FrozenBatchNorm2d + relu_ behind mb/modeling/backbone/resnet.py:294-314 and
mb/modeling/rpn/rpn.py:39-46 (mb = maskrcnn_benchmark)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.fixture()
def big_mode():
    from da_detect_amd import _lib

    lib = _lib.load()
    prev = lib.dadet_get_big_gemm()
    yield lib
    lib.dadet_set_big_gemm(prev)


def _ref64(x, w, stride, pad, kw):
    y = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    if "scale" in kw:
        y = y * kw["scale"].double().view(1, -1, 1, 1) + kw["bias"].double().view(1, -1, 1, 1)
    if "addend" in kw:
        y = y + kw["addend"].double()
    if kw.get("relu_mode") == 1:
        y = y.clamp_min(0)
    if kw.get("relu_mode") == 2:
        y = y * (kw["mask_ref"] > 0)
    return y


# N, Cin, H, W, Cout, k, stride, pad, epilogue
CASES = [
    (1, 64, 40, 52, 256, 1, 1, 0, "none"),            # 9 tiles in M (ragged: 2080 rows), K = 64: 2 K-tiles, no split
    (2, 128, 24, 40, 512, 3, 1, 1, "affine_relu"),    # 3x3 with padding, 8 x 2 tiles, K = 1152
    (1, 256, 33, 47, 300, 3, 1, 1, "add_relu"),       # ragged M (1551) and Cout (300): 7 x 2 tiles
    (2, 512, 16, 24, 256, 1, 1, 0, "add_gate"),       # 3 tiles -> K = 512 cut into 2 parts (16 K-tiles)
    (1, 1024, 16, 16, 256, 1, 1, 0, "gate"),          # 1 tile  -> 4 parts of 8 K-tiles
    (1, 256, 20, 20, 512, 3, 2, 1, "affine"),         # stride 2, 3x3: 100 rows, 2 tiles, K = 2304 -> 4 parts
    (3, 96, 14, 14, 260, 3, 1, 1, "add_relu"),        # Cin = 96 (three K-tiles per tap), 588 rows, ragged Cout
    (1, 32, 64, 64, 512, 1, 1, 0, "none"),            # K = 32: one K-tile
]


@pytest.mark.parametrize("tile_n", [256, 128])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_c%d_%dx%d_o%d_k%d_s%d_p%d_%s" % c)
def test_big_tile_kernel_against_float64_and_the_128_tile_kernel(device, big_mode, case, tile_n, monkeypatch):
    from da_detect_amd import _C

    monkeypatch.setenv("DADET_BIG_TILE_N", str(tile_n))   # conv_big_kernel<256> / conv_big128_kernel

    N, Cin, H, W, Cout, k, stride, pad, epi = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(device).contiguous(memory_format=CL)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    kw = {}
    if epi in ("affine", "affine_relu", "add_relu", "add_gate"):
        kw["scale"] = (torch.rand(Cout, generator=g) + 0.5).to(device)
        kw["bias"] = torch.randn(Cout, generator=g).to(device)
    if epi in ("add_relu", "add_gate"):
        kw["addend"] = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
    if epi in ("affine_relu", "add_relu"):
        kw["relu_mode"] = 1
    if epi in ("add_gate", "gate"):
        kw["relu_mode"] = 2
        kw["mask_ref"] = torch.randn((N, Cout, Ho, Wo), generator=g).clamp_min(0).to(device).contiguous(memory_format=CL)
    out = {}
    for mode in (0, 2):
        big_mode.dadet_set_big_gemm(mode)
        d = _C._desc(N, H, W, Cin, Cout, k, k, stride, pad, Ho, Wo)
        variant = big_mode.dadet_conv_forward_variant(ctypes.byref(d))
        assert (variant == (4 if tile_n == 256 else 5)) == (mode == 2), "mode %d must%s take the large-tile kernel (variant %d)" % (
            mode, "" if mode == 2 else " not", variant)
        out[mode] = _C.conv_forward(x, w, stride=stride, pad=pad, **kw)
        again = _C.conv_forward(x, w, stride=stride, pad=pad, **kw)
        assert torch.equal(out[mode], again), "two runs of one launch must agree bit for bit (parts summed in part order)"
    ref = _ref64(x, w, stride, pad, kw)
    errs = {}
    for mode in (0, 2):
        e = (out[mode].double() - ref).abs()
        errs[mode] = (float(e.pow(2).mean().sqrt()), float(e.max()))
    top = max(1.0, float(ref.abs().max()))
    assert errs[2][1] <= 2e-5 * top, errs
    # RMS error: fp32 accumulation error grows with the length of ONE accumulator's reduction, and the two kernels cut K
    # differently (tools/probes/big_parts_error.py: K = 2304 in one part 7.2e-7, in four 3.8e-7, the 128 x 128 kernel's
    # split-K launch 2.3e-7) — same class, not the same number
    assert errs[2][0] <= 4.0 * errs[0][0] + 1e-9 * (float(ref.abs().mean()) + 1e-30), errs
    # ReLU at an exact tie may fire on one side only; everything else agrees to fp32 rounding
    torch.testing.assert_close(out[2], out[0], rtol=2e-5, atol=2e-5 * top)


@pytest.mark.parametrize("tile_n", [256, 128])
@pytest.mark.parametrize("splits", [1, 2, 3, 4])
def test_big_tile_kernel_part_counts_agree(device, big_mode, splits, tile_n, monkeypatch):
    """the same 3x3 layer with its reduction in 1 .. 4 parts: fp32 rounding apart, and run-to-run identical"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 128, 20, 28), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((320, 128, 3, 3), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
    big_mode.dadet_set_big_gemm(2)
    monkeypatch.setenv("DADET_BIG_TILE_N", str(tile_n))
    monkeypatch.setenv("DADET_BIG_SPLITS", "1")
    one = _C.conv_forward(x, w, pad=1)
    monkeypatch.setenv("DADET_BIG_SPLITS", str(splits))
    got = _C.conv_forward(x, w, pad=1)
    for _ in range(3):
        assert torch.equal(got, _C.conv_forward(x, w, pad=1))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    torch.testing.assert_close(got, one, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


def test_big_tile_kernel_leaves_the_output_maximum(device, big_mode):
    """mode 4's hand-over: the epilogue merges max|y| into the caller's slot (the next GEMM's scale)"""
    from da_detect_amd import _C, amax as _amax

    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 256, 24, 24), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((512, 256, 3, 3), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
    big_mode.dadet_set_big_gemm(2)
    y = _C.conv_forward(x, w, pad=1, relu_mode=1)
    assert _amax.value(y) == float(y.abs().max())
