"""Deformable convolution: HIP kernels vs the torch restatement (oracle/deform_ref.py) and vs identities."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

CL = torch.channels_last


def _case(seed, N, C, H, W, Cout, k, stride, dg, modulated, scale=2.0):
    g = torch.Generator().manual_seed(seed)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn((N, C, H, W), generator=g)
    off = torch.randn((N, dg * 2 * k * k, Ho, Wo), generator=g) * scale
    mask = torch.rand((N, dg * k * k, Ho, Wo), generator=g) if modulated else None
    w = torch.randn((Cout, C, k, k), generator=g) / (C * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) if modulated else None
    return x, off, mask, w, b, pad


def test_oracle_identities_cpu():
    """the torch restatement against F.conv2d (zero offsets), v2 == v1 for unit masks, integer shifts"""
    from oracle import deform_ref as R

    x, off, mask, w, b, pad = _case(0, 2, 8, 9, 11, 12, 3, 1, 1, True)
    y0 = R.deform_conv2d(x, torch.zeros_like(off), torch.ones_like(mask), w, b, 1, pad, 1, 1)
    torch.testing.assert_close(y0, F.conv2d(x, w, b, 1, pad), rtol=1e-5, atol=1e-5)
    y1 = R.deform_conv2d(x, off, None, w, None, 1, pad, 1, 1)
    y2 = R.deform_conv2d(x, off, torch.ones_like(mask), w, None, 1, pad, 1, 1)
    torch.testing.assert_close(y1, y2)
    shift = torch.zeros_like(off)
    shift[:, 0::2] = 1.0  # dy = +1 for every tap: equals a conv of the image shifted up by one row (zero padded)
    xs = torch.zeros_like(x)
    xs[:, :, :-1] = x[:, :, 1:]
    # (rows >= 1: at output row 0 the shifted conv sees zero padding where the deformable sample still reads x[0])
    torch.testing.assert_close(R.deform_conv2d(x, shift, None, w, None, 1, pad, 1, 1)[:, :, 1:],
                               F.conv2d(xs, w, None, 1, pad)[:, :, 1:], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 16, 9, 11, 24, 3, 1, 1, False), (2, 32, 12, 10, 16, 3, 1, 1, True),
                                 (1, 64, 15, 13, 32, 3, 2, 2, True), (1, 1024, 6, 7, 8, 3, 1, 4, True),
                                 (2, 8, 7, 7, 8, 1, 1, 1, True)])
def test_deform_conv_matches_oracle(device, cfg):
    from da_detect_amd.layers.dcn import deform_conv, modulated_deform_conv
    from oracle import deform_ref as R

    N, C, H, W, Cout, k, stride, dg, modulated = cfg
    x, off, mask, w, b, pad = _case(sum(cfg[:7]), N, C, H, W, Cout, k, stride, dg, modulated)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, w)] + \
        ([mask.clone().requires_grad_(True), b.clone().requires_grad_(True)] if modulated else [])
    want = R.deform_conv2d(leaves[0], leaves[1], leaves[3] if modulated else None, leaves[2],
                           leaves[4] if modulated else None, stride, pad, 1, dg)
    gy = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    want.backward(gy)
    dl = [t.detach().to(device).contiguous(memory_format=CL) if t.dim() == 4 else t.detach().to(device) for t in leaves]
    for t in dl:
        t.requires_grad_(True)
    if modulated:
        got = modulated_deform_conv(dl[0], dl[1], dl[3], dl[2], dl[4], stride, pad, 1, 1, dg)
    else:
        got = deform_conv(dl[0], dl[1], dl[2], stride, pad, 1, 1, dg)
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-4, atol=1e-4)
    got.backward(gy.to(device))
    for name, a, r in zip(["x", "offset", "weight", "mask", "bias"], dl, leaves):
        scale = float(r.grad.abs().max()) + 1e-12
        err = float((a.grad.cpu() - r.grad).abs().max()) / scale
        assert err < 2e-4, "grad %s: %.3e" % (name, err)


@pytest.mark.gpu
def test_dfconv2d_zero_offsets_equals_conv(device):
    from da_detect_amd.layers.dcn import DFConv2d

    torch.manual_seed(0)
    m = DFConv2d(32, 48, with_modulated_dcn=False, kernel_size=3, stride=1).to(device)
    m.offset.weight.data.zero_()
    x = torch.randn(2, 32, 10, 12)
    y = m(x.to(device).contiguous(memory_format=CL)).cpu()
    torch.testing.assert_close(y, F.conv2d(x, m.conv.weight.detach().cpu(), None, 1, 1), rtol=1e-4, atol=1e-4)
