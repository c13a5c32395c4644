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
                                 (2, 8, 7, 7, 8, 1, 1, 1, True),
                                 # LDS-window backward (stride 1, C % 64 == 0): ragged tiles, 2 groups, and 4 groups of
                                 # 16 channels (lanes of one chunk in different deformable groups)
                                 (2, 64, 20, 19, 16, 3, 1, 1, True), (1, 128, 9, 17, 8, 3, 1, 2, False),
                                 (1, 64, 10, 10, 8, 3, 1, 4, True)])
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


# ---------------------------------------------------------------------------------------------------------------
# deformable PSROI pooling (deform_pool_func.py / deform_pool_kernel_cuda.cu)
def _psroi_case(seed, no_trans, gs, ncls=1, out_dim=4, P=3, part=3, spp=2, B=2, H=10, W=12, R=5, trans_std=0.1):
    g = torch.Generator().manual_seed(seed)
    data = torch.randn(B, out_dim * gs * gs, H, W, generator=g)
    x1 = torch.rand(R, generator=g) * (W * 8 - 30) - 6      # some ROIs start left of / above the map
    y1 = torch.rand(R, generator=g) * (H * 8 - 30) - 6
    bw = torch.rand(R, generator=g) * 50 + 4
    bh = torch.rand(R, generator=g) * 50 + 4
    rois = torch.stack([torch.randint(0, B, (R,), generator=g).float(), x1, y1, x1 + bw, y1 + bh], 1)
    trans = torch.randn(R, 2 * ncls, part, part, generator=g)
    grad = torch.randn(R, out_dim, P, P, generator=g)
    return data, rois, trans, grad, dict(spatial_scale=0.125, out_size=P, out_dim=out_dim, no_trans=no_trans,
                                         group_size=gs, part_size=part, sample_per_part=spp, trans_std=trans_std)


@pytest.mark.gpu
@pytest.mark.parametrize("no_trans,gs,ncls,out_dim", [(True, 1, 1, 4), (False, 1, 1, 4), (False, 3, 2, 4),
                                                      (True, 3, 1, 6), (False, 1, 1, 72)])
def test_deform_psroi_pool_matches_oracle(no_trans, gs, ncls, out_dim):
    from da_detect_amd.layers.dcn import deform_roi_pooling
    from oracle.deform_ref import deform_psroi_pool
    data, rois, trans, grad, kw = _psroi_case(7 + gs + ncls, no_trans, gs, ncls, out_dim)
    ref_out, ref_cnt, ref_gd, ref_gt = deform_psroi_pool(data, rois, trans, grad_out=grad, **kw)
    d = data.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    t = trans.cuda().requires_grad_(not no_trans)
    out = deform_roi_pooling(d, rois.cuda(), t, kw["spatial_scale"], kw["out_size"], kw["out_dim"], no_trans,
                             gs, kw["part_size"], kw["sample_per_part"], kw["trans_std"])
    out.backward(grad.cuda())
    assert (ref_cnt > 0).any() and (ref_cnt < kw["sample_per_part"] ** 2).any()   # partially-outside bins are covered
    torch.testing.assert_close(out.cpu(), ref_out, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d.grad.cpu(), ref_gd, rtol=1e-4, atol=1e-5)
    if not no_trans:
        torch.testing.assert_close(t.grad.cpu(), ref_gt, rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
def test_deform_psroi_pool_identities_and_packs():
    from da_detect_amd.layers.dcn import (DeformRoIPooling, DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack,
                                          deform_roi_pooling)
    data, rois, trans, grad, kw = _psroi_case(3, False, 1, out_dim=8)
    d, r, t = data.cuda(), rois.cuda(), trans.cuda()
    args = (kw["spatial_scale"], kw["out_size"], 8)
    plain = deform_roi_pooling(d, r, d.new_empty(0), *args, True, 1, 3, 2, 0.1)
    # zero offsets, or trans_std = 0, reduce the deformable pooling to the undeformed one
    torch.testing.assert_close(deform_roi_pooling(d, r, torch.zeros_like(t), *args, False, 1, 3, 2, 0.1), plain)
    torch.testing.assert_close(deform_roi_pooling(d, r, t, *args, False, 1, 3, 2, 0.0), plain)
    # a constant map pools to the constant wherever at least one sample is inside
    const = torch.full_like(d, 2.5)
    out = deform_roi_pooling(const, r, t, *args, False, 1, 3, 2, 0.1)
    assert torch.all((out == 0) | ((out - 2.5).abs() < 1e-5))
    torch.testing.assert_close(DeformRoIPooling(*args, no_trans=True, sample_per_part=2)(d, r, t), plain)
    # the packs initialise their last layers to zero (deform_pool_module.py:63-64, :125-126): v1 == plain,
    # v2 == plain * sigmoid(0)
    torch.manual_seed(0)
    pack = DeformRoIPoolingPack(*args, no_trans=False, trans_std=0.1, sample_per_part=2, deform_fc_channels=64).cuda()
    torch.testing.assert_close(pack(d, r), plain)
    mpack = ModulatedDeformRoIPoolingPack(*args, no_trans=False, trans_std=0.1, sample_per_part=2,
                                          deform_fc_channels=64).cuda()
    torch.testing.assert_close(mpack(d, r), plain * 0.5)
    # gradients reach the offset branch once its last layer is non-zero
    torch.nn.init.normal_(pack.offset_fc[-1].weight, std=0.01)
    dd = d.clone().requires_grad_(True)
    pack(dd, r).square().sum().backward()
    assert dd.grad.abs().sum() > 0 and pack.offset_fc[0].weight.grad.abs().sum() > 0
    # zero ROIs
    e = deform_roi_pooling(d, r[:0], d.new_empty(0), *args, True, 1, 3, 2, 0.1)
    assert e.shape == (0, 8, 3, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("modulated", [False, True])
def test_dcn_bottleneck_stage_matches_plain_stage_at_zero_offsets(modulated):
    """STAGE_WITH_DCN wiring (vendored modeling/backbone/resnet.py:110-124,286-300): with the offset conv zeroed a
    DFConv2d bottleneck equals the plain one (x0.5 through sigmoid(0) when modulated) — forward and gradients —
    also next to fused plain blocks inside one stage (the `out_private` promise must not be made to a DCN block)."""
    from da_detect_amd.modeling.backbone.resnet import BottleneckWithFixedBatchNorm, _Stage

    torch.manual_seed(0)
    dcn = {"stage_with_dcn": True, "with_modulated_dcn": modulated, "deformable_groups": 1}

    def make(cfgs):
        return _Stage(*[BottleneckWithFixedBatchNorm(64 if i == 0 else 128, 32, 128, stride=1, dcn_config=c)
                        for i, c in enumerate(cfgs)]).cuda()

    plain, mixed = make([None, None, None]), make([None, dcn, None])
    sd = plain.state_dict()
    msd = mixed.state_dict()
    for k, v in sd.items():
        if k.startswith("1.conv2."):
            msd["1.conv2.conv.weight"] = v * (2.0 if modulated else 1.0)   # undo the sigmoid(0) = 0.5 modulation
        else:
            msd[k] = v
    msd["1.conv2.offset.weight"].zero_()
    msd["1.conv2.offset.bias"].zero_()
    mixed.load_state_dict(msd)
    x = torch.randn(2, 64, 12, 16, device="cuda").contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = plain(xa), mixed(xb)
    torch.testing.assert_close(yb, ya, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    torch.testing.assert_close(xb.grad, xa.grad, rtol=1e-3, atol=1e-4)
    pa, pb = dict(plain.named_parameters()), dict(mixed.named_parameters())
    for k in ("0.conv1.weight", "2.conv3.weight", "1.conv1.weight"):
        torch.testing.assert_close(pb[k].grad, pa[k].grad, rtol=1e-3, atol=1e-4)
    scale = 0.5 if modulated else 1.0
    torch.testing.assert_close(pb["1.conv2.conv.weight"].grad, pa["1.conv2.weight"].grad * scale, rtol=1e-3, atol=1e-4)
