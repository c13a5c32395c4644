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


def _border_case(seed, N, C, H, W, Cout, k, stride, dg, modulated):
    """offsets that put samples at non-integer positions everywhere, inside the border bands (-1, 0) and (H-1, H) /
    (W-1, W) where only one corner row / column exists, exactly ON the band limits (-1, H: the reference returns 0
    there), and far outside the map"""
    x, off, mask, w, b, pad = _case(seed, N, C, H, W, Cout, k, stride, dg, modulated, scale=1.3)
    T = k * k
    Ho, Wo = off.shape[2], off.shape[3]
    g = torch.Generator().manual_seed(seed + 1)
    off = off.double()
    ys = (torch.arange(Ho, dtype=torch.float64) * stride - pad).view(Ho, 1)
    xs = (torch.arange(Wo, dtype=torch.float64) * stride - pad).view(1, Wo)
    for n in range(N):
        for grp in range(dg):
            for tap in range(T):
                i, j = tap // k, tap % k
                kind = int(torch.randint(0, 6, (1,), generator=g))
                cy, cx = grp * 2 * T + 2 * tap, grp * 2 * T + 2 * tap + 1
                frac = torch.rand((Ho, Wo), generator=g, dtype=torch.float64) * 0.98 + 0.01
                if kind == 0:      # rows in (-1, 0)
                    off[n, cy] = -frac - (ys + i)
                elif kind == 1:    # rows in (H-1, H)
                    off[n, cy] = (H - 1) + frac - (ys + i)
                elif kind == 2:    # columns in (W-1, W), rows anywhere
                    off[n, cx] = (W - 1) + frac - (xs + j)
                elif kind == 3:    # exactly on the limits: h = -1 on the upper half, w = W on the lower half
                    off[n, cy, : Ho // 2] = (-1.0 - (ys + i))[: Ho // 2]
                    off[n, cx, Ho // 2:] = (float(W) - (xs + j)).expand(Ho, Wo)[Ho // 2:]
                elif kind == 4:    # far outside
                    off[n, cy] = off[n, cy] + 3.0 * H
                # kind 5: the random non-integer offsets stay
    return x, off.float(), mask, w, b, pad


BORDER_CASES = [(2, 8, 9, 11, 12, 3, 1, 1, False), (2, 16, 10, 12, 8, 3, 1, 2, True), (1, 64, 12, 9, 16, 3, 1, 1, True),
                (1, 16, 11, 13, 8, 3, 2, 1, True), (1, 128, 8, 8, 8, 3, 1, 4, False)]


@pytest.mark.parametrize("cfg", BORDER_CASES)
def test_two_independent_oracles_agree_cpu(cfg):
    """oracle/deform_ref.py (hand-written floor / gather / validity masks) and oracle/deform_ref2.py (grid_sample on a
    normalised grid, per-tap matmul) were written independently from the CUDA kernels: they must agree in float64 on the
    forward and on every gradient, on the border-band / on-the-limit / far-outside / modulated / grouped cases"""
    from oracle import deform_ref as R1
    from oracle import deform_ref2 as R2

    N, C, H, W, Cout, k, stride, dg, modulated = cfg
    x, off, mask, w, b, pad = _border_case(sum(cfg[:7]) + 3, *cfg)
    outs = []
    for R in (R1, R2):
        leaves = [t.double().clone().requires_grad_(True) for t in (x, off, w)] + \
            ([mask.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)] if modulated else [])
        y = R.deform_conv2d(leaves[0], leaves[1], leaves[3] if modulated else None, leaves[2],
                            leaves[4] if modulated else None, stride, pad, 1, dg)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
        y.backward(gy)
        outs.append((y.detach(), [t.grad for t in leaves]))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-10, atol=1e-10)
    for name, a, r in zip(["x", "offset", "weight", "mask", "bias"], outs[0][1], outs[1][1]):
        # on an exact limit / an exact integer position the two formulations may pick different one-sided derivatives
        # of the (there non-differentiable) bilinear kernel for the OFFSET gradient; everything else is smooth
        if name == "offset":
            bad = (a - r).abs() > 1e-8 * (1 + r.abs())
            assert float(bad.double().mean()) < 0.06, "offset gradients differ on %.1f%% of the entries" % (
                100 * float(bad.double().mean()))
        else:
            torch.testing.assert_close(a, r, rtol=1e-9, atol=1e-9, msg="grad " + name)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", BORDER_CASES)
def test_deform_conv_matches_second_oracle_on_border_cases(device, cfg):
    """the HIP kernels against the float64 grid_sample oracle on the same border cases (the offset gradient is compared
    where the two oracles agree with each other, see above)"""
    from da_detect_amd.layers.dcn import deform_conv, modulated_deform_conv
    from oracle import deform_ref as R1
    from oracle import deform_ref2 as R2

    N, C, H, W, Cout, k, stride, dg, modulated = cfg
    x, off, mask, w, b, pad = _border_case(sum(cfg[:7]) + 3, *cfg)
    ref = {}
    for key, R in (("a", R1), ("b", R2)):
        leaves = [t.double().clone().requires_grad_(True) for t in (x, off, w)] + \
            ([mask.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)] if modulated else [])
        y = R.deform_conv2d(leaves[0], leaves[1], leaves[3] if modulated else None, leaves[2],
                            leaves[4] if modulated else None, stride, pad, 1, dg)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
        y.backward(gy)
        ref[key] = (y.detach(), [t.grad for t in leaves])
    src = [x, off, w] + ([mask, b] if modulated else [])
    dl = [t.to(device).contiguous(memory_format=CL) if t.dim() == 4 else t.to(device) for t in src]
    for t in dl:
        t.requires_grad_(True)
    if modulated:
        got = modulated_deform_conv(dl[0], dl[1], dl[3], dl[2], dl[4], stride, pad, 1, 1, dg)
    else:
        got = deform_conv(dl[0], dl[1], dl[2], stride, pad, 1, 1, dg)
    want = ref["b"][0]
    torch.testing.assert_close(got.detach().cpu().double(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
    got.backward(gy.float().to(device))
    for name, a, r1, r2 in zip(["x", "offset", "weight", "mask", "bias"], dl, ref["a"][1], ref["b"][1]):
        g = a.grad.cpu().double()
        scale = float(r2.abs().max()) + 1e-12
        if name == "offset":
            agree = (r1 - r2).abs() <= 1e-8 * (1 + r2.abs())
            err = float(((g - r2).abs() * agree).max()) / scale
            # where the oracles pick different one-sided derivatives the kernel must equal one of them
            side = torch.minimum((g - r1).abs(), (g - r2).abs())
            assert float((side * (~agree)).max()) / scale < 2e-4
        else:
            err = float((g - r2).abs().max()) / scale
        assert err < 2e-4, "grad %s: %.3e" % (name, err)


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


@pytest.mark.gpu
@pytest.mark.parametrize("modulated,stride,dg", [(False, 1, 1), (True, 1, 1), (False, 2, 1), (True, 1, 2)])
def test_fused_dcn_bottleneck_matches_per_conv_path_and_float64_oracle(modulated, stride, dg):
    """the one-node DCN bottleneck (_DCNBottleneckFn: offsets / modulation logits read in place, bn2 + ReLU in the GEMM
    epilogue, y1's ReLU gate and the two gradient paths into y1 merged in the offset conv's data-gradient epilogue)
    against (a) the per-conv path it replaces (DFConv2d module + standalone kernels, DADET_DCN_FUSED=0's route) and
    (b) the block written in float64 torch ops around the grid_sample oracle: output, input gradient, every parameter
    gradient — with NON-zero offsets of a pixel or two."""
    from da_detect_amd.modeling.backbone import resnet as RN
    from oracle import deform_ref2 as R2

    torch.manual_seed(3 + int(modulated) + stride)
    dcn = {"stage_with_dcn": True, "with_modulated_dcn": modulated, "deformable_groups": dg}
    blk = RN.BottleneckWithFixedBatchNorm(64, 32, 128, stride=stride, dcn_config=dcn).cuda()
    with torch.no_grad():
        blk.conv2.offset.weight.normal_(0, 0.08)
        blk.conv2.offset.bias.normal_(0, 0.5)
        for bn in (blk.bn1, blk.bn2, blk.bn3, blk.downsample[1]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
            bn.running_mean.normal_(0, 0.2)
            bn.running_var.uniform_(0.5, 2.0)
            bn._cache = None
    params = dict(blk.named_parameters())
    x = torch.randn(2, 64, 14, 18, device="cuda").contiguous(memory_format=CL)
    g = None
    res = {}
    for name, fused in (("fused", True), ("per_conv", False)):
        RN._DCN_FUSED = fused
        try:
            for p in params.values():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            assert blk.uses_fused_path() == fused
            y = blk(xi, in_relu=False, out_private=False)
            if g is None:
                g = torch.randn_like(y)
            y.backward(g)
            res[name] = (y.detach().cpu(), xi.grad.cpu(), {k: p.grad.detach().cpu().clone() for k, p in params.items()})
        finally:
            RN._DCN_FUSED = True
    assert float(res["fused"][2]["conv2.offset.weight"].abs().max()) > 0
    # (b) float64
    dd = lambda t: t.detach().cpu().double()
    P = {k: dd(v).requires_grad_(True) for k, v in params.items()}
    fold = lambda bn: tuple(dd(t).view(1, -1, 1, 1) for t in bn.folded())
    xr = dd(x).requires_grad_(True)
    s1, b1 = fold(blk.bn1)
    s2, b2 = fold(blk.bn2)
    s3, b3 = fold(blk.bn3)
    sd, bd = fold(blk.downsample[1])
    y1 = F.relu(F.conv2d(xr, P["conv1.weight"], None, stride) * s1 + b1)
    y2 = F.relu(R2.dfconv2d(y1, P["conv2.offset.weight"], P["conv2.offset.bias"], P["conv2.conv.weight"], modulated, dg)
                * s2 + b2)
    idn = F.conv2d(xr, P["downsample.0.weight"], None, stride) * sd + bd
    yr = F.relu(F.conv2d(y2, P["conv3.weight"]) * s3 + b3 + idn)
    yr.backward(dd(g))
    for name in ("fused", "per_conv"):
        y, gx, gp = res[name]
        torch.testing.assert_close(y.double(), yr.detach(), rtol=1e-4, atol=1e-4, msg=name + ": output")
        scale = float(xr.grad.abs().max())
        assert float((gx.double() - xr.grad).abs().max()) / scale < 2e-4, name + ": input gradient"
        for k, v in gp.items():
            ref = P[k].grad
            err = float((v.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            assert err < 3e-4, "%s: gradient of %s off by %.2e" % (name, k, err)
    # and the two HIP routes against each other
    torch.testing.assert_close(res["fused"][0], res["per_conv"][0], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("modulated", [False, True])
def test_deform_backward_leaves_the_offset_gradients_maximum(modulated):
    """contraction mode 4: the gradient w.r.t. the offset conv's output (gom) is an operand of that conv's weight-gradient
    GEMM; dadet_deform_sample_backward_ld_m leaves max|gom| in the slot attached to it (the pass that stores the gradients
    reduces their magnitudes on the way) — equal to the tensor's own maximum, padding columns zero"""
    from da_detect_amd import _C, amax

    if _C.get_gemm_mode() != 4:
        pytest.skip("the slot exists in contraction mode 4 only")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    N, C, H, W, k, dg = 2, 128, 24, 36, 3, 1
    T = k * k
    ld = (3 * T * dg if modulated else 2 * T * dg) + 3 & ~3
    x = torch.randn((N, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    om = (torch.randn((N, ld, H, W), generator=g) * 1.5).to(dev).contiguous(memory_format=torch.channels_last)
    gcols = torch.randn((N, T * C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gx, gom = _C.deform_sample_backward_om(x, om, gcols, k, k, 1, k // 2, 1, dg, modulated)
    used = 3 * T * dg if modulated else 2 * T * dg
    assert float(gom[:, used:].abs().max()) == 0.0 if used < ld else True
    got = amax.value(gom)
    assert got is not None and got == float(gom.abs().max()) and got > 0.0
    assert amax.slot_of(_C._nhwc(gom)) is not None          # what the weight-gradient wrapper will find


# ---------------------------------------------------------------------------------------------------------------
# the vendored tree's native entry points by the reference's names and call forms
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 32, 12, 15, 24, 3, 1, 1), (1, 64, 11, 9, 16, 3, 2, 2)])
def test_vendored_deform_conv_entry_points_by_the_reference_call_forms(device, cfg):
    """_C.deform_conv_forward / _backward_input / _backward_parameters exactly as deform_conv_func.py:38-121 of the vendored
    tree calls them — plain contiguous NCHW tensors, CALLER-allocated `output` (new_empty), `grad_input` / `grad_offset` /
    `grad_weight` (zeros_like), empty `columns` / `ones` scratch, (kW, kH, dW, dH, padW, padH, dilW, dilH, group,
    deformable_group, im2col_step) — against autograd on oracle/deform_ref.py"""
    from da_detect_amd import _C
    from oracle import deform_ref as R

    N, C, H, W, Cout, k, stride, dg = cfg
    x, off, _, w, _, pad = _case(sum(cfg), N, C, H, W, Cout, k, stride, dg, False)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, w)]
    want = R.deform_conv2d(leaves[0], leaves[1], None, leaves[2], None, stride, pad, 1, dg)
    gy = torch.randn(want.shape, generator=torch.Generator().manual_seed(2))
    want.backward(gy)
    input, offset, weight, grad_output = x.to(device), off.to(device), w.to(device), gy.to(device)
    output = input.new_empty(want.shape)
    bufs_ = [input.new_empty(0), input.new_empty(0)]       # columns, ones
    step = min(64, N)
    assert _C.deform_conv_forward(input, weight, offset, output, bufs_[0], bufs_[1], weight.size(3), weight.size(2), stride,
                                  stride, pad, pad, 1, 1, 1, dg, step) == 1
    assert output.is_contiguous() and bufs_[0].numel() == 0 and bufs_[1].numel() == 0      # the caller's scratch is left alone
    torch.testing.assert_close(output.cpu(), want.detach(), rtol=1e-4, atol=1e-4)
    grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
    _C.deform_conv_backward_input(input, offset, grad_output, grad_input, grad_offset, weight, bufs_[0], weight.size(3),
                                  weight.size(2), stride, stride, pad, pad, 1, 1, 1, dg, step)
    grad_weight = torch.zeros_like(weight)
    _C.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, bufs_[0], bufs_[1], weight.size(3),
                                       weight.size(2), stride, stride, pad, pad, 1, 1, 1, dg, 1, step)
    for name, got, ref in (("input", grad_input, leaves[0].grad), ("offset", grad_offset, leaves[1].grad),
                           ("weight", grad_weight, leaves[2].grad)):
        err = float((got.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 2e-4, "grad %s: %.3e" % (name, err)
    # the reference ACCUMULATES the weight gradient (addmm_ with beta 1, scaled): a second call with scale 0.5 adds half
    _C.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, bufs_[0], bufs_[1], weight.size(3),
                                       weight.size(2), stride, stride, pad, pad, 1, 1, 1, dg, 0.5, step)
    torch.testing.assert_close(grad_weight.cpu(), 1.5 * leaves[2].grad, rtol=1e-3, atol=2e-4 * float(leaves[2].grad.abs().max()))
    with pytest.raises(NotImplementedError):
        _C.deform_conv_forward(input, weight, offset, output, bufs_[0], bufs_[1], k, k, stride, stride, pad, pad, 1, 1, 2, dg, step)


@pytest.mark.gpu
@pytest.mark.parametrize("with_bias", [True, False])
def test_vendored_modulated_deform_conv_entry_points_by_the_reference_call_forms(device, with_bias):
    """_C.modulated_deform_conv_forward / _backward as deform_conv_func.py:167-235 calls them (kernel_h before kernel_w here,
    a one-element fake bias without one, five caller-allocated zero gradients) against autograd on oracle/deform_ref.py"""
    from da_detect_amd import _C
    from oracle import deform_ref as R

    N, C, H, W, Cout, k, stride, dg = 2, 32, 13, 10, 20, 3, 1, 2
    x, off, mask, w, b, pad = _case(31, N, C, H, W, Cout, k, stride, dg, True)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, mask, w, b)]
    want = R.deform_conv2d(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4] if with_bias else None, stride, pad, 1, dg)
    gy = torch.randn(want.shape, generator=torch.Generator().manual_seed(3))
    want.backward(gy)
    input, offset, msk, weight = x.to(device), off.to(device), mask.to(device), w.to(device)
    bias = b.to(device) if with_bias else input.new_empty(1)       # "fake tensor", deform_conv_func.py:170
    output = input.new_empty(want.shape)
    _bufs = [input.new_empty(0), input.new_empty(0)]
    _C.modulated_deform_conv_forward(input, weight, bias, _bufs[0], offset, msk, output, _bufs[1], weight.shape[2],
                                     weight.shape[3], stride, stride, pad, pad, 1, 1, 1, dg, with_bias)
    torch.testing.assert_close(output.cpu(), want.detach(), rtol=1e-4, atol=1e-4)
    grads = [torch.zeros_like(t) for t in (input, offset, msk, weight, bias)]
    _C.modulated_deform_conv_backward(input, weight, bias, _bufs[0], offset, msk, _bufs[1], grads[0], grads[3], grads[4],
                                      grads[1], grads[2], gy.to(device), weight.shape[2], weight.shape[3], stride, stride,
                                      pad, pad, 1, 1, 1, dg, with_bias)
    names = ["input", "offset", "mask", "weight"] + (["bias"] if with_bias else [])
    for name, got, leaf in zip(names, grads, leaves):
        err = float((got.cpu() - leaf.grad).abs().max()) / (float(leaf.grad.abs().max()) + 1e-12)
        assert err < 2e-4, "grad %s: %.3e" % (name, err)
    if not with_bias:
        assert float(grads[4].abs().max()) == 0.0


@pytest.mark.gpu
def test_vendored_deform_psroi_pooling_entry_points_by_the_reference_call_forms():
    """_C.deform_psroi_pooling_forward / _backward as deform_pool_func.py:38-97 calls them: caller-allocated `output`,
    `output_count`, zero `grad_input` / `grad_offset`"""
    from da_detect_amd import _C
    from oracle.deform_ref import deform_psroi_pool

    data, rois, trans, grad, kw = _psroi_case(13, False, 3, 2, 4)
    ref_out, ref_cnt, ref_gd, ref_gt = deform_psroi_pool(data, rois, trans, grad_out=grad, **kw)
    d, r, t = data.cuda(), rois.cuda(), trans.cuda()
    n = r.shape[0]
    output = d.new_empty(n, kw["out_dim"], kw["out_size"], kw["out_size"])
    output_count = d.new_empty(n, kw["out_dim"], kw["out_size"], kw["out_size"])
    _C.deform_psroi_pooling_forward(d, r, t, output, output_count, kw["no_trans"], kw["spatial_scale"], kw["out_dim"],
                                    kw["group_size"], kw["out_size"], kw["part_size"], kw["sample_per_part"], kw["trans_std"])
    torch.testing.assert_close(output.cpu(), ref_out, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(output_count.cpu(), ref_cnt)
    grad_input, grad_offset = torch.zeros_like(d), torch.zeros_like(t)
    _C.deform_psroi_pooling_backward(grad.cuda(), d, r, t, output_count, grad_input, grad_offset, kw["no_trans"],
                                     kw["spatial_scale"], kw["out_dim"], kw["group_size"], kw["out_size"], kw["part_size"],
                                     kw["sample_per_part"], kw["trans_std"])
    torch.testing.assert_close(grad_input.cpu(), ref_gd, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(grad_offset.cpu(), ref_gt, rtol=1e-4, atol=2e-5)
