"""CPU restatement of deformable convolution v1 / v2 in plain torch ops (autograd supplies the backward).

TEST INFRASTRUCTURE ONLY.  Restated from the reference's CUDA kernels — there is no CPU implementation in the
reference to run, so this oracle is NOT pinned against reference outputs ("parity unpinned"); what pins it are
identities: zero offsets + unit mask == F.conv2d, mask == 1 makes v2 == v1, and integer offsets == a shifted conv.
Reference: tools/cityscapes/maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:92-123 (bilinear with
zero-padded corners), :198-250 (v1 sampling, valid iff -1 < h < H and -1 < w < W), :578-640 (v2: sample * mask);
offset channel order g*2*kh*kw + 2*(i*kw + j) + {0: dy, 1: dx}."""
import torch


def deform_sample(x, offset, mask, kh, kw, stride, pad, dil, dg):
    """x [N,C,H,W], offset [N,dg*2*T,Ho,Wo], mask [N,dg*T,Ho,Wo] | None -> cols [N, T, C, Ho, Wo]"""
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[2], offset.shape[3]
    T = kh * kw
    cpg = C // dg
    ys = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1) * stride - pad
    xs = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo) * stride - pad
    xf = x.reshape(N, C, H * W)
    out = []
    for tap in range(T):
        i, j = tap // kw, tap % kw
        per_group = []
        for g in range(dg):
            oh = offset[:, g * 2 * T + 2 * tap]
            ow = offset[:, g * 2 * T + 2 * tap + 1]
            h = ys + i * dil + oh          # [N,Ho,Wo]
            w = xs + j * dil + ow
            valid = (h > -1) & (w > -1) & (h < H) & (w < W)
            hl, wl = torch.floor(h), torch.floor(w)
            lh, lw = h - hl, w - wl
            hl, wl = hl.long(), wl.long()
            val = 0
            for (dy, dx, wt) in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                hy, wx = hl + dy, wl + dx
                ok = valid & (hy >= 0) & (hy <= H - 1) & (wx >= 0) & (wx <= W - 1)
                idx = (hy.clamp(0, H - 1) * W + wx.clamp(0, W - 1)).view(N, 1, Ho * Wo).expand(N, cpg, Ho * Wo)
                v = torch.gather(xf[:, g * cpg:(g + 1) * cpg], 2, idx).view(N, cpg, Ho, Wo)
                val = val + v * (wt * ok.to(x.dtype)).unsqueeze(1)
            if mask is not None:
                val = val * mask[:, g * T + tap].unsqueeze(1)
            per_group.append(val)
        out.append(torch.cat(per_group, 1))
    return torch.stack(out, 1)


def deform_conv2d(x, offset, mask, weight, bias=None, stride=1, pad=0, dil=1, dg=1):
    cout, cin, kh, kw = weight.shape
    cols = deform_sample(x, offset, mask, kh, kw, stride, pad, dil, dg)      # [N,T,C,Ho,Wo]
    y = torch.einsum("ntchw,otc->nohw", cols, weight.reshape(cout, cin, kh * kw).permute(0, 2, 1))
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def deform_psroi_pool(data, rois, trans, spatial_scale, out_size, out_dim, no_trans, group_size, part_size,
                      sample_per_part, trans_std, grad_out=None):
    """Deformable PSROI pooling restated in fp32 numpy scalar loops (small cases only) from
    tools/cityscapes/maskrcnn_benchmark/csrc/cuda/deform_pool_kernel_cuda.cu:30-141 (forward) and :143-264
    (backward, incl. the reference's analytic offset gradient, which is NOT autograd of the clamped forward).
    data [B,C,H,W], rois [R,5], trans [R,2*ncls,part,part] -> out, count [R,out_dim,P,P]
    (+ grad_data, grad_trans when grad_out is given).  Parity unpinned: no CPU reference exists to run."""
    import math

    import numpy as np
    f = np.float32
    data = data.detach().numpy().astype(f)
    rois = rois.detach().numpy().astype(f)
    B, C, H, W = data.shape
    R, P = rois.shape[0], out_size
    ncls = 1 if no_trans else trans.shape[1] // 2
    tr = None if no_trans else trans.detach().numpy().astype(f)
    cec = out_dim // ncls
    out = np.zeros((R, out_dim, P, P), f)
    cnt = np.zeros((R, out_dim, P, P), f)
    gdata = np.zeros_like(data, dtype=np.float64)
    gtrans = None if no_trans else np.zeros(tr.shape, np.float64)
    go = None if grad_out is None else grad_out.detach().numpy().astype(f)
    scale, tstd = f(spatial_scale), f(trans_std)
    rnd = lambda v: f(math.floor(abs(float(v)) + 0.5) * (1 if v >= 0 else -1))  # C roundf (half away from zero)
    for n in range(R):
        b = int(rois[n, 0])
        sw = rnd(rois[n, 1]) * scale - f(0.5)
        sh = rnd(rois[n, 2]) * scale - f(0.5)
        ew = (rnd(rois[n, 3]) + f(1)) * scale - f(0.5)
        eh = (rnd(rois[n, 4]) + f(1)) * scale - f(0.5)
        rw, rh = max(ew - sw, f(0.1)), max(eh - sh, f(0.1))
        bh, bw = rh / f(P), rw / f(P)
        sbh, sbw = bh / f(sample_per_part), bw / f(sample_per_part)
        for ctop in range(out_dim):
            cls = ctop // cec
            for ph in range(P):
                for pw in range(P):
                    part_h = int(math.floor(f(ph) / f(P) * f(part_size)))
                    part_w = int(math.floor(f(pw) / f(P) * f(part_size)))
                    tx = f(0) if no_trans else tr[n, cls * 2, part_h, part_w] * tstd
                    ty = f(0) if no_trans else tr[n, cls * 2 + 1, part_h, part_w] * tstd
                    ws = f(pw) * bw + sw + tx * rw
                    hs = f(ph) * bh + sh + ty * rh
                    gw = min(max(int(math.floor(f(pw) * f(group_size) / f(P))), 0), group_size - 1)
                    gh = min(max(int(math.floor(f(ph) * f(group_size) / f(P))), 0), group_size - 1)
                    c = (ctop * group_size + gh) * group_size + gw
                    samples = []
                    s = f(0)
                    for ih in range(sample_per_part):
                        for iw in range(sample_per_part):
                            w = ws + f(iw) * sbw
                            h = hs + f(ih) * sbh
                            if w < -0.5 or w > W - 0.5 or h < -0.5 or h > H - 0.5:
                                continue
                            w = min(max(w, f(0)), f(W - 1))
                            h = min(max(h, f(0)), f(H - 1))
                            x1, x2, y1, y2 = int(math.floor(w)), int(math.ceil(w)), int(math.floor(h)), int(math.ceil(h))
                            dx, dy = w - f(x1), h - f(y1)
                            v = ((f(1) - dx) * (f(1) - dy) * data[b, c, y1, x1] + (f(1) - dx) * dy * data[b, c, y2, x1]
                                 + dx * (f(1) - dy) * data[b, c, y1, x2] + dx * dy * data[b, c, y2, x2])
                            s = f(s + v)
                            samples.append((x1, x2, y1, y2, dx, dy))
                    k = len(samples)
                    out[n, ctop, ph, pw] = f(0) if k == 0 else s / f(k)
                    cnt[n, ctop, ph, pw] = k
                    if go is None or k == 0:
                        continue
                    d = go[n, ctop, ph, pw] / f(k)
                    for (x0, x1, y0, y1, dx, dy) in samples:
                        gdata[b, c, y0, x0] += (1 - dx) * (1 - dy) * d
                        gdata[b, c, y1, x0] += (1 - dx) * dy * d
                        gdata[b, c, y0, x1] += dx * (1 - dy) * d
                        gdata[b, c, y1, x1] += dx * dy * d
                        if no_trans:
                            continue
                        U00, U01, U10, U11 = data[b, c, y0, x0], data[b, c, y1, x0], data[b, c, y0, x1], data[b, c, y1, x1]
                        gx = (U11 * dy + U10 * (1 - dy) - U01 * dy - U00 * (1 - dy)) * tstd * d * rw
                        gy = (U11 * dx + U01 * (1 - dx) - U10 * dx - U00 * (1 - dx)) * tstd * d * rh
                        gtrans[n, cls * 2, part_h, part_w] += gx
                        gtrans[n, cls * 2 + 1, part_h, part_w] += gy
    res = [torch.from_numpy(out), torch.from_numpy(cnt)]
    if go is not None:
        res += [torch.from_numpy(gdata.astype(f)), None if no_trans else torch.from_numpy(gtrans.astype(f))]
    return res
