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
