"""A SECOND, independently written CPU statement of deformable convolution v1 / v2, in float64.

TEST INFRASTRUCTURE ONLY (only tests/ may import it).  Like oracle/deform_ref.py it is restated from the reference's
CUDA kernels (tools/cityscapes/maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:92-123, :198-250, :578-640 —
no CPU implementation exists in the reference to run), so in the strict sense BOTH stay "parity unpinned".  What this
file adds is independence: it shares no code and no technique with deform_ref.py.

  deform_ref.py : manual floor / four gathers over a flattened map / explicit validity masks / one einsum over a
                  materialised [N,T,C,Ho,Wo] column tensor.
  this file     : torch.nn.functional.grid_sample (bilinear, padding_mode="zeros", align_corners=True) on absolute
                  sampling positions converted to normalised coordinates, one tap at a time, each tap contracted with its
                  own [Cout, Cin] weight slice by a matmul and accumulated — no column tensor, no hand-written bilinear
                  weights, no hand-written validity rule.  Gradients for x / offset / mask / weight / bias come from
                  autograd through grid_sample's own backward.

Why grid_sample's zero padding IS the reference's rule: the kernel returns 0 unless -1 < h < H and -1 < w < W and, inside
that band, drops every bilinear corner that lies outside the map (deform_conv_kernel_cuda.cu:100-119).  Bilinear
interpolation of the map extended by zeros does exactly that: a sample at h <= -1 or h >= H has both row corners outside
(weight only on zeros), a sample in (-1, 0) or (H-1, H) keeps the one corner row that exists.  tests/test_deform_gpu.py
compares deform_ref.py, this file and the HIP kernels on non-integer offsets, samples in the border bands (-1, 0) and
(H-1, H), samples far outside, modulated and multi-group cases.
"""
import torch
import torch.nn.functional as F


def deform_conv2d(x, offset, mask, weight, bias=None, stride=1, pad=0, dil=1, dg=1):
    """x [N,C,H,W], offset [N, dg*2*kh*kw, Ho, Wo] (channel g*2T + 2*tap + {0: dy, 1: dx}), mask [N, dg*kh*kw, Ho, Wo] or
    None, weight [Cout, C, kh, kw] -> [N, Cout, Ho, Wo]; computed in the dtype of x (use float64)."""
    N, C, H, W = x.shape
    cout, cin, kh, kw = weight.shape
    assert cin == C and C % dg == 0 and H > 1 and W > 1      # a single row / column has no extent to normalise by
    Ho, Wo = offset.shape[2], offset.shape[3]
    T = kh * kw
    cg = C // dg
    base_y = (torch.arange(Ho, dtype=x.dtype) * stride - pad).view(1, Ho, 1).expand(N, Ho, Wo)
    base_x = (torch.arange(Wo, dtype=x.dtype) * stride - pad).view(1, 1, Wo).expand(N, Ho, Wo)
    y = x.new_zeros((N, cout, Ho, Wo))
    for i in range(kh):
        for j in range(kw):
            tap = i * kw + j
            for g in range(dg):
                py = base_y + i * dil + offset[:, g * 2 * T + 2 * tap]
                px = base_x + j * dil + offset[:, g * 2 * T + 2 * tap + 1]
                # absolute pixel position -> grid_sample's normalised coordinate (align_corners=True: -1 <-> pixel 0,
                # +1 <-> pixel size-1)
                gy = 2.0 * py / (H - 1) - 1.0
                gx = 2.0 * px / (W - 1) - 1.0
                grid = torch.stack([gx, gy], dim=-1)                                   # [N, Ho, Wo, 2] (x first)
                s = F.grid_sample(x[:, g * cg:(g + 1) * cg], grid, mode="bilinear", padding_mode="zeros",
                                  align_corners=True)                                   # [N, cg, Ho, Wo]
                if mask is not None:
                    s = s * mask[:, g * T + tap].unsqueeze(1)
                wk = weight[:, g * cg:(g + 1) * cg, i, j]                               # [Cout, cg]
                y = y + torch.matmul(wk, s.reshape(N, cg, Ho * Wo)).view(N, cout, Ho, Wo)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def dfconv2d(x, w_offset, b_offset, weight, modulated, dg=1, stride=1):
    """DFConv2d (vendored layers/misc.py:114-203): offset-predicting conv, `[:, :2*T*dg]` offsets, `[:, -T*dg:]` sigmoid
    modulation, (modulated) deformable conv without bias"""
    kh = weight.shape[2]
    T = kh * weight.shape[3]
    om = F.conv2d(x, w_offset, b_offset, stride=stride, padding=kh // 2)
    if not modulated:
        return deform_conv2d(x, om, None, weight, None, stride, kh // 2, 1, dg)
    return deform_conv2d(x, om[:, :2 * T * dg], om[:, -T * dg:].sigmoid(), weight, None, stride, kh // 2, 1, dg)
