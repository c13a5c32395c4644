"""Autograd wrappers of the fused DA-head kernels (da_detect_amd/csrc/da_heads.hip).

`da_image_head` evaluates the image-level domain classifier ONCE and returns everything the reference obtains
from its two passes behind GRL(-w) and GRL(+w) (reference: maskrcnn_benchmark/modeling/da_heads/da_heads.py:
409-419; DAImgHead :32-37; losses da_heads/loss.py:80-98, layers/consistency_loss.py:12-14):
    bce       mean BCE-with-logits against the per-image domain label        (adversarial path, GRL weight w_adv)
    mean_sig  per-image spatial mean of sigmoid(logit)                       (consistency path, GRL weight w_cst)
The gradient-reversal weights are applied inside backward: head parameters receive d(bce) + d(cst), the
feature map receives w_adv * d(bce) + w_cst * d(cst).  w_adv may be a float, a 0-d device tensor, or a
callable mapping the (detached) bce value to such a tensor (adaptive GRL, resolved without a host sync).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import _C

CL = torch.channels_last


class _DAImageHead(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, labels, w_adv, w_cst):
        N, C, H, W = x.shape
        t = _C.conv_forward(x, w1, None, b1, relu_mode=1)           # [N,C1,H,W] NHWC == [M][C1]
        w2v = w2.reshape(-1).contiguous()
        logits, sums = _C.da_img_head_loss_forward(t, w2v, b2, labels, N, H * W)
        ctx.save_for_backward(x, w1, t, w2v, logits, labels)
        ctx.dims = (N, H, W)
        ctx.w2_shape = tuple(w2.shape)
        bce = sums[:, 0].sum() / float(N * H * W)
        # an adaptive reversal weight is a function of this very loss value (AdvGRL): resolve it now, it is
        # only consumed by backward
        ctx.w_adv = w_adv(bce.detach()) if callable(w_adv) else w_adv
        ctx.w_cst = w_cst
        mean_sig = sums[:, 1] / float(H * W)
        out_logits = logits.view(N, 1, H, W)
        ctx.mark_non_differentiable(out_logits)
        return bce, mean_sig, out_logits

    @staticmethod
    @once_differentiable
    def backward(ctx, g_bce, g_mean_sig, _g_logits):
        x, w1, t, w2v, logits, labels = ctx.saved_tensors
        N, H, W = ctx.dims
        need_x = ctx.needs_input_grad[0]
        a_bce = (g_bce / float(N * H * W)).reshape(1).expand(N)
        a_sig = g_mean_sig / float(H * W)
        coef = torch.stack([a_bce, a_sig, a_bce * ctx.w_adv, a_sig * ctx.w_cst], dim=1).contiguous()
        g_t_w, g_t_x, g_w2, g_b2 = _C.da_img_head_loss_backward(t, w2v, logits, labels, coef, N, H * W,
                                                                need_x=need_x)
        g_w1 = _C.conv_wgrad(x, g_t_w, tuple(w1.shape), 1, 0)
        g_b1 = _C.colsum(g_t_w)
        g_x = _C.conv_forward(g_t_x, _C.conv_weight_transpose(w1)) if need_x else None
        return g_x, g_w1, g_b1, g_w2.view(ctx.w2_shape), g_b2, None, None, None


def da_image_head(x, conv1, conv2, labels, w_adv, w_cst):
    """x [N,C,H,W]; conv1 / conv2 the DAImgHead convolutions; labels float [N] (1 = source)."""
    return _DAImageHead.apply(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias, labels, w_adv, w_cst)


class _TripletW(Function):
    """nn.TripletMarginLoss(margin, p=2) on [1,C,H,W] maps: distance over the LAST axis (W)
    (reference: da_heads/loss.py:180-200)."""

    @staticmethod
    def forward(ctx, a, p, n, margin):
        loss, dist = _C.triplet_w_forward(a, p, n, margin)
        ctx.save_for_backward(a, p, n, dist)
        ctx.margin = margin
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, p, n, dist = ctx.saved_tensors
        _, C, H, W = a.shape
        g_scale = (g / float(C * H)).reshape(1).contiguous()
        ga, gp, gn = _C.triplet_w_backward(a, p, n, dist, g_scale, ctx.margin, need=ctx.needs_input_grad[:3])
        return ga, gp, gn, None


def triplet_margin_loss_w(anchor, positive, negative, margin):
    assert anchor.dim() == 4 and anchor.shape[0] == 1, "image-level triplet loss expects [1,C,H,W] maps"
    return _TripletW.apply(anchor, positive, negative, float(margin))
