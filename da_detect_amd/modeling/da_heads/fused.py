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
from ...utils.streams import bias_grad

CL = torch.channels_last


class _DAImageHead(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, labels, w_adv, w_cst):
        N, C, H, W = x.shape
        t = _C.conv_forward(x, w1, None, b1, relu_mode=1)           # [N,C1,H,W] NHWC == [M][C1]
        w2v = w2.reshape(-1).contiguous()
        logits, sums = _C.da_img_head_loss_forward(t, w2v, b2, labels, N, H * W)
        ctx.save_for_backward(x, w1, t, w2v, logits, labels)
        ctx.b1 = b1
        ctx.dims = (N, H, W)
        ctx.w2_shape = tuple(w2.shape)
        bce = sums[:, 0].sum() / float(N * H * W)
        # an adaptive reversal weight is a function of this very loss value (AdvGRL): resolve it now, it is
        # only consumed by backward
        ctx.w_adv = w_adv(bce.detach()) if callable(w_adv) else w_adv
        ctx.w_cst = w_cst
        ctx.sig_used = True
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
        # the four per-image coefficients (1 / (N H W), 1 / (H W), the two reversal weights) are formed in the kernel
        g_t_w, g_t_x, g_w2, g_b2 = _C.da_img_head_loss_backward_g(
            t, w2v, logits, labels, g_bce.reshape(1), g_mean_sig if ctx.sig_used else None, ctx.w_adv, ctx.w_cst, N,
            H * W, need_x=need_x)
        g_w1 = _C.conv_wgrad(x, g_t_w, tuple(w1.shape), 1, 0)
        g_b1 = bias_grad(ctx.b1, g_t_w)
        g_x = _C.conv_forward(g_t_x, _C.conv_weight_transpose(w1)) if need_x else None
        return g_x, g_w1, g_b1, g_w2.view(ctx.w2_shape), g_b2, None, None, None


def da_image_head(x, conv1, conv2, labels, w_adv, w_cst):
    """x [N,C,H,W]; conv1 / conv2 the DAImgHead convolutions; labels float [N] (1 = source)."""
    return _DAImageHead.apply(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias, labels, w_adv, w_cst)


class _DAInsHead(Function):
    """The instance-level domain classifier (DAInsHead, da_heads.py:40-68) with its losses, for all the passes the
    reference makes over the same ROI features in one iteration, as ONE autograd node.

    The reference runs the head once behind GRL(-w) for the BCE (da_heads/loss.py:95-97) and once behind GRL(+w) for the
    consistency term (layers/consistency_loss.py:3-27), each pass drawing its own dropout masks (da_heads.py:421-424).
    Up to the first dropout the passes compute the same thing, so here
      forward : h1 = relu(fc1 x)                                   ONE GEMM for all passes
                h1s = [h1 * m1_a ; h1 * m1_b]                      one launch
                h2 = relu(fc2 h1s) * [m2_a ; m2_b]                 ONE GEMM over the stacked rows (+ the mask multiply)
                fc3 + BCE sum + consistency sum                    one launch (csrc/da_heads.hip da_ins_fwd_kernel)
      backward: tail (d fc2-preactivation, d fc3, d means)         one launch
                fc2 weight / bias / data gradients                 GEMMs over the stacked rows
                merge of the passes, ReLU gate, BOTH reversal weights   one launch (da_ins_merge_kernel)
                fc1 weight / bias gradients from the unweighted sum, feature gradient from the GRL-weighted sum.
    `kinds`: ("bce",), ("cst",) or ("bce", "cst") — pass order = row order = order of the dropout draws."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, labels, means, masks1, masks2, grl, kinds, n_src):
        R = x.shape[0]
        P = len(kinds)
        r_bce = R if "bce" in kinds else 0
        r_cst = R if "cst" in kinds else 0
        assert P == masks1.shape[0] == masks2.shape[0] and kinds in (("bce",), ("cst",), ("bce", "cst"))
        C0, C1, C2 = x.shape[1], w1.shape[0], w2.shape[0]
        h1 = _C.conv_forward(x.reshape(R, C0, 1, 1), w1.view(C1, C0, 1, 1), None, b1, relu_mode=1).view(R, C1)
        h1s = _C.da_ins_dropout_rows(h1, masks1)
        h2 = _C.conv_forward(h1s.view(P * R, C1, 1, 1), w2.view(C2, C1, 1, 1), None, b2, relu_mode=1).view(P * R, C2)
        h2.mul_(masks2.view(P * R, C2))
        w3v = w3.reshape(-1).contiguous()
        labels_f = labels.to(torch.float32) if r_bce else None
        logits, sums = _C.da_ins_tail_forward(h2, w3v, b3, labels_f, means if r_cst else None, r_bce, r_cst, n_src)
        ctx.save_for_backward(x, w1, w2, w3v, h1, h1s, masks1, h2, logits, labels_f, means if r_cst else None, grl)
        ctx.conf = (R, P, r_bce, r_cst, n_src, tuple(w3.shape))
        ctx.biases = (b1, b2)
        levels = int(means.shape[0]) if r_cst else 1
        bce = sums[0] / float(max(r_bce, 1))
        cst = sums[1] / float(max(r_cst, 1) * levels)
        out_logits = logits.view(P, R)
        ctx.mark_non_differentiable(out_logits)
        return bce, cst, out_logits

    @staticmethod
    @once_differentiable
    def backward(ctx, g_bce, g_cst, _g_logits):
        x, w1, w2, w3v, h1, h1s, masks1, h2, logits, labels_f, means, grl = ctx.saved_tensors
        R, P, r_bce, r_cst, n_src, w3_shape = ctx.conf
        need_x = ctx.needs_input_grad[0]
        C0, C1, C2 = x.shape[1], w1.shape[0], w2.shape[0]
        levels = int(means.shape[0]) if means is not None else 1
        coef = torch.stack([g_bce.reshape(()) / float(max(r_bce, 1)),
                            g_cst.reshape(()) / float(max(r_cst, 1) * levels)]).contiguous()
        # every mask value is 0 or 1 / keep, and a hidden unit that survived (h2 != 0) has mask 1 / keep
        g_z2, g_w3, g_b3, g_means = _C.da_ins_tail_backward(h2, w3v, logits, labels_f, means, coef, INS_DROPOUT_INV_KEEP,
                                                            r_bce, r_cst, n_src)
        g_z2 = g_z2.view(P * R, C2, 1, 1)
        g_w2 = _C.conv_wgrad(h1s.view(P * R, C1, 1, 1), g_z2, (C2, C1, 1, 1), 1, 0).view(C2, C1)
        g_b2 = bias_grad(ctx.biases[1], g_z2)
        g_h1s = _C.conv_forward(g_z2, _C.conv_weight_transpose(w2.view(C2, C1, 1, 1)))
        g1_w, g1_x = _C.da_ins_merge(g_h1s.view(P * R, C1), masks1, h1, grl, need_x=need_x)
        g1_w4 = g1_w.view(R, C1, 1, 1)
        g_w1 = _C.conv_wgrad(x.reshape(R, C0, 1, 1), g1_w4, (C1, C0, 1, 1), 1, 0).view(C1, C0)
        g_b1 = bias_grad(ctx.biases[0], g1_w4)
        g_x = None
        if need_x:
            g_x = _C.conv_forward(g1_x.view(R, C1, 1, 1), _C.conv_weight_transpose(w1.view(C1, C0, 1, 1))).view(R, C0)
        return (g_x, g_w1, g_b1, g_w2, g_b2, g_w3.view(w3_shape), g_b3, None, g_means) + (None,) * 5


INS_DROPOUT_P = 0.5                                   # DAInsHead: F.dropout(p=0.5) after fc1 and fc2 (da_heads.py:63,65)
INS_DROPOUT_INV_KEEP = 1.0 / (1.0 - INS_DROPOUT_P)


def da_instance_head(x, head, labels, means, masks1, masks2, grl, kinds, n_src):
    """x [R, C] ROI feature vectors; head a DAInsHead; labels [R] domain labels (1 = source); means [L, 2] per-level mean
    sigmoid of the image head on (source, target) or None; masks1 / masks2 [P, R, 1024] dropout masks of the passes; grl
    float [P] device tensor of the passes' gradient-reversal weights -> (mean BCE, consistency loss, logits [P, R])"""
    return _DAInsHead.apply(x, head.fc1_da.weight, head.fc1_da.bias, head.fc2_da.weight, head.fc2_da.bias,
                            head.fc3_da.weight, head.fc3_da.bias, labels, means, masks1, masks2, grl, tuple(kinds),
                            int(n_src))


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
