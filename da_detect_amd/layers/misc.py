"""Small layers of the reference's operator API (reference: maskrcnn_benchmark/layers/{batch_norm,misc,
gradient_scalar_layer,smooth_l1_loss,consistency_loss,sigmoid_focal_loss}.py), backed by the HIP library."""

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from .. import amax as _amax
from .conv import conv2d_affine_act

CL = torch.channels_last


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (layers/batch_norm.py:6-24): buffers
    weight / bias / running_mean / running_var, y = x * scale + shift with scale = weight * rsqrt(var)
    (no epsilon, like the reference).  Inside the backbone the affine is folded into the conv epilogue
    through `folded()`; called standalone it runs the channel_affine kernel."""

    def __init__(self, n):
        super(FrozenBatchNorm2d, self).__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._cache = None

    def folded(self):
        """(scale, shift) fp32 vectors, cached until a buffer is modified (e.g. by load_state_dict)"""
        key = (self.weight._version, self.bias._version, self.running_mean._version,
               self.running_var._version, self.weight.device)
        if self._cache is None or self._cache[0] != key:
            scale = self.weight * self.running_var.rsqrt()
            shift = self.bias - self.running_mean * scale
            self._cache = (key, scale.contiguous(), shift.contiguous())
        return self._cache[1], self._cache[2]

    def forward(self, x):
        scale, shift = self.folded()
        return _ChannelAffine.apply(x, scale, shift)


class _ChannelAffine(Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        ctx.save_for_backward(scale)
        return _C.channel_affine(x, scale, shift)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return _C.relu_bn_backward(g, None, scale)[1], None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose forward runs the implicit-GEMM kernel; keeps the reference wrapper's behaviour of
    returning a correctly-shaped empty tensor for empty inputs (layers/misc.py:30-43).  Parameters are kept
    in channels_last memory format ([Cout][KH][KW][Cin], the kernel's native weight layout)."""

    def __init__(self, *args, **kwargs):
        super(Conv2d, self).__init__(*args, **kwargs)
        if self.groups != 1 or self.dilation != (1, 1) or self.kernel_size[0] != self.kernel_size[1] or \
                self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("Conv2d: only square, undilated, ungrouped convolutions are on the HIP path")
        self.weight.data = self.weight.data.contiguous(memory_format=CL)

    def forward(self, x, scale=None, shift=None, residual=None, relu=False):
        if scale is None:
            return conv2d_affine_act(x, self.weight, None, self.bias, residual, self.stride[0],
                                     self.padding[0], relu)
        assert self.bias is None
        return conv2d_affine_act(x, self.weight, scale, shift, residual, self.stride[0], self.padding[0], relu)


class _GradientScalarLayer(Function):
    """identity forward, weight * grad backward (layers/gradient_scalar_layer.py:4-13)"""

    @staticmethod
    def forward(ctx, input, weight):
        ctx.weight = weight
        return _amax.carry(input.view_as(input), input)

    @staticmethod
    def backward(ctx, grad_output):
        return ctx.weight * grad_output, None


gradient_scalar = _GradientScalarLayer.apply


class GradientScalarLayer(nn.Module):
    def __init__(self, weight):
        super(GradientScalarLayer, self).__init__()
        self.weight = weight

    def forward(self, input):
        return gradient_scalar(input, self.weight)

    def __repr__(self):
        return "{}(weight={})".format(self.__class__.__name__, self.weight)


def smooth_l1_loss(input, target, beta=1.0 / 9, size_average=True):
    """smooth-L1 with the extra beta (layers/smooth_l1_loss.py:6-16)"""
    n = torch.abs(input - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.mean() if size_average else loss.sum()


class _RPNLoss(Function):
    """(loss_objectness, loss_rpn_box_reg) of one level with the gradients produced by the same launch"""

    @staticmethod
    def forward(ctx, objectness, box_regression, sampled_inds, labels_sampled, pos_inds, targets_pos, beta):
        losses, g_obj, g_reg = _C.rpn_loss(objectness, box_regression, sampled_inds, labels_sampled, pos_inds,
                                           targets_pos, beta)
        ctx.save_for_backward(g_obj, g_reg)
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1):
        g_obj, g_reg = ctx.saved_tensors
        return g_obj * g0, g_reg * g1, None, None, None, None, None


class _RPNHeadLossRows(Function):
    """RPN losses of one level AND the single-conv RPN head's whole backward, on the sampled rows only.

    The losses read the head's maps at the S sampled anchors (<= 256 per labelled image; rpn/loss.py:125-143), so
    d loss / d maps is zero at every other pixel and every GEMM of the head's backward (modeling/rpn/rpn.py:39-46 under
    autograd: two 1x1 convolutions, the ReLU, the 3x3 convolution) has at most S non-zero rows out of N*H*W.  Here they
    run on those rows: G [S, 5A] from the loss kernel (row form), the hidden activations t and the 3x3 operand rows of x
    gathered at the rows' pixels, four small GEMMs (through the same conv entry points, as 1x1 convolutions over S "pixels")
    and one scatter for the data gradient.  Same sums as the dense backward — zero rows contribute nothing — in another
    order.  Inputs: x [n,C,H,W] (the head's input), conv / cls_logits / bbox_pred parameters, t = relu(conv(x)), the
    head's output maps, the sample."""

    @staticmethod
    def forward(ctx, x, w3, b3, wc, bc, wb, bb, t, objectness, box_regression, sampled_inds, labels_sampled, n_pos,
                targets_pos, beta, level=None):
        # level (anchors per image over the pyramid, this level's offset, this level's anchors): one level of an FPN head —
        # rows of anchors on other levels are zero and contribute nothing below; the levels' losses add up
        losses, rows, pixels = _C.rpn_loss_rows(objectness, box_regression, sampled_inds, labels_sampled, n_pos,
                                                targets_pos, beta, level=level)
        ctx.save_for_backward(x, w3, wc, wb, t, rows, pixels)
        ctx.A = objectness.shape[1]
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1):
        x, w3, wc, wb, t, rows, pixels = ctx.saved_tensors
        A, S, ldg = ctx.A, rows.shape[0], rows.shape[1]
        C = x.shape[1]
        k = w3.shape[2]
        # upstream factors of the two losses (1 and 1 when they enter the total loss unweighted)
        scale = torch.cat([g0.reshape(1).expand(A), g1.reshape(1).expand(4 * A), rows.new_zeros(ldg - 5 * A)])
        G = (rows * scale).view(S, ldg, 1, 1)
        # (gathered rows are bounded by their map's largest magnitude; views are new objects: handed on by hand, amax.py)
        t_rows = _C.gather_pixel_taps(t, pixels)
        t_rows = _amax.carry(t_rows.view(S, C, 1, 1), t_rows)
        x_cols = _C.gather_pixel_taps(x, pixels, k, k // 2)
        x_cols = _amax.carry(x_cols.view(S, k * k * C, 1, 1), x_cols)
        # the two 1x1 heads as one [5A (+pad), C] matrix, like the forward pass (layers.conv1x1_multi)
        w_head = torch.cat([wc.reshape(A, C), wb.reshape(4 * A, C), rows.new_zeros(ldg - 5 * A, C)], 0)
        d_head = _C.conv_wgrad(t_rows, G, (ldg, C, 1, 1))
        b_head = G.view(S, ldg).sum(0)
        # gradient of the hidden activations at the rows' pixels, gated by the ReLU (relu_mode 2: mask_ref > 0)
        gt = _C.conv_forward(G, w_head.t().contiguous().view(C, ldg, 1, 1), relu_mode=2, mask_ref=t_rows)
        dw3 = _C.conv_wgrad(x_cols, gt, (w3.shape[0], k * k * C, 1, 1))
        dw3 = dw3.view(w3.shape[0], k, k, C).permute(0, 3, 1, 2)          # the parameter's channels_last layout
        db3 = gt.view(S, -1).sum(0)
        # data gradient: y[r][tap][ci] = sum_co gt[r][co] * w3[co][tap][ci], added at pixel_r + tap offset
        # W^T [K, Cout]: the 3x3 weight viewed as a 1x1 convolution over K = (tap, ci) and transposed — through the
        # per-step cache of transposed weights (one batched launch per optimizer step) instead of a 38 MB copy here
        w_1x1 = w3.permute(0, 2, 3, 1).reshape(w3.shape[0], k * k * C, 1, 1)
        y = _C.conv_forward(gt, _C.conv_weight_transpose(w_1x1))
        dx = _C.scatter_pixel_taps_add(y.view(S, k * k, C), pixels, tuple(x.shape), k, k // 2)
        return (dx, dw3, db3, d_head[:A], b_head[:A], d_head[A:5 * A], b_head[A:5 * A]) + (None,) * 9


rpn_head_loss_rows = _RPNHeadLossRows.apply


class _RPNHeadLossRowsPyramid(Function):
    """_RPNHeadLossRows for the SHARED head of a feature pyramid, all levels in one node.  Every sampled anchor lies on
    exactly one level, so the row-form gradient G [S, 5A], the hidden rows and the 3x3 operand rows of ALL levels fit the
    same S-row buffers: each level's launches write the rows tagged with its id (dadet_rpn_loss_rows_level with shared rows,
    dadet_gather_pixel_taps_level), the four GEMMs of the head's backward run ONCE, and the data gradient is scattered per
    level.  Same sums as one node per level (cross-level terms are products with zero rows), a fifth of the launches and no
    autograd additions of the head's parameter gradients across levels.
    apply(w3, b3, wc, bc, wb, bb, sampled_inds, labels_sampled, n_pos, targets_pos, beta, levels, *x_t_obj_reg) with
    levels = [(anchors per image, offset, count)] and x, t, objectness, box_regression per level; gradients for the six
    parameters and every x."""

    @staticmethod
    def forward(ctx, w3, b3, wc, bc, wb, bb, sampled_inds, labels_sampled, n_pos, targets_pos, beta, levels, *maps):
        L = len(levels)
        xs, ts, objs, regs = maps[:L], maps[L:2 * L], maps[2 * L:3 * L], maps[3 * L:4 * L]
        A = objs[0].shape[1]
        S = int(sampled_inds.numel())
        ldg = (5 * A + 3) // 4 * 4
        dev = objs[0].device
        rows = torch.zeros((S, ldg), dtype=torch.float32, device=dev)
        pixels = torch.empty(S, dtype=torch.int32, device=dev)
        row_level = torch.full((S,), -1, dtype=torch.int32, device=dev)
        losses = []
        for lvl in range(L):
            l, _, _ = _C.rpn_loss_rows(objs[lvl], regs[lvl], sampled_inds, labels_sampled, n_pos, targets_pos, beta,
                                       level=levels[lvl], shared=(lvl, rows, pixels, row_level))
            losses.append(l)
        total = torch.stack(losses).sum(0)
        ctx.save_for_backward(w3, wc, wb, rows, pixels, row_level, *xs, *ts)
        ctx.A, ctx.L = A, L
        return total[0], total[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1):
        saved = ctx.saved_tensors
        w3, wc, wb, rows, pixels, row_level = saved[:6]
        A, L = ctx.A, ctx.L
        xs, ts = saved[6:6 + L], saved[6 + L:6 + 2 * L]
        S, ldg = rows.shape
        C = xs[0].shape[1]
        k = w3.shape[2]
        scale = torch.cat([g0.reshape(1).expand(A), g1.reshape(1).expand(4 * A), rows.new_zeros(ldg - 5 * A)])
        G = (rows * scale).view(S, ldg, 1, 1)
        # (zero-filled: a row no level claims — there is none — must not put uninitialised memory next to its zero G row)
        t_rows = torch.zeros((S, 1, C), dtype=torch.float32, device=rows.device)
        x_cols = torch.zeros((S, k * k, C), dtype=torch.float32, device=rows.device)
        for lvl in range(L):
            _C.gather_pixel_taps(ts[lvl], pixels, row_level=row_level, level=lvl, out=t_rows)
            _C.gather_pixel_taps(xs[lvl], pixels, k, k // 2, row_level=row_level, level=lvl, out=x_cols)
        t_rows, x_cols = t_rows.view(S, C, 1, 1), x_cols.view(S, k * k * C, 1, 1)
        w_head = torch.cat([wc.reshape(A, C), wb.reshape(4 * A, C), rows.new_zeros(ldg - 5 * A, C)], 0)
        d_head = _C.conv_wgrad(t_rows, G, (ldg, C, 1, 1))
        b_head = G.view(S, ldg).sum(0)
        gt = _C.conv_forward(G, w_head.t().contiguous().view(C, ldg, 1, 1), relu_mode=2, mask_ref=t_rows)
        dw3 = _C.conv_wgrad(x_cols, gt, (w3.shape[0], k * k * C, 1, 1))
        dw3 = dw3.view(w3.shape[0], k, k, C).permute(0, 3, 1, 2)
        db3 = gt.view(S, -1).sum(0)
        w_1x1 = w3.permute(0, 2, 3, 1).reshape(w3.shape[0], k * k * C, 1, 1)
        y = _C.conv_forward(gt, _C.conv_weight_transpose(w_1x1)).view(S, k * k, C)
        dxs = [_C.scatter_pixel_taps_add(y, pixels, tuple(xs[lvl].shape), k, k // 2, row_level=row_level, level=lvl)
               for lvl in range(L)]
        return (dw3, db3, d_head[:A], b_head[:A], d_head[A:5 * A], b_head[A:5 * A]) + (None,) * 6 + tuple(dxs) + \
            (None,) * (3 * L)


rpn_head_loss_rows_pyramid = _RPNHeadLossRowsPyramid.apply


class _FastRCNNLoss(Function):
    """(classification_loss, box_loss) with the gradients produced by the same launch"""

    @staticmethod
    def forward(ctx, class_logits, box_regression, src, labels_src, rows_pos, map_inds, targets_pos):
        losses, g_cls, g_reg = _C.fast_rcnn_loss(class_logits, box_regression, src, labels_src, rows_pos, map_inds,
                                                 targets_pos)
        ctx.save_for_backward(g_cls, g_reg)
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1):
        g_cls, g_reg = ctx.saved_tensors
        return g_cls * g0, g_reg * g1, None, None, None, None, None


class _FastRCNNLossRows(Function):
    """the same two losses from per-row targets (loss_labels < 0: row outside the losses)"""

    @staticmethod
    def forward(ctx, class_logits, box_regression, loss_labels, regression_targets):
        losses, g_cls, g_reg = _C.fast_rcnn_loss_rows(class_logits, box_regression, loss_labels, regression_targets)
        ctx.save_for_backward(g_cls, g_reg)
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1):
        g_cls, g_reg = ctx.saved_tensors
        return g_cls * g0, g_reg * g1, None, None


rpn_loss_fused = _RPNLoss.apply
fast_rcnn_loss_rows_fused = _FastRCNNLossRows.apply
fast_rcnn_loss_fused = _FastRCNNLoss.apply


def consistency_loss(img_feas, ins_fea, ins_labels, size_average=True):
    """|mean_hw(p_img_i) - p_ins_ij| (layers/consistency_loss.py:3-27).  `img_feas` is a list of per-level
    [N,1,H,W] probability maps, or (fused path) of per-level [N] tensors that already hold the spatial mean."""
    loss = []
    n_src = getattr(ins_labels, "_n_src_host", None)   # set by the box head, which knows it without a round trip
    if n_src is None:
        n_src = int(torch.nonzero(ins_labels).size(0))
    intervals = [n_src, ins_fea.size(0) - n_src]
    for lvl in img_feas:
        means = lvl if lvl.dim() == 1 else torch.mean(lvl.reshape(lvl.shape[0], -1), 1)
        assert means.shape[0] == 2, \
            "only batch size=2 is supported for consistency loss now, received batch size: {}".format(means.shape[0])
        rows = torch.cat([means[i].view(1, 1).repeat(intervals[i], 1) for i in range(2)], dim=0)
        loss.append(torch.abs(rows - ins_fea))
    loss = torch.cat(loss, dim=1)
    return loss.mean() if size_average else loss.sum()


class _SigmoidFocalLoss(Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.num_classes, ctx.gamma, ctx.alpha = logits.shape[1], gamma, alpha
        return _C.sigmoid_focalloss_forward(logits, targets, ctx.num_classes, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        return _C.sigmoid_focalloss_backward(logits, targets, d_loss.contiguous(), ctx.num_classes, ctx.gamma,
                                             ctx.alpha), None, None, None


sigmoid_focal_loss = _SigmoidFocalLoss.apply


class SigmoidFocalLoss(nn.Module):
    """sum of the per-element focal loss (layers/sigmoid_focal_loss.py:55-76)"""

    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        return sigmoid_focal_loss(logits, targets, self.gamma, self.alpha).sum()

    def __repr__(self):
        return "{}(gamma={}, alpha={})".format(self.__class__.__name__, self.gamma, self.alpha)


class _AvgPoolHW(Function):
    """nn.AvgPool2d(k) on a k x k map == mean over H*W -> [R, C]"""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[2], x.shape[3])
        return _C.avgpool_forward(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return _C.avgpool_backward(g, *ctx.hw)


_POOL_MEMO = True      # False: every call pools again (tools/probes/variant_ab.py)


def global_avg_pool(x):
    """mean over H x W -> [R, C].  A second call on the SAME tensor object (same version, same grad mode) returns the first
    call's result: the box predictor (roi_box_predictors.py:17,28) and the instance-level domain classifier
    (da_heads.py:402-407) both average-pool the res5 head's [R, 2048, 7, 7] output — one pooling pass, and in backward the
    two consumers' gradients meet as [R, 2048] vectors in front of ONE un-pooling pass instead of as two 205 MB maps that
    autograd then adds (da, 512 ROIs: -1 avgpool forward, -1 backward, -1 add of 101 us; profiles/r06_step_timeline_da.txt)."""
    if x.shape[0] == 0:
        return x.new_empty((0, x.shape[1]))
    memo = x.__dict__.get("_dadet_pooled") if _POOL_MEMO else None
    if memo is not None and memo[0] == x._version and memo[1] == torch.is_grad_enabled():
        return memo[2]
    y = _AvgPoolHW.apply(x)
    x._dadet_pooled = (x._version, torch.is_grad_enabled(), y)
    return y
