"""Convolution / linear layers on the implicit-GEMM HIP kernels, with hand-written backward.

`conv2d_affine_act` is the fused unit the backbone is made of: conv -> per-channel affine (a folded
FrozenBatchNorm2d, or a bias) -> (+ residual) -> ReLU, i.e. what the reference executes as four ATen
calls per layer (reference: maskrcnn_benchmark/modeling/backbone/resnet.py:294-314, layers/batch_norm.py:19-24).
Backward = ReLU/affine gating kernel, data gradient by the same forward kernel on transposed weights,
weight gradient by the split-K wgrad kernel, bias gradient by a column sum.
"""
import weakref

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from .. import amax as _amax
from ..utils.streams import WgradLane, bias_grad

CL = torch.channels_last


class _ConvAffineAct(Function):
    @staticmethod
    def forward(ctx, x, weight, scale, bias, residual, stride, pad, relu, out_size):
        y = _C.conv_forward(x, weight, scale, bias, addend=residual, stride=stride, pad=pad,
                            relu_mode=1 if relu else 0, out_size=out_size)
        ctx.stride, ctx.pad, ctx.relu = stride, pad, relu
        ctx.has_res = residual is not None
        ctx.x_shape = tuple(x.shape)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.bias = bias          # (not saved for its values: backward only asks whether it has a direct gradient slot)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        need_x, need_w, _, need_b, need_res = ctx.needs_input_grad[:5]
        need_res = need_res and ctx.has_res
        k = weight.shape[2]
        # S = gy gated by the ReLU (gradient wrt the affine output); g = S * scale (gradient wrt the conv output)
        if y is None and scale is None:
            S = g = gy.contiguous(memory_format=CL)
        else:
            S, g = _C.relu_bn_backward(gy, y, scale, want_unscaled=(need_res or need_b) and scale is not None)
            if S is None:
                S = g
        gx = gw = gb = None
        lane = WgradLane(gy.device)
        if need_w:      # beside the data gradient, on the weight-gradient stream
            gw = lane.run_into(weight, lambda acc: _C.conv_wgrad(x, g, tuple(weight.shape), ctx.stride, ctx.pad,
                                                                 dw=acc, accumulate=True),
                               lambda: _C.conv_wgrad(x, g, tuple(weight.shape), ctx.stride, ctx.pad), x, g)
        if need_x:
            wt = _C.conv_weight_transpose(weight)
            if ctx.stride == 1:
                gx = _C.conv_forward(g, wt, stride=1, pad=k - 1 - ctx.pad)
                if tuple(gx.shape) != ctx.x_shape:  # out_size-trimmed forward (stem): not needed by any caller
                    raise NotImplementedError("dgrad of an out_size-trimmed convolution")
            elif k == 1 and ctx.pad == 0:
                gx = _C.conv_forward(g, wt, stride=1, pad=0, out_spatial_stride=ctx.stride,
                                     out_hw=ctx.x_shape[2:])
            else:
                raise NotImplementedError("data gradient of a strided %dx%d convolution (the reference's "
                                          "configs keep the stride in the 1x1, STRIDE_IN_1X1=True)" % (k, k))
        if need_b:
            gb = bias_grad(ctx.bias, S)
        lane.join()
        return gx, gw, None, gb, (S if need_res else None), None, None, None, None


def conv2d_affine_act(x, weight, scale=None, bias=None, residual=None, stride=1, padding=0, relu=False,
                      out_size=None):
    """relu?( conv2d(x, weight) * scale + bias + residual ) on the HIP device."""
    if x.shape[0] == 0:
        Ho, Wo = out_size or _C.conv_out_size(x.shape[2], x.shape[3], weight.shape[2], weight.shape[3], stride,
                                              padding)
        return x.new_empty((0, weight.shape[0], Ho, Wo))
    pad = (-weight.shape[0]) % 4
    if pad and residual is None:
        # output widths that are not a multiple of 4 (e.g. the 18 / 27 offset channels of DFConv2d): zero rows are
        # appended for the kernels (their data-gradient GEMM contracts over Cout) and sliced off again
        w = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, pad))
        sc = F.pad(scale, (0, pad), value=1.0) if scale is not None else None
        bi = F.pad(bias, (0, pad)) if bias is not None else None
        y = _ConvAffineAct.apply(x, w, sc, bi, None, stride, padding, relu, out_size)
        return y[:, :weight.shape[0]]
    return _ConvAffineAct.apply(x, weight, scale, bias, residual, stride, padding, relu, out_size)


def _pad_rows(t, rows):
    return t if rows == 0 else F.pad(t, (0, 0) * (t.dim() - 1) + (0, rows))


def linear(x, weight, bias=None, relu=False):
    """y = relu?(x @ weight.T + bias) as a 1x1 convolution on [M, C, 1, 1].  Output widths that are not a
    multiple of 4 are zero-padded for the kernel and sliced back (the padding rows get zero gradient)."""
    out_f = weight.shape[0]
    pad = (-out_f) % 4
    w = _pad_rows(weight, pad).view(out_f + pad, weight.shape[1], 1, 1)
    b = _pad_rows(bias, pad) if bias is not None else None
    # (views are new tensor objects: the largest magnitude the producer attached is handed on by hand, amax.py)
    y4 = conv2d_affine_act(_amax.carry(x.reshape(x.shape[0], x.shape[1], 1, 1), x), w, None, b, relu=relu)
    y = _amax.carry(y4.view(y4.shape[0], out_f + pad), y4)
    return y[:, :out_f] if pad else y


_FUSED_1X1 = {}      # (ids of the weights and biases) -> (stamp, fused weight, fused bias, weak references): conv1x1_multi


def conv1x1_multi(x, weights, biases, relu=False):
    """several 1x1 convolutions sharing the input run as ONE GEMM (e.g. RPN cls_logits + bbox_pred,
    rpn/rpn.py:44-45; cls_score + bbox_pred, roi_box_predictors.py:31-32).  Returns one tensor per weight,
    each a channel slice of the fused output."""
    sizes = [w.shape[0] for w in weights]
    total = sum(sizes)
    pad = (-total) % 4
    key = stamp = None
    if not torch.is_grad_enabled() and all(t.is_cuda for t in weights):
        # no graph to build (the shared RPN head of a pyramid runs five times per step like this): the fused weight / bias
        # of one weight epoch are built once — 4 launches per call otherwise, and one more pass for the fused weight's
        # largest magnitude (contraction mode 4)
        key = tuple(id(t) for t in list(weights) + list(biases))
        stamp = (_C.weight_epoch(),) + tuple(t._version for t in list(weights) + list(biases))
        hit = _FUSED_1X1.get(key)
        if hit is not None and hit[0] == stamp and all(r() is t for r, t in zip(hit[3], list(weights) + list(biases))):
            w, b = hit[1], hit[2]
            key = None
    if key is not None or stamp is None:
        w = torch.cat([wi.reshape(wi.shape[0], wi.shape[1], 1, 1) for wi in weights], 0)
        b = torch.cat(list(biases), 0)
        if pad:
            w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, pad))
            b = F.pad(b, (0, pad))
        if key is not None:
            if len(_FUSED_1X1) > 64:
                _FUSED_1X1.clear()
            _FUSED_1X1[key] = (stamp, w, b, [weakref.ref(t) for t in list(weights) + list(biases)])
    y = conv2d_affine_act(x, w, None, b, relu=relu)
    outs, o = [], 0
    for s in sizes:
        outs.append(y[:, o:o + s])
        o += s
    return outs
