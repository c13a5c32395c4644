"""ResNet C4 / C5 / FPN bodies and the res5 ROI head, running on the fused implicit-GEMM kernels.

Module tree and parameter / buffer names are the reference's (reference:
maskrcnn_benchmark/modeling/backbone/resnet.py:80-314) so released checkpoints load by key:
  stem.conv1, stem.bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}.
What differs is the execution: every conv -> FrozenBN -> (+identity) -> ReLU group is ONE kernel launch
(the affine, the residual add and the ReLU live in the GEMM epilogue), activations stay NHWC end to end, and
the 3-channel stem input is staged as NHWC4 so the 7x7 conv is an implicit GEMM over a 7x8x4 window.
"""
from collections import namedtuple

import torch
import torch.nn.functional as F
from torch import nn

from ... import _C
from ...layers import Conv2d, FrozenBatchNorm2d, conv2d_affine_act
from ...utils.registry import Registry
from ...utils import streams
from ...utils.streams import WgradLane

StageSpec = namedtuple("StageSpec", ["index", "block_count", "return_features"])


def _specs(*rows):
    return tuple(StageSpec(index=i, block_count=c, return_features=r) for (i, c, r) in rows)


# (stage index, number of bottlenecks, returned?)  — resnet.py:42-74 of the reference
ResNet50StagesTo5 = _specs((1, 3, False), (2, 4, False), (3, 6, False), (4, 3, True))
ResNet50StagesTo4 = _specs((1, 3, False), (2, 4, False), (3, 6, True))
ResNet101StagesTo5 = _specs((1, 3, False), (2, 4, False), (3, 23, False), (4, 3, True))
ResNet101StagesTo4 = _specs((1, 3, False), (2, 4, False), (3, 23, True))
ResNet50FPNStagesTo5 = _specs((1, 3, True), (2, 4, True), (3, 6, True), (4, 3, True))
ResNet101FPNStagesTo5 = _specs((1, 3, True), (2, 4, True), (3, 23, True), (4, 3, True))
ResNet152FPNStagesTo5 = _specs((1, 3, True), (2, 8, True), (3, 36, True), (4, 3, True))


# one reduction launch per block for its split weight gradients (dadet_conv_wgrad_reduce_batch); 0: one pass per tensor
_WGRAD_BATCH = True
# ONE GEMM launch for a block's 3 - 4 weight gradients where they qualify (_C.conv_wgrad_group; res4 / res5 in contraction
# mode 4): the chip's workgroup slots are shared among them, so each parks a third to a quarter of the partial sums its own
# full-chip launch would and reduces 3 - 4 x as many rows per workgroup.  DADET_WGRAD_GROUP=0: one launch per layer.
_WGRAD_GROUP = __import__("os").environ.get("DADET_WGRAD_GROUP", "1") == "1"


class _WgradGroup(object):
    """the weight gradients of one backward node, collected for one grouped launch.  Only with direct accumulation (every
    weight has a persistent gradient buffer, streams.direct_grad_target: the node then returns None for them) and a batched
    reduction pass; `take` returns False for a request that must go the per-layer way."""

    _known = {}       # (shapes of the node's requests) -> the library accepted the group

    def __init__(self, lane, batch, weights):
        self.lane, self.batch, self.reqs = lane, batch, []
        self.on = (_WGRAD_GROUP and batch is not None and _C._mode4() and 2 <= len(weights) <= _C.WGRAD_GROUP_MAX
                   and all(streams.direct_grad_target(w) is not None for w in weights))
        self.key = None
        if self.on:
            self.key = tuple(tuple(w.shape) for w in weights) + (lane.on,)
            self.on = self._known.get(self.key, True) is not False

    def take(self, w, xin, gout, stride, pad, scale, shape=None):
        if not self.on:
            return False
        self.reqs.append(dict(x=xin, gy=gout, weight_shape=tuple(w.shape) if shape is None else shape, stride=stride,
                              pad=pad, out_scale=scale, dw=streams.direct_grad_target(w), accumulate=True))
        return True

    def issue(self):
        """every collected request: one grouped launch, or (first time a node of these shapes is refused: layers of
        fewer than 256 channels, another contraction mode) one launch each — remembered, so that the node's later
        backward passes issue them in their usual places"""
        if not self.reqs:
            return
        reqs, batch = self.reqs, self.batch
        key = self.key + tuple((tuple(r["x"].shape), r["stride"]) for r in reqs)

        def run():
            ok = self._known.get(key, True) and _C.conv_wgrad_group(reqs, batch)
            if not ok:
                self._known[key] = self._known[self.key] = False
                for r in reqs:
                    _C.conv_wgrad(r["x"], r["gy"], r["weight_shape"], r["stride"], r["pad"], out_scale=r["out_scale"],
                                  dw=r["dw"], accumulate=True, pending=batch)

        self.lane.run_group(run, *[t for r in reqs for t in (r["x"], r["gy"])])
        self.reqs = []


class _BottleneckFn(torch.autograd.Function):
    """One bottleneck (resnet.py:294-314) as ONE autograd node with a hand-scheduled backward.

    forward : 3 (or 4) fused conv launches, exactly as Bottleneck.forward.
    backward: every ReLU / FrozenBN gate rides in a GEMM epilogue instead of its own elementwise pass —
      S3 = G * [out > 0]                         (skipped when the only consumer of `out` already gated, see below)
      S2 = [y2 > 0] * dgrad_conv3(S3)             S1 = [y1 > 0] * dgrad_conv2(S2)        (relu_mode 2 epilogue)
      dX = [x > 0]? * (dgrad_conv1(S1) + S3 | dgrad_downsample(S3))                      (addend epilogue)
      dW_k = scale_k * wgrad(input_k, S_k)                                                (out_scale of the reduce)
    the FrozenBN scales are folded into the transposed weights / the wgrad reduce, so no scaled copy of any S exists.
    in_relu    : x is the output of a ReLU whose producer gates its incoming gradient by [x > 0] anyway; returning
                 dX already gated is then exact for every parameter gradient and lets that producer skip its pass.
    out_private: `out` feeds ONLY a following _BottleneckFn with in_relu (inside _Stage), so G arrives gated."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, s1, b1, s2, b2, s3, b3, sd, bd, stride, in_relu, out_private):
        y1 = _C.conv_forward(x, w1, s1, b1, stride=stride, relu_mode=1)
        y2 = _C.conv_forward(y1, w2, s2, b2, pad=1, relu_mode=1)
        idn = x if wd is None else _C.conv_forward(x, wd, sd, bd, stride=stride)
        out = _C.conv_forward(y2, w3, s3, b3, addend=idn, relu_mode=1)
        ctx.stride, ctx.in_relu, ctx.out_private = stride, in_relu, out_private
        ctx.save_for_backward(x, y1, y2, out, w1, w2, w3, wd, s1, s2, s3, sd)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G):
        x, y1, y2, out, w1, w2, w3, wd, s1, s2, s3, sd = ctx.saved_tensors
        need_x, n1, n2, n3, nd = ctx.needs_input_grad[:5]
        stride = ctx.stride
        S3 = G.contiguous(memory_format=torch.channels_last) if ctx.out_private else \
            _C.relu_bn_backward(G, out, None)[1]
        dw1 = dw2 = dw3 = dwd = dx = None
        lane = WgradLane(G.device, rows=G.shape[0] * G.shape[2] * G.shape[3])
        # the block's 3 - 4 weight gradients share ONE reduction launch over their split partial results (_WGRAD_BATCH)
        batch = _C.WgradBatch() if _WGRAD_BATCH else None

        group = _WgradGroup(lane, batch, [w for w, need in ((w3, n3), (w2, n2), (w1, n1), (wd, nd and wd is not None))
                                          if need])

        def wgrad(w, xin, gout, st, pd, sc):
            if group.take(w, xin, gout, st, pd, sc):
                return None
            return lane.run_into(w, lambda acc: _C.conv_wgrad(xin, gout, tuple(w.shape), st, pd, out_scale=sc, dw=acc,
                                                              accumulate=True, pending=batch),
                                 lambda: _C.conv_wgrad(xin, gout, tuple(w.shape), st, pd, out_scale=sc, pending=batch),
                                 xin, gout)

        if n3:
            dw3 = wgrad(w3, y2, S3, 1, 0, s3)
        S2 = _C.conv_forward(S3, _C.conv_weight_transpose(w3, s3), relu_mode=2, mask_ref=y2)
        if n2:
            dw2 = wgrad(w2, y1, S2, 1, 1, s2)
        if n1 or need_x:
            S1 = _C.conv_forward(S2, _C.conv_weight_transpose(w2, s2), pad=1, relu_mode=2, mask_ref=y1)
        if n1:
            dw1 = wgrad(w1, x, S1, stride, 0, s1)
        if nd and wd is not None:
            dwd = wgrad(wd, x, S3, stride, 0, sd)
        group.issue()
        lane.reduce_batch(batch)
        if need_x:
            gate = dict(relu_mode=2, mask_ref=x) if ctx.in_relu else {}
            hw = tuple(x.shape[2:])
            if wd is None:
                dx = _C.conv_forward(S1, _C.conv_weight_transpose(w1, s1), addend=S3, **gate)
            elif stride == 1:
                t = _C.conv_forward(S3, _C.conv_weight_transpose(wd, sd))
                dx = _C.conv_forward(S1, _C.conv_weight_transpose(w1, s1), addend=t, out=t, **gate)
            else:  # both 1x1 stride-s data gradients scatter onto the same (s*h, s*w) lattice of a zero map
                t = _C.conv_forward(S3, _C.conv_weight_transpose(wd, sd), out_spatial_stride=stride, out_hw=hw)
                dx = _C.conv_forward(S1, _C.conv_weight_transpose(w1, s1), addend=t, out=t,
                                     out_spatial_stride=stride, out_hw=hw, **gate)
        lane.join()
        return (dx, dw1, dw2, dw3, dwd) + (None,) * 11


class _DCNBottleneckFn(torch.autograd.Function):
    """A bottleneck whose 3x3 is a deformable convolution (vendored tree: modeling/backbone/resnet.py:286-312 with
    layers/misc.py:114-203 `DFConv2d` as conv2) as ONE autograd node, with the plain block's epilogue fusions:

    forward : y1 = relu(bn1(conv1(x)))                                          1 GEMM launch
              om = offset_conv(y1)          (3x3, bias; 18 | 27 channels padded to 20 | 28)   1 GEMM launch
              cols = deform_sample(y1, om)  (offsets / modulation logits read in place, sigmoid inside)   1 launch
              y2 = relu(bn2(cols (*) W2))   (1x1 GEMM over K = 9 * C, affine + ReLU in its epilogue)      1 GEMM launch
              out = relu(bn3(conv3(y2)) + identity)                              1 - 2 GEMM launches
    backward: S3, S2 as in _BottleneckFn (gates in the dgrad epilogues); gcols = S2 (*) (s2 W2)^T; the sampling backward
              gives d y1 (sampled path) and d om (written into the offset conv's padded output gradient in place); the
              offset conv's data gradient ADDS d y1 of the sampled path and applies y1's ReLU gate in its epilogue:
              S1 = [y1 > 0] * (dgrad_off(d om) + d y1).  No standalone affine / ReLU / ReLU-backward / concatenation
              kernel runs (the per-conv path used ~14 of them per block).
    The reference runs bn2 and the ReLU as separate modules after DFConv2d and materialises offset / mask slices."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, w_off, b_off, s1, b1, s2, b2, s3, b3, sd, bd, stride, in_relu, out_private,
                modulated, dg):
        cout2, cin2, kh, kw = w2.shape
        y1 = _C.conv_forward(x, w1, s1, b1, stride=stride, relu_mode=1)
        wo4, bo4 = _pad_out_channels(w_off, b_off)
        om = _C.conv_forward(y1, wo4, None, bo4, pad=kh // 2)
        cols = _C.deform_sample_forward_om(y1, om, kh, kw, 1, kh // 2, 1, dg, modulated)
        y2 = _C.conv_forward(cols, _as_1x1(w2), s2, b2, relu_mode=1)
        idn = x if wd is None else _C.conv_forward(x, wd, sd, bd, stride=stride)
        out = _C.conv_forward(y2, w3, s3, b3, addend=idn, relu_mode=1)
        ctx.conf = (stride, in_relu, out_private, modulated, dg, kh, kw, w_off.shape[0], wo4.shape[0])
        ctx.b_off = b_off
        ctx.save_for_backward(x, y1, om, cols, y2, out, w1, w2, w3, wd, w_off, s1, s2, s3, sd)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G):
        x, y1, om, cols, y2, out, w1, w2, w3, wd, w_off, s1, s2, s3, sd = ctx.saved_tensors
        stride, in_relu, out_private, modulated, dg, kh, kw, n_offch, n_offpad = ctx.conf
        need_x, n1, n2, n3, nd, n_off, n_boff = ctx.needs_input_grad[:7]
        cout2, cin2 = w2.shape[0], w2.shape[1]
        S3 = G.contiguous(memory_format=torch.channels_last) if out_private else _C.relu_bn_backward(G, out, None)[1]
        dw1 = dw2 = dw3 = dwd = dx = dwo = dbo = None
        lane = WgradLane(G.device, rows=G.shape[0] * G.shape[2] * G.shape[3])
        batch = _C.WgradBatch() if _WGRAD_BATCH else None

        # (the offset branch's gradient has rows padded beyond its 18 | 27 channels: not a member of the grouped launch)
        group = _WgradGroup(lane, batch, [w for w, need in ((w3, n3), (w2, n2), (w1, n1), (wd, nd and wd is not None))
                                          if need])

        def wgrad(w, xin, gout, st, pd, sc, shape=None):
            shape = tuple(w.shape) if shape is None else shape
            if w is not w_off and group.take(w, xin, gout, st, pd, sc, shape=shape):
                return None
            return lane.run_into(w, lambda acc: _C.conv_wgrad(xin, gout, shape, st, pd, out_scale=sc, dw=acc,
                                                              accumulate=True, pending=batch),
                                 lambda: _C.conv_wgrad(xin, gout, shape, st, pd, out_scale=sc, pending=batch),
                                 xin, gout)

        if n3:
            dw3 = wgrad(w3, y2, S3, 1, 0, s3)
        S2 = _C.conv_forward(S3, _C.conv_weight_transpose(w3, s3), relu_mode=2, mask_ref=y2)
        w2_1x1 = _as_1x1(w2)
        if n2:
            # [Cout][kh][kw][Cin] IS the 1x1 weight [Cout][K = (tap, ci)]: the gradient lands in w2's own layout
            dw2 = wgrad(w2, cols, S2, 1, 0, s2, shape=(cout2, kh * kw * cin2, 1, 1))
        gcols = _C.conv_forward(S2, _C.conv_weight_transpose(w2_1x1, s2))
        gy1, gom = _C.deform_sample_backward_om(y1, om, gcols, kh, kw, 1, kh // 2, 1, dg, modulated)
        if n_off:
            # gom's rows are padded to a multiple of four channels; the weight gradient keeps the parameter's own shape
            # (conv_wgrad reads the padded rows, dadet_conv_wgrad_partials_ld) and lands in its gradient buffer like every
            # other weight of the block
            dwo = wgrad(w_off, y1, gom, 1, kh // 2, None)
        if n_boff:
            dbo = streams.bias_grad(ctx.b_off, gom, cols=n_offch)
        # d y1 = [y1 > 0] * (sampled path + offset-conv path); the transposed weights carry zero columns for the padding
        S1 = _C.conv_forward(gom, _C.conv_weight_transpose(w_off, cout_pad=n_offpad), pad=kh // 2, addend=gy1, out=gy1,
                             relu_mode=2, mask_ref=y1)
        if n1:
            dw1 = wgrad(w1, x, S1, stride, 0, s1)
        if nd and wd is not None:
            dwd = wgrad(wd, x, S3, stride, 0, sd)
        group.issue()
        lane.reduce_batch(batch)
        if need_x:
            gate = dict(relu_mode=2, mask_ref=x) if in_relu else {}
            hw = tuple(x.shape[2:])
            if wd is None:
                dx = _C.conv_forward(S1, _C.conv_weight_transpose(w1, s1), addend=S3, **gate)
            elif stride == 1:
                t = _C.conv_forward(S3, _C.conv_weight_transpose(wd, sd))
                dx = _C.conv_forward(S1, _C.conv_weight_transpose(w1, s1), addend=t, out=t, **gate)
            else:
                t = _C.conv_forward(S3, _C.conv_weight_transpose(wd, sd), out_spatial_stride=stride, out_hw=hw)
                dx = _C.conv_forward(S1, _C.conv_weight_transpose(w1, s1), addend=t, out=t,
                                     out_spatial_stride=stride, out_hw=hw, **gate)
        lane.join()
        if dw2 is not None:     # returned to autograd (no persistent gradient buffer): back in w2's logical shape
            dw2 = torch.as_strided(dw2, (cout2, cin2, kh, kw), (kh * kw * cin2, 1, kw * cin2, cin2))
        return (dx, dw1, dw2, dw3, dwd, dwo, dbo) + (None,) * 13


def _as_1x1(weight):
    """[Cout,Cin,kh,kw] (physically [Cout][kh][kw][Cin]) -> the same bytes as [Cout, kh*kw*Cin, 1, 1], K = (tap, ci)"""
    cout, cin, kh, kw = weight.shape
    return weight.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(cout, kh * kw * cin, 1, 1)


class _PaddedWeights(object):
    """weights / biases with zero output channels appended up to a multiple of 4 (16-byte rows for the kernels that
    contract over them), kept in persistent buffers.  Padding on the spot cost 5 launches per deformable block and forward
    pass (two F.pad = fill + copy each, one layout copy): 150 launches of ~6 us per step of R-101-FPN-DCN.  Here every
    registered parameter is refreshed by ONE multi-tensor copy, the first time one of them is asked for after the weights
    changed (autograd version of the parameter, or the weight epoch that the fused optimizer bumps)."""

    def __init__(self):
        self.entries = {}          # id(weight) -> dict(ref, wp, bp, bias_ref)
        self.stamp = None

    def _stamp(self):
        ws = [e["w"]() for e in self.entries.values()]
        bs = [e["b"]() for e in self.entries.values() if e["b"] is not None]
        if any(t is None for t in ws) or any(t is None for t in bs):
            return None                                    # a parameter is gone: rebuild
        return (_C.weight_epoch(), tuple(t._version for t in ws), tuple(t._version for t in bs))

    def _fresh(self):
        stamp = self._stamp()
        if stamp is not None and stamp == self.stamp:
            return
        for k in [k for k, e in self.entries.items() if e["w"]() is None or (e["b"] is not None and e["b"]() is None)]:
            del self.entries[k]
        dst, src = [], []
        for e in self.entries.values():
            w = e["w"]()
            dst.append(e["wp"][:w.shape[0]])
            src.append(w.detach())
            if e["b"] is not None:
                dst.append(e["bp"][:w.shape[0]])
                src.append(e["b"]().detach())
        if dst:
            torch._foreach_copy_(dst, src)
        self.stamp = self._stamp()

    def get(self, weight, bias, pad):
        import weakref

        e = self.entries.get(id(weight))
        if e is None or e["w"]() is not weight or (bias is not None) != (e["b"] is not None) \
                or (bias is not None and e["b"]() is not bias):
            cout, cin, kh, kw = weight.shape
            wp = torch.zeros((cout + pad, cin, kh, kw), dtype=weight.dtype, device=weight.device).contiguous(
                memory_format=torch.channels_last)
            bp = torch.zeros(cout + pad, dtype=bias.dtype, device=bias.device) if bias is not None else None
            wp._dadet_amax_like = weakref.ref(weight)     # same largest magnitude as the parameter (amax.py, mode 4)
            self.entries[id(weight)] = dict(w=weakref.ref(weight), b=weakref.ref(bias) if bias is not None else None,
                                            wp=wp, bp=bp)
            self.stamp = None
        self._fresh()
        e = self.entries[id(weight)]
        return e["wp"], e["bp"]


_PADDED = _PaddedWeights()
# (measured against the first fused version — weights padded on the spot, their gradient computed on the padded shape,
# reduced at once, sliced and handed to autograd: R-101-FPN-DCN 65.2 -> 64.0 ms per step on one box)


def _pad_out_channels(weight, bias):
    """zero output channels appended up to a multiple of 4 (16-byte rows for the kernels that contract over them)"""
    pad = (-weight.shape[0]) % 4
    if pad == 0:
        return weight, bias
    if weight.is_cuda and weight.is_leaf and weight.requires_grad and (bias is None or bias.is_leaf):
        return _PADDED.get(weight, bias, pad)
    w = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, pad)).contiguous(memory_format=torch.channels_last)
    return w, (F.pad(bias, (0, pad)) if bias is not None else None)


# DADET_DCN_FUSED=0: DCN bottlenecks on the per-conv autograd path (DFConv2d module + standalone bn2 / ReLU kernels)
_DCN_FUSED = __import__("os").environ.get("DADET_DCN_FUSED", "1") == "1"


class Bottleneck(nn.Module):
    """1x1 (carries the stride when stride_in_1x1) -> 3x3 -> 1x1, FrozenBN after each, projection shortcut
    when the channel count changes (resnet.py:227-314)."""

    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1, stride,
                 dilation, norm_func, dcn_config=None):
        super(Bottleneck, self).__init__()
        dcn_config = dcn_config or {}
        if num_groups != 1 or dilation != 1:
            raise NotImplementedError("grouped / dilated bottlenecks are outside the DA Faster R-CNN path")
        self.downsample = None
        if in_channels != out_channels:
            self.downsample = nn.Sequential(
                Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False),
                norm_func(out_channels))
            nn.init.kaiming_uniform_(self.downsample[0].weight, a=1)
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False)
        self.bn1 = norm_func(bottleneck_channels)
        self.with_dcn = bool(dcn_config.get("stage_with_dcn", False))
        if self.with_dcn:
            # vendored tree only (tools/cityscapes/maskrcnn_benchmark/modeling/backbone/resnet.py:286-300)
            from ...layers.dcn import DFConv2d
            self.conv2 = DFConv2d(bottleneck_channels, bottleneck_channels,
                                  with_modulated_dcn=dcn_config.get("with_modulated_dcn", False), kernel_size=3,
                                  stride=stride_3x3, groups=num_groups, dilation=dilation,
                                  deformable_groups=dcn_config.get("deformable_groups", 1), bias=False)
        else:
            self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3,
                                padding=1, bias=False)
        self.bn2 = norm_func(bottleneck_channels)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False)
        self.bn3 = norm_func(out_channels)
        for l in (self.conv1, self.conv3) + (() if self.with_dcn else (self.conv2,)):
            nn.init.kaiming_uniform_(l.weight, a=1)

    def forward(self, x, in_relu=False, out_private=False, stride=None):
        """in_relu / out_private: structural promises made by _Stage (see _BottleneckFn); a direct call makes none.
        stride: overrides the stride of conv1 and the shortcut (see input_is_strided_1x1) — fused path only"""
        if self.with_dcn:
            assert stride is None
            if self._dcn_fusable(x):
                c2 = self.conv2
                wd = sd = bd = None
                if self.downsample is not None:
                    wd, (sd, bd) = self.downsample[0].weight, self.downsample[1].folded()
                return _DCNBottleneckFn.apply(x, self.conv1.weight, c2.conv.weight, self.conv3.weight, wd,
                                              c2.offset.weight, c2.offset.bias, *self.bn1.folded(), *self.bn2.folded(),
                                              *self.bn3.folded(), sd, bd, self.conv1.stride[0], in_relu, out_private,
                                              c2.with_modulated_dcn, c2.conv.deformable_groups)
            return self._forward_dcn(x)
        if self.conv2.stride[0] == 1:   # STRIDE_IN_1X1: the stride (if any) sits in conv1 and the shortcut
            if x.shape[0] == 0 and stride is None:
                return self._forward_per_conv(x)
            wd = sd = bd = None
            if self.downsample is not None:
                wd, (sd, bd) = self.downsample[0].weight, self.downsample[1].folded()
            return _BottleneckFn.apply(x, self.conv1.weight, self.conv2.weight, self.conv3.weight, wd,
                                       *self.bn1.folded(), *self.bn2.folded(), *self.bn3.folded(), sd, bd,
                                       self.conv1.stride[0] if stride is None else stride, in_relu, out_private)
        assert stride is None
        return self._forward_per_conv(x)

    def input_is_strided_1x1(self):
        """the block reads its input ONLY through stride-s 1x1 convolutions (conv1 and a projection shortcut), i.e. only
        the pixels (i * s, j * s): -> s, else 1"""
        s = self.conv1.stride[0]
        if (self.uses_fused_path() and s > 1 and self.downsample is not None and self.conv1.kernel_size == (1, 1)
                and self.downsample[0].kernel_size == (1, 1) and self.downsample[0].stride[0] == s):
            return s
        return 1

    def uses_fused_path(self):
        if self.with_dcn:
            return self._dcn_fusable(None)
        return self.conv2.stride[0] == 1

    def _dcn_fusable(self, x):
        """the fused node covers what the DA configurations use: 3x3 deformable conv of stride 1 (STRIDE_IN_1X1) without
        bias, one group; anything else takes the per-conv path"""
        c = self.conv2.conv
        ok = (_DCN_FUSED and c.stride == 1 and c.dilation == 1 and c.groups == 1 and c.bias is None
              and c.kernel_size == (3, 3) and c.in_channels % (4 * c.deformable_groups) == 0)
        return ok and (x is None or (x.is_cuda and x.shape[0] > 0))

    def _forward_dcn(self, x):
        """conv2 is a DFConv2d (offset conv + deformable conv): bn2 + ReLU run as the standalone affine kernel"""
        out = self.conv1(x, *self.bn1.folded(), relu=True)
        out = F.relu(self.bn2(self.conv2(out)))
        identity = x if self.downsample is None else self.downsample[0](x, *self.downsample[1].folded())
        return self.conv3(out, *self.bn3.folded(), residual=identity, relu=True)

    def _forward_per_conv(self, x):
        out = self.conv1(x, *self.bn1.folded(), relu=True)
        out = self.conv2(out, *self.bn2.folded(), relu=True)
        identity = x if self.downsample is None else self.downsample[0](x, *self.downsample[1].folded())
        # relu(bn3(conv3(out)) + identity) in one epilogue
        return self.conv3(out, *self.bn3.folded(), residual=identity, relu=True)


class BottleneckWithFixedBatchNorm(Bottleneck):
    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups=1, stride_in_1x1=True,
                 stride=1, dilation=1, dcn_config=None):
        super(BottleneckWithFixedBatchNorm, self).__init__(
            in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1, stride, dilation,
            norm_func=FrozenBatchNorm2d, dcn_config=dcn_config)


class BaseStem(nn.Module):
    """conv 7x7/2 (3 -> 64) + FrozenBN + ReLU + maxpool 3x3/2 (resnet.py:317-336)"""

    def __init__(self, cfg, norm_func):
        super(BaseStem, self).__init__()
        out_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        self.conv1 = Conv2d(3, out_channels, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_func(out_channels)
        nn.init.kaiming_uniform_(self.conv1.weight, a=1)
        self._w4 = None

    def _padded_weight(self):
        """[64,3,7,7] -> [64,4,7,8] (zero 4th channel and 8th column) in channels_last"""
        w = self.conv1.weight
        if w.requires_grad and torch.is_grad_enabled():
            return F.pad(w, (0, 1, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last)
        key = (w._version, w.device)
        if self._w4 is None or self._w4[0] != key:
            with torch.no_grad():
                self._w4 = (key, F.pad(w, (0, 1, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last))
        return self._w4[1]

    def forward(self, x):
        N, C, H, W = x.shape
        assert C == 3, "the stem expects BGR images [N,3,H,W]"
        x4 = _C.nchw3_to_nhwc4(x)
        out_size = _C.conv_out_size(H, W, 7, 7, 2, 3)
        scale, shift = self.bn1.folded()
        y = conv2d_affine_act(x4, self._padded_weight(), scale, shift, None, 2, 3, True, out_size)
        if y.requires_grad:
            raise NotImplementedError("max-pool backward: the stem is frozen in every DA configuration "
                                      "(MODEL.BACKBONE.FREEZE_CONV_BODY_AT >= 1)")
        return _C.maxpool3x3s2(y)


class StemWithFixedBatchNorm(BaseStem):
    def __init__(self, cfg):
        super(StemWithFixedBatchNorm, self).__init__(cfg, norm_func=FrozenBatchNorm2d)


def _make_stage(block, in_channels, bottleneck_channels, out_channels, block_count, num_groups,
                stride_in_1x1, first_stride, dilation=1, dcn_config=None):
    blocks, stride = [], first_stride
    for _ in range(block_count):
        blocks.append(block(in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1, stride,
                            dilation=dilation, dcn_config=dcn_config))
        stride = 1
        in_channels = out_channels
    return _Stage(*blocks)


class _Stage(nn.Sequential):
    """nn.Sequential of bottlenecks (same parameter names) that tells each block what its neighbours are: every
    block but the first consumes a ReLU output, every block but the last feeds only its successor."""

    input_is_relu = False   # set by the owner when the stage input is itself a ReLU output (ResNet.layer2..4)

    def forward(self, x, first_stride=None):
        n = len(self)
        blocks = list(self)
        for i, block in enumerate(blocks):
            # "private" only if the consumer really is a fused block that gates what it returns
            private = i < n - 1 and blocks[i + 1].uses_fused_path()
            if i == 0 and first_stride is not None:
                x = block(x, in_relu=self.input_is_relu, out_private=private, stride=first_stride)
            else:
                x = block(x, in_relu=(i > 0 or self.input_is_relu), out_private=private)
        return x


class ResNet(nn.Module):
    def __init__(self, cfg):
        super(ResNet, self).__init__()
        stem_module = _STEM_MODULES[cfg.MODEL.RESNETS.STEM_FUNC]
        stage_specs = _STAGE_SPECS[cfg.MODEL.BACKBONE.CONV_BODY]
        block = _TRANSFORMATION_MODULES[cfg.MODEL.RESNETS.TRANS_FUNC]
        self.stem = stem_module(cfg)
        num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
        in_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        stage2_bottleneck = num_groups * cfg.MODEL.RESNETS.WIDTH_PER_GROUP
        stage2_out = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
        self.stages, self.return_features = [], {}
        for spec in stage_specs:
            name = "layer" + str(spec.index)
            factor = 2 ** (spec.index - 1)
            out_channels = stage2_out * factor
            self.add_module(name, _make_stage(block, in_channels, stage2_bottleneck * factor, out_channels,
                                              spec.block_count, num_groups, cfg.MODEL.RESNETS.STRIDE_IN_1X1,
                                              first_stride=int(spec.index > 1) + 1,
                                              dcn_config={
                                                  "stage_with_dcn": cfg.MODEL.RESNETS.STAGE_WITH_DCN[spec.index - 1],
                                                  "with_modulated_dcn": cfg.MODEL.RESNETS.WITH_MODULATED_DCN,
                                                  "deformable_groups": cfg.MODEL.RESNETS.DEFORMABLE_GROUPS}))
            in_channels = out_channels
            # layer1 consumes max-pooled ReLU output (frozen, no data gradient); layer2.. consume a block's ReLU
            getattr(self, name).input_is_relu = spec.index > 1
            self.stages.append(name)
            self.return_features[name] = spec.return_features
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)

    def _freeze_backbone(self, freeze_at):
        """stage 0 = stem, stage i = layer{i} (resnet.py:127-136)"""
        for stage_index in range(max(freeze_at, 0)):
            m = self.stem if stage_index == 0 else getattr(self, "layer" + str(stage_index))
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        outputs = []
        x = self.stem(x)
        for name in self.stages:
            x = getattr(self, name)(x)
            if self.return_features[name]:
                outputs.append(x)
        return outputs


class ResNetHead(nn.Module):
    """res5 as an ROI head: stages of bottlenecks applied to pooled ROI features (resnet.py:148-194)"""

    def __init__(self, block_module, stages, num_groups=1, width_per_group=64, stride_in_1x1=True,
                 stride_init=None, res2_out_channels=256, dilation=1):
        super(ResNetHead, self).__init__()
        factor = 2 ** (stages[0].index - 1)
        out_channels = res2_out_channels * factor
        in_channels = out_channels // 2
        bottleneck_channels = num_groups * width_per_group * factor
        block = _TRANSFORMATION_MODULES[block_module]
        self.stages = []
        stride = stride_init
        for stage in stages:
            name = "layer" + str(stage.index)
            if not stride:
                stride = int(stage.index > 1) + 1
            self.add_module(name, _make_stage(block, in_channels, bottleneck_channels, out_channels,
                                              stage.block_count, num_groups, stride_in_1x1, first_stride=stride,
                                              dilation=dilation))
            stride = None
            self.stages.append(name)

    def forward(self, x, first_stride=None):
        for i, stage in enumerate(self.stages):
            x = getattr(self, stage)(x, first_stride=first_stride) if (i == 0 and first_stride is not None) \
                else getattr(self, stage)(x)
        return x

    def input_bin_stride(self):
        """s > 1: the head reads only the pixels (i * s, j * s) of its input (Bottleneck.input_is_strided_1x1)"""
        return getattr(self, self.stages[0])[0].input_is_strided_1x1()


_TRANSFORMATION_MODULES = Registry({"BottleneckWithFixedBatchNorm": BottleneckWithFixedBatchNorm})
_STEM_MODULES = Registry({"StemWithFixedBatchNorm": StemWithFixedBatchNorm})
_STAGE_SPECS = Registry({
    "R-50-C4": ResNet50StagesTo4, "R-50-C5": ResNet50StagesTo5,
    "R-101-C4": ResNet101StagesTo4, "R-101-C5": ResNet101StagesTo5,
    "R-50-FPN": ResNet50FPNStagesTo5, "R-50-FPN-RETINANET": ResNet50FPNStagesTo5,
    "R-101-FPN": ResNet101FPNStagesTo5, "R-101-FPN-RETINANET": ResNet101FPNStagesTo5,
    "R-152-FPN": ResNet152FPNStagesTo5,
})
