"""Region proposal network (reference: maskrcnn_benchmark/modeling/rpn/rpn.py:13-138)."""
import os

import torch
from torch import nn

from ...layers import Conv2d, conv1x1_multi
from ...layers.misc import rpn_head_loss_rows, rpn_head_loss_rows_pyramid
from .. import registry
from ..box_coder import BoxCoder
from .anchor_generator import make_anchor_generator
from .inference import make_rpn_postprocessor
from .loss import make_rpn_loss_evaluator
from ..elision import elision_enabled, leading_source_images
from ...utils.streams import record, side_section, side_stream



@registry.RPN_HEADS.register("SingleConvRPNHead")
class RPNHead(nn.Module):
    """3x3 conv + ReLU, then 1x1 objectness (A) and 1x1 box deltas (4A) (rpn.py:13-46).  The bias + ReLU is
    in the 3x3 GEMM's epilogue and the two 1x1 convs run as ONE GEMM of width 5A (padded to a multiple of 4)."""

    def __init__(self, cfg, in_channels, num_anchors):
        super(RPNHead, self).__init__()
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls_logits = Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):
            torch.nn.init.normal_(l.weight, std=0.01)
            torch.nn.init.constant_(l.bias, 0)

    keep_hidden = False     # True: forward also leaves the 3x3 conv's ReLU output of every level in self.hidden

    def forward(self, x):
        logits, bbox_reg = [], []
        self.hidden = []
        for feature in x:
            t = self.conv(feature, relu=True)
            if self.keep_hidden:
                self.hidden.append(t)
            cls, box = conv1x1_multi(t, [self.cls_logits.weight, self.bbox_pred.weight],
                                     [self.cls_logits.bias, self.bbox_pred.bias])
            logits.append(cls)
            bbox_reg.append(box)
        return logits, bbox_reg


# the single-conv head's backward over the sampled anchors' rows only (layers.misc._RPNHeadLossRows); 0: autograd's dense one
_ROW_BACKWARD = __import__("os").environ.get("DADET_RPN_ROW_BACKWARD", "1") == "1"


class _InjectGrad(torch.autograd.Function):
    """identity whose backward adds gradients computed earlier: `g` the RPN branch's (RPNModule.early_backward; it may
    cover the leading images only), `e` another early branch's (the image-level DA head's), ready once `event` fired"""

    @staticmethod
    def forward(ctx, x, g, e, event):
        ctx.event = event
        ctx.save_for_backward(*[t for t in (g, e) if t is not None])
        ctx.has = (g is not None, e is not None)
        from ... import amax
        return amax.carry(x.view_as(x), x)

    @staticmethod
    def backward(ctx, grad_out):
        saved = list(ctx.saved_tensors)
        g = saved.pop(0) if ctx.has[0] else None
        e = saved.pop(0) if ctx.has[1] else None
        if ctx.event is not None:
            torch.cuda.current_stream(grad_out.device).wait_event(ctx.event)
        out = grad_out + e if e is not None else None
        if g is not None:
            if out is None and g.shape[0] == grad_out.shape[0]:
                out = grad_out + g
            else:
                out = grad_out.clone() if out is None else out
                out[:g.shape[0]] += g
        return out, None, None, None


class RPNModule(torch.nn.Module):
    """early_backward (opt-in, set by engine.trainer.train_step's schedule): the RPN branch is cut out of the main
    autograd graph and its backward (RPN head dgrad / wgrad, ~4 ms of MFMA work at 1024x2048) is queued as soon as
    the RPN losses exist — i.e. right in front of the box head's proposal sampling, a stretch of a few hundred
    tiny launches and host synchronisations during which the GPU would otherwise idle.  The gradient w.r.t. the
    feature maps is handed back to the main graph through `bridge_features`, so parameter gradients are the same
    sums as without the option.  Requirements on the caller: gradients are zeroed BEFORE the forward pass, the RPN
    losses enter the total loss with weight 1, and something downstream (the box head) back-propagates into the
    features."""

    early_backward = False
    # Set by the detector for ONE call (training): the number of leading images whose proposals somebody reads; None =
    # all.  The images behind them (target-domain images when no loss reads instance-level features, the auxiliary
    # image of a triplet batch: generalized_rcnn.py:100 passes proposals[0:2] on) get no RPN head pass and no proposal
    # selection on the overlapped schedule — their entry in the returned list is None.
    live_images = None
    # callable run ONCE right after the early backward has been queued (the detector issues the image-level DA branch
    # there); left in place when the call took a path without an early backward
    after_early_backward = None

    def __init__(self, cfg):
        super(RPNModule, self).__init__()
        self.cfg = cfg.clone()
        anchor_generator = make_anchor_generator(cfg)
        in_channels = cfg.MODEL.BACKBONE.OUT_CHANNELS
        head = registry.RPN_HEADS[cfg.MODEL.RPN.RPN_HEAD](cfg, in_channels,
                                                         anchor_generator.num_anchors_per_location()[0])
        rpn_box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.anchor_generator = anchor_generator
        self.head = head
        self.box_selector_train = make_rpn_postprocessor(cfg, rpn_box_coder, is_train=True)
        self.box_selector_test = make_rpn_postprocessor(cfg, rpn_box_coder, is_train=False)
        self.loss_evaluator = make_rpn_loss_evaluator(cfg, rpn_box_coder)
        self.proposals_ready = None  # event: proposals exist (recorded before the RPN losses / early backward)
        self._feature_grads = None
        self.inputs_ready = None     # event recorded by the detector before the backbone is queued

    def forward(self, images, features, targets=None):
        self._feature_grads = None
        live, self.live_images = self.live_images, None
        early = (self.training and self.early_backward and torch.is_grad_enabled() and not self.cfg.MODEL.RPN_ONLY
                 and all(f.requires_grad for f in features))
        if early and features[0].is_cuda and len(features) == 1 and elision_enabled():
            return self._forward_train_overlapped(images, features, targets, live)
        from ... import amax
        # (a detached alias is a new tensor object: its map's largest magnitude is handed on by hand, amax.py)
        head_in = [amax.carry(f.detach().requires_grad_(True), f) for f in features] if early else features
        if (early and features[0].is_cuda and len(features) > 1 and elision_enabled() and _ROW_BACKWARD
                and isinstance(self.head, RPNHead)):
            # feature pyramid: the shared head runs WITHOUT autograd on every level and keeps its hidden activations; its
            # backward is hand-written over the sampled anchors' rows, level by level (_finish_overlapped) — the dense
            # backward of the 3x3 convolution over all five maps was 0.8 TFLOP per step of R-101-FPN for <= 512 non-zero rows
            with torch.no_grad():
                self.head.keep_hidden = True
                objectness, rpn_box_regression = self.head(head_in)
                hidden, self.head.hidden, self.head.keep_hidden = list(self.head.hidden), [], False
            anchors = self.anchor_generator(images, features)
            return self._finish_overlapped(anchors, head_in, objectness, rpn_box_regression, list(objectness),
                                           list(rpn_box_regression), targets, len(targets), hidden)
        objectness, rpn_box_regression = self.head(head_in)
        anchors = self.anchor_generator(images, features)
        if not self.training:
            return self._forward_test(anchors, objectness, rpn_box_regression)
        if not (early and objectness[0].is_cuda):
            return self._forward_train(anchors, objectness, rpn_box_regression, targets)
        return self._finish_overlapped(anchors, head_in, objectness, rpn_box_regression,
                                       [o.detach() for o in objectness], [r.detach() for r in rpn_box_regression],
                                       targets, len(targets))

    def _forward_train_overlapped(self, images, features, targets, live):
        """single-level overlapped schedule with the head's work restricted to the images that need it.  The RPN losses
        are taken on anchors of source-domain images only (rpn/loss.py:57-98 labels nothing else; they come first in the
        batch), so the gradient of every other image's objectness / regression map is identically zero: the head runs
        WITH autograd on the leading source images only — its backward GEMMs then cover those rows and nothing else —
        without autograd on the other images whose proposals are read, and not at all on the rest (`live_images`)."""
        f = features[0]
        n_img = f.shape[0]
        n_grad = leading_source_images(targets) or n_img      # unusual batch order: no restriction
        n_live = n_img if live is None else max(n_grad, min(int(live), n_img))
        from ... import amax
        # (a slice is bounded by the whole map's largest magnitude: amax.py)
        head_in = [amax.carry(f[:n_grad].detach().requires_grad_(True), f)]
        hidden = None
        if _ROW_BACKWARD and isinstance(self.head, RPNHead):
            # the head's backward is hand-written over the sampled rows (layers.misc._RPNHeadLossRows): no autograd graph
            with torch.no_grad():
                self.head.keep_hidden = True
                objectness, rpn_box_regression = self.head(head_in)
                hidden, self.head.hidden, self.head.keep_hidden = self.head.hidden[0], [], False
        else:
            objectness, rpn_box_regression = self.head(head_in)
        sel_obj, sel_reg = [objectness[0].detach()], [rpn_box_regression[0].detach()]
        if n_live > n_grad:
            with torch.no_grad():
                o, r = self.head([amax.carry(f[n_grad:n_live], f)])
            sel_obj, sel_reg = [torch.cat([sel_obj[0], o[0]], dim=0)], [torch.cat([sel_reg[0], r[0]], dim=0)]
        anchors = self.anchor_generator(images, features)
        return self._finish_overlapped(anchors, head_in, objectness, rpn_box_regression, sel_obj, sel_reg, targets,
                                       n_live, hidden)

    def _finish_overlapped(self, anchors, head_in, objectness, rpn_box_regression, sel_obj, sel_reg, targets, n_live,
                           hidden=None):
        # overlapped schedule: losses + the RPN branch's backward (+ the image-level DA head through the hook) go to the
        # compute stream first; proposal selection (sort, decode, single-workgroup NMS sweeps) then runs on the side stream
        # underneath them, and the box head's sampling continues there (ROIBoxHead.forward).
        # (issuing the selection chain BEFORE the losses was measured in round 3 and removed: R-50-C4 19.00 / 19.04 vs
        # 19.07 / 18.99 ms, R-101-FPN-DCN 64.8 / 65.5 -> 68.7 / 70.0 ms — the host then sits in the per-level round trips
        # with nothing queued on the compute stream)
        dev = objectness[0].device
        prep = self._prepare_loss_targets(anchors, targets)
        main = torch.cuda.current_stream(dev)
        head_done = main.record_event()
        side = side_stream(dev)

        def select():
            side.wait_event(head_done)
            # the head's maps were allocated on the compute stream and are released when forward() returns: without this
            # the caching allocator may hand their blocks to a later compute-stream allocation while the side stream still
            # reads them
            record([sel_obj, sel_reg], side)
            with torch.cuda.stream(side), torch.no_grad():
                out = self.box_selector_train(anchors[:n_live], sel_obj, sel_reg, targets[:n_live])
                self.proposals_ready = side.record_event()
            return out

        if isinstance(hidden, list) and prep["sampled_inds"].numel() > 0:
            # one launch chain per pyramid level over the SAME sampled rows (rows of the other levels are zero); every
            # level normalises by the same count, so the levels' losses add up to rpn/loss.py:125-143
            h = self.head
            per_image = sum(int(o.shape[1] * o.shape[2] * o.shape[3]) for o in objectness)
            levels, off = [], 0
            for o in objectness:
                cnt = int(o.shape[1] * o.shape[2] * o.shape[3])
                levels.append((per_image, off, cnt))
                off += cnt
            loss_objectness, loss_rpn_box_reg = rpn_head_loss_rows_pyramid(
                h.conv.weight, h.conv.bias, h.cls_logits.weight, h.cls_logits.bias, h.bbox_pred.weight, h.bbox_pred.bias,
                prep["sampled_inds"], prep["labels_sampled"], int(prep["pos_inds"].numel()),
                prep["regression_targets_pos"], 1.0 / 9, levels, *head_in, *hidden, *objectness, *rpn_box_regression)
        elif hidden is not None and not isinstance(hidden, list) and prep["sampled_inds"].numel() > 0:
            h = self.head
            loss_objectness, loss_rpn_box_reg = rpn_head_loss_rows(
                head_in[0], h.conv.weight, h.conv.bias, h.cls_logits.weight, h.cls_logits.bias, h.bbox_pred.weight,
                h.bbox_pred.bias, hidden, objectness[0], rpn_box_regression[0], prep["sampled_inds"],
                prep["labels_sampled"], int(prep["pos_inds"].numel()), prep["regression_targets_pos"], 1.0 / 9)
        else:
            if hidden is not None:      # nothing sampled: the dense path's autograd graph does not exist — build it
                objectness, rpn_box_regression = self.head(head_in)
            loss_objectness, loss_rpn_box_reg = self.loss_evaluator.finish(objectness, rpn_box_regression, prep)
        torch.autograd.backward([loss_objectness + loss_rpn_box_reg])
        self._feature_grads = [f.grad for f in head_in]
        hook, self.after_early_backward = self.after_early_backward, None
        if hook is not None:
            hook()
        boxes = select()
        record(boxes, main)
        boxes = list(boxes) + [None] * (len(targets) - n_live)
        return boxes, {"loss_objectness": loss_objectness.detach(), "loss_rpn_box_reg": loss_rpn_box_reg.detach()}

    def bridge_features(self, features, extra=None, extra_ready=None):
        """features whose backward also delivers the RPN branch's gradient (no-op unless early_backward ran);
        `extra`: gradients of another branch that ran its backward early (the image-level DA head), same layout, valid
        on the consuming stream once the event `extra_ready` has fired (the wait is placed in the backward pass)"""
        grads, self._feature_grads = self._feature_grads, None
        if grads is None and extra is None:
            return features
        grads = grads if grads is not None else [None] * len(features)
        extra = extra if extra is not None else [None] * len(features)
        return [_InjectGrad.apply(f, g, e, extra_ready) for f, g, e in zip(features, grads, extra)]

    def _prepare_loss_targets(self, anchors, targets):
        """RPNLossComputation.prepare on a side stream: its ~150 small launches and host synchronisations overlap
        the backbone instead of stalling behind it (program order — hence the order of the random draws — is
        unchanged: it still runs before the box head's sampler)."""
        dev = anchors[0][0].bbox.device
        ready, self.inputs_ready = self.inputs_ready, None
        if not self.anchor_generator.last_call_was_cached:
            ready = None                  # first step: the anchors were just built on the compute stream
        with side_section(dev, after=ready) as done, torch.no_grad():
            prep = self.loss_evaluator.prepare(anchors, targets)
            done(prep)
        return prep

    def _forward_train(self, anchors, objectness, rpn_box_regression, targets):
        prep = self._prepare_loss_targets(anchors, targets)
        if self.cfg.MODEL.RPN_ONLY:
            boxes = anchors
        else:
            with torch.no_grad():
                boxes = self.box_selector_train(anchors, objectness, rpn_box_regression, targets)
            if objectness[0].is_cuda:     # the box head's sampling (side stream) may start from here
                self.proposals_ready = torch.cuda.current_stream(objectness[0].device).record_event()
        loss_objectness, loss_rpn_box_reg = self.loss_evaluator.finish(objectness, rpn_box_regression, prep)
        return boxes, {"loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg}

    def _forward_test(self, anchors, objectness, rpn_box_regression):
        boxes = self.box_selector_test(anchors, objectness, rpn_box_regression)
        if self.cfg.MODEL.RPN_ONLY:
            inds = [b.get_field("objectness").sort(descending=True)[1] for b in boxes]
            boxes = [b[ind] for b, ind in zip(boxes, inds)]
        return boxes, {}


def build_rpn(cfg):
    if cfg.MODEL.RETINANET_ON:
        raise NotImplementedError("RetinaNet is outside the DA Faster R-CNN path (SURVEY.md section 2.1 row 8)")
    return RPNModule(cfg)
