"""Domain-adaptation heads (reference: maskrcnn_benchmark/modeling/da_heads/da_heads.py:12-440).

Module / parameter names follow the reference (`imghead.conv1_da`, `imghead.conv2_da`, `inshead.fc{1,2,3}_da`).
Execution differs: the image-level classifier is evaluated once by the fused kernels (fused.py) instead of
two (three with AdvGRL) full passes; the instance-level classifier keeps the reference's separate passes
because each pass draws its own dropout masks (da_heads.py:61-68).
"""
import contextlib

import torch
import torch.nn.functional as F
from torch import nn

from ...layers import Conv2d, GradientScalarLayer, global_avg_pool, linear
from ...layers.misc import gradient_scalar
from ...utils import rng
from ...utils.streams import other_stream
from .fused import INS_DROPOUT_P, da_image_head, da_instance_head
from .loss import TripletMargins, da_consist_loss, da_ins_loss, image_domain_labels


_EARLY_DA = True   # A/B switch of early_image_level


class DAImgHead(nn.Module):
    """1x1 conv C->512, ReLU, 1x1 conv 512->1 (da_heads.py:12-37)"""

    def __init__(self, in_channels):
        super(DAImgHead, self).__init__()
        self.conv1_da = Conv2d(in_channels, 512, kernel_size=1, stride=1)
        self.conv2_da = Conv2d(512, 1, kernel_size=1, stride=1)
        for l in (self.conv1_da, self.conv2_da):
            torch.nn.init.normal_(l.weight, std=0.001)
            torch.nn.init.constant_(l.bias, 0)

    def fused(self, feature, labels, w_adv, w_cst):
        """-> (mean BCE, per-image mean sigmoid, logits [N,1,H,W]) for one feature level"""
        return da_image_head(feature, self.conv1_da, self.conv2_da, labels, w_adv, w_cst)

    def forward(self, x):
        """plain logits per level (reference signature); no loss fusion"""
        out = []
        for feature in x:
            t = self.conv1_da(feature, relu=True)
            out.append(linear_map(t, self.conv2_da))
        return out


def linear_map(t, conv):
    """1x1 conv with a single output channel through the padded-linear path"""
    N, C, H, W = t.shape
    flat = t.permute(0, 2, 3, 1).reshape(-1, C)
    y = linear(flat, conv.weight.reshape(conv.weight.shape[0], C), conv.bias)
    return y.reshape(N, H, W, -1).permute(0, 3, 1, 2)


class DAInsHead(nn.Module):
    """fc 2048->1024, ReLU, dropout, fc 1024->1024, ReLU, dropout, fc 1024->1 (da_heads.py:40-68)"""

    def __init__(self, in_channels):
        super(DAInsHead, self).__init__()
        self.fc1_da = nn.Linear(in_channels, 1024)
        self.fc2_da = nn.Linear(1024, 1024)
        self.fc3_da = nn.Linear(1024, 1)
        for l in (self.fc1_da, self.fc2_da):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        nn.init.normal_(self.fc3_da.weight, std=0.05)
        nn.init.constant_(self.fc3_da.bias, 0)
        self.in_channels = in_channels

    def forward(self, x):
        x = linear(x, self.fc1_da.weight, self.fc1_da.bias, relu=True)
        if self.training:
            x = x * rng.dropout_mask(tuple(x.shape), 0.5, x.device)
        x = linear(x, self.fc2_da.weight, self.fc2_da.bias, relu=True)
        if self.training:
            x = x * rng.dropout_mask(tuple(x.shape), 0.5, x.device)
        return linear(x, self.fc3_da.weight, self.fc3_da.bias)


_FUSED_INS = True   # 0: one DAInsHead.forward per pass + ATen losses
_GRL_CACHE = {}


def _grl_vector(weights, device):
    """float [P] device tensor of the passes' gradient-reversal weights.  Plain floats come from a cache (a
    torch.tensor(list, device=...) per step is a blocking host->device copy); 0-d device tensors (AdvGRL) are stacked"""
    if all(not isinstance(w, torch.Tensor) for w in weights):
        key = (tuple(float(w) for w in weights), str(device))
        v = _GRL_CACHE.get(key)
        if v is None:
            v = _GRL_CACHE[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
        return v
    return torch.stack([w.reshape(()).to(torch.float32) if isinstance(w, torch.Tensor)
                        else _grl_vector([w], device)[0] for w in weights])


def _draw_pass_masks(passes, rows, device):
    """dropout masks of `passes` head passes in the reference's order of draws (pass by pass: after fc1, after fc2;
    da_heads.py:63,65) -> (masks1 [P, R, 1024], masks2 [P, R, 1024])"""
    m1, m2 = [], []
    for _ in range(passes):
        m1.append(rng.dropout_mask((rows, 1024), INS_DROPOUT_P, device))
        m2.append(rng.dropout_mask((rows, 1024), INS_DROPOUT_P, device))
    return torch.stack(m1), torch.stack(m2)


def _n_source_rows(da_ins_labels):
    n_src = getattr(da_ins_labels, "_n_src_host", None)   # set by the box head, which knows it without a round trip
    return int(torch.nonzero(da_ins_labels).size(0)) if n_src is None else int(n_src)


def _vector_ins_features(cfg):
    """True when the box head already yields one vector per ROI.  The reference tests CONV_BODY.startswith('V')
    (da_heads.py:367-369); its FPN combination was never wired (AvgPool2d on the FPN2MLP vector and a 2048-wide
    first layer) — here an MLP box head (FPN2MLPFeatureExtractor) counts as "vector" too."""
    return (cfg.MODEL.BACKBONE.CONV_BODY.startswith("V")
            or "MLP" in cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR)


def _ins_input_dim(cfg):
    if _vector_ins_features(cfg):
        return cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM
    return cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * 8


def _pool_ins(feat, resnet_backbone):
    """AvgPool2d(7) + flatten of the [R,2048,7,7] ROI features (da_heads.py:402-407)"""
    if resnet_backbone and feat.dim() == 4:
        return global_avg_pool(feat)
    return feat.reshape(feat.size(0), -1)


class DomainAdaptationModule(torch.nn.Module):
    """image-level + instance-level adversarial losses and their consistency regulariser
    (da_heads.py:354-440, losses da_heads/loss.py:55-104)"""

    def __init__(self, cfg):
        super(DomainAdaptationModule, self).__init__()
        self.cfg = cfg.clone()
        da = cfg.MODEL.DA_HEADS
        self.resnet_backbone = cfg.MODEL.BACKBONE.CONV_BODY.startswith("R")
        self.avgpool = nn.AvgPool2d(kernel_size=7, stride=7)
        self.img_weight, self.ins_weight, self.cst_weight = (da.DA_IMG_LOSS_WEIGHT, da.DA_INS_LOSS_WEIGHT,
                                                             da.DA_CST_LOSS_WEIGHT)
        self.grl_img = GradientScalarLayer(-1.0 * da.DA_IMG_GRL_WEIGHT)
        self.grl_ins = GradientScalarLayer(-1.0 * da.DA_INS_GRL_WEIGHT)
        self.grl_img_consist = GradientScalarLayer(1.0 * da.DA_IMG_GRL_WEIGHT)
        self.grl_ins_consist = GradientScalarLayer(1.0 * da.DA_INS_GRL_WEIGHT)
        self.imghead = DAImgHead(cfg.MODEL.BACKBONE.OUT_CHANNELS)
        self.inshead = DAInsHead(_ins_input_dim(cfg))

    # opt-in (engine.trainer.enable_overlapped_rpn_backward), same contract as RPNModule.early_backward
    early_backward = False

    @property
    def needs_instance_features(self):
        """False: no loss reads the instance-level features (the box head may then leave the target-domain ROIs out)"""
        return self.ins_weight > 0 or self.cst_weight > 0

    def _image_level(self, img_features, targets):
        """image head: one fused evaluation serves both the BCE (GRL -w) and the consistency (GRL +w) paths.
        Several levels (FPN): the reference concatenates the per-level logits along dim 0 (loss.py:81-92), which
        only works for equal map sizes, i.e. never for a pyramid.  The extension used here is what a dim-1
        concatenation would give: one BCE mean over the elements of all levels, and the consistency term averaged
        over levels (consistency_loss.py:13-27 already concatenates its per-level columns along dim 1)."""
        labels = image_domain_labels(targets)
        per_level = [self.imghead.fused(f, labels, self.grl_img.weight, self.grl_img_consist.weight)
                     for f in img_features]
        if len(per_level) == 1:
            da_img_loss = per_level[0][0]
        else:
            sizes = [float(f.shape[2] * f.shape[3]) for f in img_features]
            da_img_loss = sum(l[0] * s for l, s in zip(per_level, sizes)) / sum(sizes)
        return da_img_loss, [l[1] for l in per_level]

    def early_image_level(self, img_features, targets):
        """The image-level loss needs the backbone features only.  In the reference's order it is evaluated after
        the box head, in the stretch between the box head's forward and backward where nothing large can run
        (tools/gap_analysis.py: 1.5 ms of small kernels).  With `early_backward` it — and its backward — is queued
        right behind the RPN branch's backward instead, in front of the box head, where the compute stream otherwise
        waits for the proposals.  Returns the gradients w.r.t. the feature maps (to be injected by
        RPNModule.bridge_features) or None when not applicable: the consistency term back-propagates through the
        image head together with the instance head, so it keeps the reference's order."""
        self._early = None
        if not (_EARLY_DA and self.training and self.early_backward and torch.is_grad_enabled() and self.cst_weight == 0
                and self.img_weight > 0 and all(f.is_cuda and f.requires_grad for f in img_features)):
            return None
        from ... import amax
        head_in = [amax.carry(f.detach().requires_grad_(True), f) for f in img_features]
        da_img_loss, _ = self._image_level(head_in, targets)
        loss = self.img_weight * da_img_loss
        torch.autograd.backward([loss])
        self._early = loss.detach()
        # the loss value is read on the compute stream later (forward): it waits for this point of the current stream
        self._early_ready = torch.cuda.current_stream(loss.device).record_event()
        return [f.grad for f in head_in]

    def forward(self, img_features, da_ins_feature, da_ins_labels, targets=None):
        if not self.training:
            return {}
        early, self._early = getattr(self, "_early", None), None
        if early is not None:
            torch.cuda.current_stream(early.device).wait_event(self._early_ready)
            early.record_stream(torch.cuda.current_stream(early.device))
        if da_ins_feature is None:
            # the box head left the instance-level features out because no loss reads them (ROIBoxHead.forward)
            assert not self.needs_instance_features
            if early is not None:
                return {"loss_da_image": early}
            da_img_loss, _ = self._image_level(img_features, targets)
            return {"loss_da_image": self.img_weight * da_img_loss} if self.img_weight > 0 else {}
        da_ins_feature = _pool_ins(da_ins_feature, self.resnet_backbone)
        if _FUSED_INS and da_ins_feature.is_cuda and da_ins_feature.shape[0] > 0 \
                and self.inshead.fc1_da.out_features == 1024 and self.inshead.fc2_da.out_features == 1024:
            return self._forward_fused_instance(img_features, da_ins_feature, da_ins_labels, targets, early)
        # instance head: adversarial pass then consistency pass, each with its own dropout masks (same program order
        # of the random draws as the reference).  They are independent of the image head: on the GPU they run on a
        # side stream beside it, and the compute stream only waits for them if a loss uses them.
        need_ins = self.ins_weight > 0 or self.cst_weight > 0
        side = main = None
        if da_ins_feature.is_cuda:
            main = torch.cuda.current_stream(da_ins_feature.device)
            side = other_stream(da_ins_feature.device)
            side.wait_stream(main)
            da_ins_feature.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            da_ins_features = self.inshead(self.grl_ins(da_ins_feature))
            da_ins_consist = self.inshead(self.grl_ins_consist(da_ins_feature)).sigmoid()
        if side is not None and need_ins:
            main.wait_stream(side)
            da_ins_features.record_stream(main)
            da_ins_consist.record_stream(main)
        losses = {}
        if early is not None:
            losses["loss_da_image"] = early
            img_mean_sig = None        # only the consistency term reads it, and that excludes the early path
        else:
            da_img_loss, img_mean_sig = self._image_level(img_features, targets)
        if self.img_weight > 0 and early is None:
            losses["loss_da_image"] = self.img_weight * da_img_loss
        if self.ins_weight > 0:
            losses["loss_da_instance"] = self.ins_weight * da_ins_loss(da_ins_features, da_ins_labels)
        if self.cst_weight > 0:
            losses["loss_da_consistency"] = self.cst_weight * da_consist_loss(img_mean_sig, da_ins_consist,
                                                                              da_ins_labels)
        return losses

    def _forward_fused_instance(self, img_features, feat, da_ins_labels, targets, early):
        """image head first (the consistency rows read its per-image means), then BOTH instance-head passes and both
        instance-level losses as one autograd node (fused.da_instance_head).  The dropout masks of both passes are drawn
        in the reference's order even when a pass's loss weight is 0 (its draws still advance the random stream)."""
        losses = {}
        img_mean_sig = None
        if early is not None:
            losses["loss_da_image"] = early
        else:
            da_img_loss, img_mean_sig = self._image_level(img_features, targets)
            if self.img_weight > 0:
                losses["loss_da_image"] = self.img_weight * da_img_loss
        R = feat.shape[0]
        masks1, masks2 = _draw_pass_masks(2, R, feat.device)       # adversarial pass, then consistency pass
        kinds, grl, sel = [], [], []
        if self.ins_weight > 0:
            kinds.append("bce"), grl.append(self.grl_ins.weight), sel.append(0)
        if self.cst_weight > 0:
            kinds.append("cst"), grl.append(self.grl_ins_consist.weight), sel.append(1)
        if not kinds:
            return losses
        if len(sel) == 1:
            masks1, masks2 = masks1[sel[0]:sel[0] + 1], masks2[sel[0]:sel[0] + 1]
        means = torch.stack(list(img_mean_sig)) if self.cst_weight > 0 else None        # [levels, 2]
        if means is not None:
            assert means.shape[1] == 2, \
                "only batch size=2 is supported for consistency loss now, received batch size: {}".format(means.shape[1])
        bce, cst, _ = da_instance_head(feat, self.inshead, da_ins_labels, means, masks1, masks2,
                                       _grl_vector(grl, feat.device), kinds, _n_source_rows(da_ins_labels))
        if self.ins_weight > 0:
            losses["loss_da_instance"] = self.ins_weight * bce
        if self.cst_weight > 0:
            losses["loss_da_consistency"] = self.cst_weight * cst
        return losses


class DomainAdaptationModule_triplet(torch.nn.Module):
    """component-wise DA losses with AdvGRL and domain-level triplet regularisation (da_heads.py:72-344)"""

    def __init__(self, cfg):
        super(DomainAdaptationModule_triplet, self).__init__()
        self.cfg = cfg.clone()
        da = cfg.MODEL.DA_HEADS
        self.resnet_backbone = cfg.MODEL.BACKBONE.CONV_BODY.startswith("R")
        self.avgpool = nn.AvgPool2d(kernel_size=7, stride=7)
        self.img_weight, self.ins_weight, self.cst_weight = (da.DA_IMG_LOSS_WEIGHT, da.DA_INS_LOSS_WEIGHT,
                                                             da.DA_CST_LOSS_WEIGHT)
        self.triplet_img_weight, self.triplet_ins_weight = da.DA_TRIPLET_IMG_WEIGHT, da.DA_TRIPLET_INS_WEIGHT
        self.grl_img = GradientScalarLayer(-1.0 * da.DA_IMG_GRL_WEIGHT)
        self.grl_ins = GradientScalarLayer(-1.0 * da.DA_INS_GRL_WEIGHT)
        self.grl_img_consist = GradientScalarLayer(1.0 * da.DA_IMG_GRL_WEIGHT)
        self.grl_ins_consist = GradientScalarLayer(1.0 * da.DA_INS_GRL_WEIGHT)
        self.imghead = DAImgHead(cfg.MODEL.BACKBONE.OUT_CHANNELS)
        self.inshead = DAInsHead(_ins_input_dim(cfg))
        self.loss_evaluator = TripletMargins()
        self.triplet_ins = [1]
        self.triplet_img = [1]
        self.advGRL = da.DA_ADV_GRL
        self.advGRL_threshold = da.DA_ADV_GRL_THRESHOLD
        self.adv_img_weight, self.adv_ins_weight = da.DA_IMG_advGRL_WEIGHT, da.DA_INS_advGRL_WEIGHT
        self.triplet_metric_img, self.triplet_metric_ins = da.TRIPLET_MARGIN_IMG, da.TRIPLET_MARGIN_INS
        self.triplet_max_margin = da.TRIPLET_MAX_MARGIN
        # the AdvGRL gate: BCE-with-logits of logits (0.7, 0.3) against labels (1, 0) (da_heads.py:175)
        self.adv_gate = float(F.binary_cross_entropy_with_logits(torch.tensor([[0.7, 0.3]]),
                                                                 torch.tensor([[1.0, 0.0]])))

    def adv_grl_weight(self, current_loss, base_weight, adv_weight):
        """adaptive reversal weight (da_heads.py:173-195) as a 0-d device tensor, no host sync:
        loss <= gate -> -adv_weight * min(threshold, 1 / loss), else the fixed -base_weight.
        (The reference calls .numpy() on the loss here, which fails on device tensors — SURVEY.md fact 10;
        this is the intended rule with the loss detached.)"""
        loss = current_loss.detach()
        adaptive = -adv_weight * torch.clamp(1.0 / loss, max=float(self.advGRL_threshold))
        return torch.where(loss <= self.adv_gate, adaptive, torch.full_like(loss, -base_weight))

    @property
    def needs_instance_features(self):
        return self.ins_weight > 0 or self.cst_weight > 0

    def forward(self, img_features, da_ins_feature, da_ins_labels, da_ins_feas_set, img_fea_set, targets=None):
        if not self.training:
            return {}
        assert len(img_features) == 1
        losses = {}
        if self.triplet_ins_weight > 0:
            s, p, n = [_pool_ins(f, self.resnet_backbone) for f in da_ins_feas_set]
            loss = self.loss_evaluator.triplet_ins_loss(
                s, p, n, self.triplet_ins[-1], adaptive=False, lr=0.001, max_margin=self.triplet_max_margin,
                margin=self.triplet_metric_ins)
            losses["triplet_loss_instance"] = self.triplet_ins_weight * loss
            self.triplet_ins = [_later_value(loss)]  # only [-1] is ever read
        if self.triplet_img_weight > 0:
            loss = self.loss_evaluator.triplet_img_loss(
                img_fea_set[0][0], img_fea_set[1][0], img_fea_set[2][0], self.triplet_img[-1], adaptive=True,
                lr=0.001, max_margin=self.triplet_max_margin, margin=self.triplet_metric_img)
            losses["triplet_loss_image"] = self.triplet_img_weight * loss
            self.triplet_img = [_later_value(loss)]

        need_img = self.img_weight > 0 or self.cst_weight > 0
        if need_img:
            labels = image_domain_labels(targets)
            w_adv = self.grl_img.weight
            if self.advGRL and self.img_weight > 0:
                # the head's own forward value IS the reference's detached "current loss" (da_heads.py:128-130):
                # the adaptive weight is resolved from it inside the fused op, no second pass
                w_adv = lambda cur: self.adv_grl_weight(cur, -self.grl_img.weight, self.adv_img_weight)  # noqa: E731
            da_img_loss, img_mean_sig, _ = self.imghead.fused(img_features[0], labels, w_adv,
                                                             self.grl_img_consist.weight)
            if self.img_weight > 0:
                losses["loss_da_image"] = self.img_weight * da_img_loss
        fused_ins = (_FUSED_INS and (self.ins_weight > 0 or self.cst_weight > 0) and da_ins_feature is not None
                     and da_ins_feature.is_cuda and da_ins_feature.shape[0] > 0)
        if fused_ins:
            losses.update(self._fused_instance_losses(_pool_ins(da_ins_feature, self.resnet_backbone), da_ins_labels,
                                                      img_mean_sig if self.cst_weight > 0 else None))
            return losses
        if self.ins_weight > 0:
            feat = _pool_ins(da_ins_feature, self.resnet_backbone)
            cur = da_ins_loss(self.inshead(feat.detach()), da_ins_labels)  # own dropout draws, as the reference
            if self.advGRL:
                w = self.adv_grl_weight(cur, -self.grl_ins.weight, self.adv_ins_weight)
                grl_fea = gradient_scalar(feat, w)
            else:
                grl_fea = self.grl_ins(feat)
            losses["loss_da_instance"] = self.ins_weight * da_ins_loss(self.inshead(grl_fea), da_ins_labels)
        if self.cst_weight > 0:
            feat = _pool_ins(da_ins_feature, self.resnet_backbone)
            ins_consist = self.inshead(self.grl_ins_consist(feat)).sigmoid()
            losses["loss_da_consistency"] = self.cst_weight * da_consist_loss(img_mean_sig, ins_consist,
                                                                              da_ins_labels)
        return losses


class _LaterValue(object):
    """the value of a 0-d device tensor, read on the host one iteration LATER.  The reference keeps `loss.detach().cpu()`
    of the triplet losses (da_heads.py:320,325) for the next iteration's adaptive margin (loss.py:128-222: `prev_loss ==
    0.0`), i.e. a device->host round trip in the middle of every forward pass; here the copy into pinned memory is queued
    behind the loss and the comparison — made when the NEXT iteration asks — waits for an event that fired long ago."""

    def __init__(self, t):
        self.host = torch.empty((), dtype=t.dtype, pin_memory=True)
        self.host.copy_(t.detach(), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(t.device))

    def value(self):
        self.event.synchronize()
        return float(self.host)

    def __eq__(self, other):
        return self.value() == other

    def __ne__(self, other):
        return self.value() != other

    def __float__(self):
        return self.value()


def _later_value(loss):
    loss = loss.detach()
    return _LaterValue(loss) if loss.is_cuda else loss.cpu()


def _triplet_fused_instance_losses(self, feat, da_ins_labels, img_mean_sig):
    """instance-level BCE (da_heads.py:147-169, with the AdvGRL weight of :173-195) and consistency (:276-291) of the
    component-wise module through fused.da_instance_head.  Order of the dropout draws as in the reference: the detached
    "current loss" pass, the adversarial pass, the consistency pass."""
    R, dev = feat.shape[0], feat.device
    n_src = _n_source_rows(da_ins_labels)
    kinds, grl = [], []
    if self.ins_weight > 0:
        m1, m2 = _draw_pass_masks(1, R, dev)
        w = self.grl_ins.weight
        if self.advGRL:
            with torch.no_grad():     # the value the reference detaches (da_heads.py:150-152): forward only
                cur, _, _ = da_instance_head(feat.detach(), self.inshead, da_ins_labels, None, m1, m2,
                                             _grl_vector([0.0], dev), ("bce",), n_src)
            w = self.adv_grl_weight(cur, -self.grl_ins.weight, self.adv_ins_weight)
        kinds.append("bce"), grl.append(w)
    if self.cst_weight > 0:
        kinds.append("cst"), grl.append(self.grl_ins_consist.weight)
    masks1, masks2 = _draw_pass_masks(len(kinds), R, dev)
    means = None
    if self.cst_weight > 0:
        means = torch.stack(list(img_mean_sig) if isinstance(img_mean_sig, (list, tuple)) else [img_mean_sig])
        assert means.shape[1] == 2, \
            "only batch size=2 is supported for consistency loss now, received batch size: {}".format(means.shape[1])
    bce, cst, _ = da_instance_head(feat, self.inshead, da_ins_labels, means, masks1, masks2, _grl_vector(grl, dev), kinds,
                                   n_src)
    out = {}
    if self.ins_weight > 0:
        out["loss_da_instance"] = self.ins_weight * bce
    if self.cst_weight > 0:
        out["loss_da_consistency"] = self.cst_weight * cst
    return out


DomainAdaptationModule_triplet._fused_instance_losses = _triplet_fused_instance_losses


def build_da_heads(cfg):
    return DomainAdaptationModule(cfg) if cfg.MODEL.DOMAIN_ADAPTATION_ON else []


def build_da_heads_triplet(cfg):
    return DomainAdaptationModule_triplet(cfg) if cfg.MODEL.DOMAIN_ADAPTATION_ON else []
