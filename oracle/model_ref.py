"""CPU restatement of the DA Faster R-CNN training step (torch fp32 on the CPU + oracle/dadet_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/dadet_oracle.c): imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by da_detect_amd/.  It is a flat, functional re-statement — plain tensors and dicts, no
BoxList / nn.Module — of what the reference executes for one training iteration, written against the reference
source, each function citing the lines it follows.  Floating-point contractions are torch's own CPU kernels (the
"plain PyTorch fp32 reference" for the GEMM-shaped kernels); NMS and ROIAlign go through the C oracle.

Parity status: PINNED against the imported Python reference (tests/golden/make_golden.py runs the reference from
/root/reference on the CPU with identical weights / inputs / RNG seed and stores its loss dictionaries and
intermediate tensors under tests/golden/; tests/test_oracle_golden.py replays them through this file).

Random draws follow the reference's call order on the global CPU generator (sampler randperm:
modeling/balanced_positive_negative_sampler.py:57-58; F.dropout in da_heads.py:63,65), so one torch.manual_seed
reproduces the reference's sampled indices and dropout masks.

`DeviceDraws` (below) is the one piece that restates the PRODUCT instead of the reference: the HIP path's default
samplers (da_detect_amd/csrc/sampling.hip) keep the reference's sampling RULE (counts per class, ignored rows, ascending
order) but draw the random subset from splitmix64 keys instead of torch.randperm.  Given the seeds the product consumed,
the oracle takes the same subsets, so the default (benchmarked) GPU path can be compared with this file on the same
sample (tests/test_default_path_gpu.py).
"""
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O

BELOW_LOW, BETWEEN = -1, -2


# ------------------------------------------------------------------------------------------------- layers
def frozen_bn(x, sd, prefix):
    """x * scale + bias with scale = weight * rsqrt(running_var) (layers/batch_norm.py:19-24)"""
    scale = sd[prefix + ".weight"] * sd[prefix + ".running_var"].rsqrt()
    bias = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def bottleneck(x, sd, p, stride):
    """modeling/backbone/resnet.py:294-314 with STRIDE_IN_1X1 (the stride sits in conv1 and the shortcut)"""
    out = F.relu(frozen_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride), sd, p + ".bn1"))
    out = F.relu(frozen_bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2"))
    out = frozen_bn(F.conv2d(out, sd[p + ".conv3.weight"], None), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        x = frozen_bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    return F.relu(out + x)


def stage(x, sd, prefix, blocks, first_stride):
    for b in range(blocks):
        x = bottleneck(x, sd, "%s.%d" % (prefix, b), first_stride if b == 0 else 1)
    return x


def backbone_c4(images, sd, blocks=(3, 4, 6)):
    """stem + layer1..3 (resnet.py:138-145, 331-336)"""
    p = "backbone.body"
    x = F.relu(frozen_bn(F.conv2d(images, sd[p + ".stem.conv1.weight"], None, 2, 3), sd, p + ".stem.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    x = stage(x, sd, p + ".layer1", blocks[0], 1)
    x = stage(x, sd, p + ".layer2", blocks[1], 2)
    return stage(x, sd, p + ".layer3", blocks[2], 2)


# ------------------------------------------------------------------------------------------------ anchors
def cell_anchors(stride, sizes, ratios):
    """modeling/rpn/anchor_generator.py:222-291"""
    def whc(a):
        w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, cx, cy):
        ws, hs = ws[:, None], hs[:, None]
        return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1), cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))

    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    w, h, cx, cy = whc(base)
    ws = np.round(np.sqrt(w * h / np.array(ratios, dtype=np.float64)))
    hs = np.round(ws * np.array(ratios, dtype=np.float64))
    out = []
    for a in mk(ws, hs, cx, cy):
        w, h, cx, cy = whc(a)
        sc = np.array(sizes, dtype=np.float64) / stride
        out.append(mk(w * sc, h * sc, cx, cy))
    return torch.from_numpy(np.vstack(out)).float()


def grid_anchors(fh, fw, stride, cell):
    """anchor index = (y*fw + x)*A + a (anchor_generator.py:73-97)"""
    sx = torch.arange(0, fw * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, fh * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4)


# ----------------------------------------------------------------------------------------- box arithmetic
def box_area(b):
    return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)


def box_iou(a, b):
    """structures/boxlist_ops.py:56-91"""
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (box_area(a)[:, None] + box_area(b) - inter)


def encode(ref, prop, weights):
    """modeling/box_coder.py:22-50"""
    ew, eh = prop[:, 2] - prop[:, 0] + 1, prop[:, 3] - prop[:, 1] + 1
    ecx, ecy = prop[:, 0] + 0.5 * ew, prop[:, 1] + 0.5 * eh
    gw, gh = ref[:, 2] - ref[:, 0] + 1, ref[:, 3] - ref[:, 1] + 1
    gcx, gcy = ref[:, 0] + 0.5 * gw, ref[:, 1] + 0.5 * gh
    wx, wy, ww, wh = weights
    return torch.stack((wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * torch.log(gw / ew),
                        wh * torch.log(gh / eh)), dim=1)


def decode(codes, boxes, weights, clip=math.log(1000.0 / 16)):
    """modeling/box_coder.py:52-95 for [N,4] codes"""
    w, h = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = codes[:, 0] / wx, codes[:, 1] / wy
    dw = torch.clamp(codes[:, 2] / ww, max=clip)
    dh = torch.clamp(codes[:, 3] / wh, max=clip)
    pcx, pcy = dx * w + cx, dy * h + cy
    pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw - 1, pcy + 0.5 * ph - 1), dim=1)


def decode_multi(codes, boxes, weights, clip=math.log(1000.0 / 16)):
    """modeling/box_coder.py:52-95 for [N, 4*K] codes (K classes per box) -> [N, 4*K]"""
    w, h = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = codes[:, 0::4] / wx, codes[:, 1::4] / wy
    dw = torch.clamp(codes[:, 2::4] / ww, max=clip)
    dh = torch.clamp(codes[:, 3::4] / wh, max=clip)
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
    out = torch.zeros_like(codes)
    out[:, 0::4], out[:, 1::4] = pcx - 0.5 * pw, pcy - 0.5 * ph
    out[:, 2::4], out[:, 3::4] = pcx + 0.5 * pw - 1, pcy + 0.5 * ph - 1
    return out


def matcher(iou, high, low, allow_low_quality):
    """modeling/matcher.py:42-112"""
    vals, matches = iou.max(dim=0)
    all_matches = matches.clone()
    matches[vals < low] = BELOW_LOW
    matches[(vals >= low) & (vals < high)] = BETWEEN
    if allow_low_quality:
        best, _ = iou.max(dim=1)
        upd = torch.nonzero(iou == best[:, None])[:, 1]
        matches[upd] = all_matches[upd]
    return matches


def sample_pos_neg(labels, batch_size, positive_fraction):
    """modeling/balanced_positive_negative_sampler.py:27-76 for one image -> (pos mask, neg mask)"""
    positive = torch.nonzero(labels >= 1).squeeze(1)
    negative = torch.nonzero(labels == 0).squeeze(1)
    num_pos = min(positive.numel(), int(batch_size * positive_fraction))
    num_neg = min(negative.numel(), batch_size - num_pos)
    perm1 = torch.randperm(positive.numel())[:num_pos]
    perm2 = torch.randperm(negative.numel())[:num_neg]
    pm = torch.zeros_like(labels, dtype=torch.bool)
    nm = torch.zeros_like(labels, dtype=torch.bool)
    pm[positive[perm1]] = 1
    nm[negative[perm2]] = 1
    return pm, nm


class DeviceDraws(object):
    """random draws of the product's default GPU path, replayed on the CPU: `seeds` are the 64-bit values the
    product passed to dadet_sample_anchors / dadet_sample_rois (da_detect_amd/utils/rng.py next_seed), in call order;
    `masks` the multiplicative dropout masks it drew for DAInsHead (da_heads.py:63,65), in call order."""

    def __init__(self, seeds, masks=()):
        self.seeds, self.masks = list(seeds), [m.detach().cpu() for m in masks]
        self.taken_seeds = self.taken_masks = 0

    def seed(self):
        self.taken_seeds += 1
        return int(self.seeds[self.taken_seeds - 1])

    def mask(self, shape):
        if not self.masks:
            # the product drew none: it evaluates the instance-level head only when a loss reads it, the reference (and
            # this restatement) always (da_heads.py:402-439) — whatever mask is used here reaches no loss
            return torch.full(tuple(shape), 2.0) * (torch.rand(tuple(shape), generator=torch.Generator().manual_seed(0)) < 0.5)
        self.taken_masks += 1
        m = self.masks[self.taken_masks - 1]
        assert tuple(m.shape) == tuple(shape), (tuple(m.shape), tuple(shape))
        return m

    def exhausted(self):
        return self.taken_seeds == len(self.seeds) and self.taken_masks == len(self.masks)


def device_sample_keys(seed, n):
    """32-bit random key of row i = top half of the splitmix64 finaliser of seed + (i + 1) * 0x9E3779B97F4A7C15
    (da_detect_amd/csrc/sampling.hip `sample_key`)"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)).astype(np.uint32)


def sample_pos_neg_device(labels, batch_size, positive_fraction, seed):
    """same counts as sample_pos_neg (balanced_positive_negative_sampler.py:27-76); the uniformly random subset of
    each class is "the k rows with the smallest keys, ties to the lower index" (sample_rois_kernel: bitonic sort of
    (class, key, index); sample_anchors_kernel: radix select of the k-th key + lowest-index ties) -> (pos mask, neg mask)"""
    lab = labels.numpy()
    keys = device_sample_keys(seed, lab.shape[0])
    positive = np.nonzero(lab >= 1)[0]
    negative = np.nonzero(lab == 0)[0]
    num_pos = min(positive.size, int(batch_size * positive_fraction))
    num_neg = min(negative.size, batch_size - num_pos)
    pm = torch.zeros(lab.shape[0], dtype=torch.bool)
    nm = torch.zeros(lab.shape[0], dtype=torch.bool)
    for rows, k, mask in ((positive, num_pos, pm), (negative, num_neg, nm)):
        order = np.lexsort((rows, keys[rows]))      # by key, then by index
        mask[torch.from_numpy(rows[order[:k]])] = True
    return pm, nm


def smooth_l1(x, t, beta):
    """layers/smooth_l1_loss.py:6-16, summed"""
    n = torch.abs(x - t)
    return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta).sum()


# ---------------------------------------------------------------------------------------------- ROIAlign
class _RoiAlign(torch.autograd.Function):
    """forward: csrc/cpu/ROIAlign_cpu.cpp; backward: csrc/cuda/ROIAlign_cuda.cu:178-254 (C oracle)"""

    @staticmethod
    def forward(ctx, x, rois, scale, ph, pw, sr):
        ctx.save_for_backward(rois)
        ctx.args = (scale, ph, pw, sr, tuple(x.shape))
        # the C oracle is fp32 like the reference operator; a float64 run of this file (noise-floor measurements of
        # the gradient tests) keeps its dtype around it
        return torch.from_numpy(O.roi_align_forward(x.detach().numpy(), rois.numpy(), scale, ph, pw, sr)).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        scale, ph, pw, sr, (B, C, H, W) = ctx.args
        gin = O.roi_align_backward(g.contiguous().numpy(), rois.numpy(), scale, ph, pw, B, C, H, W, sr)
        return torch.from_numpy(gin).to(g.dtype), None, None, None, None, None


def roi_align(x, rois, scale, ph, pw, sr):
    return _RoiAlign.apply(x, rois, scale, ph, pw, sr)


# --------------------------------------------------------------------------------------------------- RPN
def rpn_head(feat, sd):
    """modeling/rpn/rpn.py:39-46"""
    t = F.relu(F.conv2d(feat, sd["rpn.head.conv.weight"], sd["rpn.head.conv.bias"], 1, 1))
    return (F.conv2d(t, sd["rpn.head.cls_logits.weight"], sd["rpn.head.cls_logits.bias"]),
            F.conv2d(t, sd["rpn.head.bbox_pred.weight"], sd["rpn.head.bbox_pred.bias"]))


def flatten_hwa(layer, C):
    """rpn/utils.py:10-14: [N, A*C, H, W] -> [N, H*W*A, C]"""
    N, _, H, W = layer.shape
    return layer.view(N, -1, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)


def rpn_proposals(objectness, deltas, anchors, image_sizes, gts, cfg, training):
    """rpn/inference.py:76-152: per image (boxes [P,4], objectness [P]).  Equal scores are ranked by ascending
    anchor index (stable sort); the reference's topk leaves that order unspecified."""
    rpn = cfg.MODEL.RPN
    pre = rpn.PRE_NMS_TOP_N_TRAIN if training else rpn.PRE_NMS_TOP_N_TEST
    post = rpn.POST_NMS_TOP_N_TRAIN if training else rpn.POST_NMS_TOP_N_TEST
    N = objectness.shape[0]
    scores = flatten_hwa(objectness, 1).reshape(N, -1).sigmoid()
    d = flatten_hwa(deltas, 4)
    pre = min(pre, scores.shape[1])
    out = []
    for i in range(N):
        s, order = torch.sort(scores[i], descending=True, stable=True)
        s, order = s[:pre], order[:pre]
        h, w = image_sizes[i]
        boxes = torch.from_numpy(O.decode_clip(d[i][order].numpy(), anchors[order].numpy(), (1.0, 1.0, 1.0, 1.0),
                                               math.log(1000.0 / 16), w, h))
        if rpn.MIN_SIZE > 0:
            keep = ((boxes[:, 2] - boxes[:, 0] + 1 >= rpn.MIN_SIZE) & (boxes[:, 3] - boxes[:, 1] + 1 >= rpn.MIN_SIZE))
            boxes, s = boxes[keep], s[keep]
        keep = torch.from_numpy(O.nms(boxes.numpy(), s.numpy(), rpn.NMS_THRESH, 0))[:post]
        boxes, s = boxes[keep], s[keep]
        if training and gts[i]["is_source"].any():  # add_gt_proposals, source images only
            boxes = torch.cat([boxes, gts[i]["boxes"]], 0)
            s = torch.cat([s, torch.ones(len(gts[i]["boxes"]))], 0)
        out.append((boxes, s))
    return out


def rpn_losses(objectness, deltas, anchors, image_sizes, gts, cfg, draws=None, record=None):
    """rpn/loss.py:57-143"""
    rpn = cfg.MODEL.RPN
    labels, reg_targets = [], []
    for i, gt in enumerate(gts):
        if not gt["is_source"].any():
            continue
        h, w = image_sizes[i]
        m = matcher(box_iou(gt["boxes"], anchors), rpn.FG_IOU_THRESHOLD, rpn.BG_IOU_THRESHOLD, True)
        matched = gt["boxes"][m.clamp(min=0)]
        lab = (m >= 0).to(torch.float32)
        lab[m == BELOW_LOW] = 0
        t = rpn.STRADDLE_THRESH
        visible = (anchors[:, 0] >= -t) & (anchors[:, 1] >= -t) & (anchors[:, 2] < w + t) & (anchors[:, 3] < h + t)
        lab[~visible] = -1
        lab[m == BETWEEN] = -1
        labels.append(lab)
        reg_targets.append(encode(matched, anchors, (1.0, 1.0, 1.0, 1.0)))
    pos, neg = [], []
    for lab in labels:
        if draws is not None:
            pm, nm = sample_pos_neg_device(lab, rpn.BATCH_SIZE_PER_IMAGE, rpn.POSITIVE_FRACTION, draws.seed())
        else:
            pm, nm = sample_pos_neg(lab, rpn.BATCH_SIZE_PER_IMAGE, rpn.POSITIVE_FRACTION)
        pos.append(pm)
        neg.append(nm)
    pos_inds = torch.nonzero(torch.cat(pos)).squeeze(1)
    neg_inds = torch.nonzero(torch.cat(neg)).squeeze(1)
    sampled = torch.cat([pos_inds, neg_inds])
    if record is not None:
        record.update(rpn_pos_inds=pos_inds, rpn_neg_inds=neg_inds)
    if isinstance(objectness, (list, tuple)):
        # several feature levels: per image the levels are concatenated in order (rpn/utils.py:17-45
        # concat_box_prediction_layers), `anchors` is the same concatenation of the per-level anchor grids
        obj = torch.cat([flatten_hwa(o, 1) for o in objectness], dim=1).reshape(-1)
        reg = torch.cat([flatten_hwa(d, 4) for d in deltas], dim=1).reshape(-1, 4)
    else:
        obj = flatten_hwa(objectness, 1).reshape(-1)
        reg = flatten_hwa(deltas, 4).reshape(-1, 4)
    labels, reg_targets = torch.cat(labels), torch.cat(reg_targets)
    box_loss = smooth_l1(reg[pos_inds], reg_targets[pos_inds], 1.0 / 9) / sampled.numel()
    obj_loss = F.binary_cross_entropy_with_logits(obj[sampled], labels[sampled])
    return obj_loss, box_loss


# ---------------------------------------------------------------------------------------------- box head
def box_head_targets(proposals, gts, cfg, sample_for_da):
    """roi_heads/box_head/loss.py:68-93 -> per image (labels, regression targets, domain flag)"""
    rh = cfg.MODEL.ROI_HEADS
    out = []
    for (boxes, _), gt in zip(proposals, gts):
        is_source = bool(gt["is_source"].any())
        m = matcher(box_iou(gt["boxes"], boxes), rh.FG_IOU_THRESHOLD, rh.BG_IOU_THRESHOLD, False)
        idx = m.clamp(min=0) if is_source else m
        lab = gt["labels"][idx].to(torch.int64).clone()
        lab[m == BELOW_LOW] = 0
        lab[m == BETWEEN] = -1
        reg = encode(gt["boxes"][idx], boxes, rh.BBOX_REG_WEIGHTS)
        if (not is_source) or sample_for_da:
            lab[:] = 0
        out.append((lab, reg, is_source))
    return out


def subsample(proposals, gts, cfg, sample_for_da=False, draws=None):
    """loss.py:95-163 -> per image dict(boxes, labels, reg, domain)"""
    rh = cfg.MODEL.ROI_HEADS
    tg = box_head_targets(proposals, gts, cfg, sample_for_da)
    masks = []
    for (boxes, _), (lab, _, _) in zip(proposals, tg):
        if draws is None:
            masks.append(sample_pos_neg(lab, rh.BATCH_SIZE_PER_IMAGE, rh.POSITIVE_FRACTION))
        elif sample_for_da and len(boxes) <= rh.BATCH_SIZE_PER_IMAGE:
            # all labels are 0 and there are no more rows than the cap: every row is taken, and the product draws
            # nothing (FastRCNNLossComputation._subsample_for_da_fused)
            masks.append((torch.zeros(len(boxes), dtype=torch.bool), torch.ones(len(boxes), dtype=torch.bool)))
        else:
            masks.append(sample_pos_neg_device(lab, rh.BATCH_SIZE_PER_IMAGE, rh.POSITIVE_FRACTION, draws.seed()))
    out = []
    for (boxes, _), (lab, reg, src), (pm, nm) in zip(proposals, tg, masks):
        idx = torch.nonzero(pm | nm).squeeze(1)
        out.append(dict(boxes=boxes[idx], labels=lab[idx], reg=reg[idx],
                        domain=torch.full((idx.numel(),), src, dtype=torch.bool), idx=idx))
    return out


def roi_feature(feat, samples, sd, cfg):
    """poolers.py:78-121 + ResNetHead (roi_box_feature_extractors.py:42-45)"""
    bh = cfg.MODEL.ROI_BOX_HEAD
    rois = torch.cat([torch.cat([torch.full((len(s["boxes"]), 1), float(i)), s["boxes"]], 1)
                      for i, s in enumerate(samples)], 0)
    x = roi_align(feat, rois, bh.POOLER_SCALES[0], bh.POOLER_RESOLUTION, bh.POOLER_RESOLUTION,
                  bh.POOLER_SAMPLING_RATIO)
    return stage(x, sd, "roi_heads.box.feature_extractor.head.layer4", 3, 2)


def box_losses(x, samples, sd, cfg):
    """roi_box_predictors.py:28-33 + loss.py:165-221"""
    v = F.avg_pool2d(x, 7).flatten(1)
    logits = F.linear(v, sd["roi_heads.box.predictor.cls_score.weight"], sd["roi_heads.box.predictor.cls_score.bias"])
    reg = F.linear(v, sd["roi_heads.box.predictor.bbox_pred.weight"], sd["roi_heads.box.predictor.bbox_pred.bias"])
    labels = torch.cat([s["labels"] for s in samples])
    targets = torch.cat([s["reg"] for s in samples])
    dom = torch.cat([s["domain"] for s in samples])
    logits, reg, labels, targets = logits[dom], reg[dom], labels[dom], targets[dom]
    cls_loss = F.cross_entropy(logits, labels)
    pos = torch.nonzero(labels > 0).squeeze(1)
    cols = 4 * labels[pos][:, None] + torch.tensor([0, 1, 2, 3])
    box_loss = smooth_l1(reg[pos[:, None], cols], targets[pos], 1.0) / labels.numel()
    return cls_loss, box_loss, dom


# ---------------------------------------------------------------------------------------------- DA heads
def img_head(x, sd, p):
    """da_heads.py:32-37"""
    t = F.relu(F.conv2d(x, sd[p + ".imghead.conv1_da.weight"], sd[p + ".imghead.conv1_da.bias"]))
    return F.conv2d(t, sd[p + ".imghead.conv2_da.weight"], sd[p + ".imghead.conv2_da.bias"])


def ins_head(x, sd, p, training=True, draws=None):
    """da_heads.py:61-68 (dropout masks drawn exactly like F.dropout on the CPU, or replayed from `draws`)"""
    drop = (lambda t: t * draws.mask(t.shape).to(t.dtype)) if (draws is not None and training) else \
        (lambda t: F.dropout(t, p=0.5, training=training))
    x = F.relu(F.linear(x, sd[p + ".inshead.fc1_da.weight"], sd[p + ".inshead.fc1_da.bias"]))
    x = drop(x)
    x = F.relu(F.linear(x, sd[p + ".inshead.fc2_da.weight"], sd[p + ".inshead.fc2_da.bias"]))
    x = drop(x)
    return F.linear(x, sd[p + ".inshead.fc3_da.weight"], sd[p + ".inshead.fc3_da.bias"])


class _GRL(torch.autograd.Function):
    """layers/gradient_scalar_layer.py:4-13"""

    @staticmethod
    def forward(ctx, x, w):
        ctx.w = w
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.w * g, None


def img_bce(logits, img_labels):
    """da_heads/loss.py:80-98: per-pixel label = is_source of the image, mean over N*H*W"""
    lab = img_labels.view(-1, 1, 1, 1).expand_as(logits)
    return F.binary_cross_entropy_with_logits(logits.permute(0, 2, 3, 1).reshape(logits.shape[0], -1),
                                              lab.permute(0, 2, 3, 1).reshape(logits.shape[0], -1))


def consistency(img_sig, ins_sig, ins_labels):
    """layers/consistency_loss.py:3-27 (N == 2, source ROIs first)"""
    n_src = int(torch.nonzero(ins_labels).size(0))
    means = img_sig.reshape(img_sig.shape[0], -1).mean(1)
    rows = torch.cat([means[0].view(1, 1).repeat(n_src, 1), means[1].view(1, 1).repeat(ins_sig.size(0) - n_src, 1)], 0)
    return torch.abs(rows - ins_sig).mean()


def da_losses_plain(feat, ins_feat, ins_labels, img_labels, sd, cfg, draws=None):
    """DomainAdaptationModule.forward (da_heads.py:388-440)"""
    da = cfg.MODEL.DA_HEADS
    p = "da_heads"
    v = F.avg_pool2d(ins_feat, 7).flatten(1)
    img_logits = img_head(_GRL.apply(feat, -da.DA_IMG_GRL_WEIGHT), sd, p)
    ins_logits = ins_head(_GRL.apply(v, -da.DA_INS_GRL_WEIGHT), sd, p, draws=draws)
    img_cst = img_head(_GRL.apply(feat, da.DA_IMG_GRL_WEIGHT), sd, p).sigmoid()
    ins_cst = ins_head(_GRL.apply(v, da.DA_INS_GRL_WEIGHT), sd, p, draws=draws).sigmoid()
    out = {}
    if da.DA_IMG_LOSS_WEIGHT > 0:
        out["loss_da_image"] = da.DA_IMG_LOSS_WEIGHT * img_bce(img_logits, img_labels)
    if da.DA_INS_LOSS_WEIGHT > 0:
        out["loss_da_instance"] = da.DA_INS_LOSS_WEIGHT * F.binary_cross_entropy_with_logits(
            ins_logits.squeeze(), ins_labels.float())
    if da.DA_CST_LOSS_WEIGHT > 0:
        out["loss_da_consistency"] = da.DA_CST_LOSS_WEIGHT * consistency(img_cst, ins_cst, ins_labels)
    return out


def adv_weight(cur_loss, base, adv, threshold):
    """intended AdvGRL rule (da_heads.py:173-195)"""
    gate = float(F.binary_cross_entropy_with_logits(torch.tensor([[0.7, 0.3]]), torch.tensor([[1.0, 0.0]])))
    cur = float(cur_loss.detach())
    return -adv * min(float(threshold), 1.0 / cur) if cur <= gate else -base


def da_losses_triplet(feat2, ins_feat, ins_labels, img_labels, feat3, ins_set, state, sd, cfg, draws=None):
    """DomainAdaptationModule_triplet.forward (da_heads.py:293-344); `state` carries the adaptive margins and
    the previous triplet losses"""
    da = cfg.MODEL.DA_HEADS
    p = "da_heads_triplet"
    out = {}
    if da.DA_TRIPLET_INS_WEIGHT > 0:
        s, q, n = [F.avg_pool2d(f, 7).flatten(1) for f in ins_set]
        state["margin_ins"] = da.TRIPLET_MARGIN_INS
        loss = F.triplet_margin_loss(s, q, n, margin=state["margin_ins"], p=2)
        out["triplet_loss_instance"] = da.DA_TRIPLET_INS_WEIGHT * loss
    if da.DA_TRIPLET_IMG_WEIGHT > 0:
        if state.get("margin_img", 0.0) == 0.0:
            state["margin_img"] = da.TRIPLET_MARGIN_IMG
        if state.get("prev_img", 1) == 0.0 and int(state["margin_img"]) != int(da.TRIPLET_MAX_MARGIN):
            state["margin_img"] += 0.001
        loss = F.triplet_margin_loss(feat3[0:1], feat3[1:2], feat3[2:3], margin=state["margin_img"], p=2)
        out["triplet_loss_image"] = da.DA_TRIPLET_IMG_WEIGHT * loss
        state["prev_img"] = float(loss.detach())
    if da.DA_IMG_LOSS_WEIGHT > 0:
        cur = img_bce(img_head(feat2, sd, p).detach(), img_labels)
        w = adv_weight(cur, da.DA_IMG_GRL_WEIGHT, da.DA_IMG_advGRL_WEIGHT, da.DA_ADV_GRL_THRESHOLD) \
            if da.DA_ADV_GRL else -da.DA_IMG_GRL_WEIGHT
        out["loss_da_image"] = da.DA_IMG_LOSS_WEIGHT * img_bce(img_head(_GRL.apply(feat2, w), sd, p), img_labels)
    v = F.avg_pool2d(ins_feat, 7).flatten(1)
    if da.DA_INS_LOSS_WEIGHT > 0:
        cur = F.binary_cross_entropy_with_logits(ins_head(v.detach(), sd, p, draws=draws).squeeze(), ins_labels.float())
        w = adv_weight(cur, da.DA_INS_GRL_WEIGHT, da.DA_INS_advGRL_WEIGHT, da.DA_ADV_GRL_THRESHOLD) \
            if da.DA_ADV_GRL else -da.DA_INS_GRL_WEIGHT
        out["loss_da_instance"] = da.DA_INS_LOSS_WEIGHT * F.binary_cross_entropy_with_logits(
            ins_head(_GRL.apply(v, w), sd, p, draws=draws).squeeze(), ins_labels.float())
    if da.DA_CST_LOSS_WEIGHT > 0:
        img_cst = img_head(_GRL.apply(feat2, da.DA_IMG_GRL_WEIGHT), sd, p).sigmoid()
        ins_cst = ins_head(_GRL.apply(v, da.DA_INS_GRL_WEIGHT), sd, p, draws=draws).sigmoid()
        out["loss_da_consistency"] = da.DA_CST_LOSS_WEIGHT * consistency(img_cst, ins_cst, ins_labels)
    return out


# -------------------------------------------------------------------------------------------- full model
def box_head_pass(feat, proposals, gts, sd, cfg, with_losses=True, draws=None):
    """ROIBoxHead.forward in training (box_head.py:36-118)"""
    with torch.no_grad():
        samples = subsample(proposals, gts, cfg, draws=draws)
    x = roi_feature(feat, samples, sd, cfg)
    cls_loss, box_loss, dom = box_losses(x, samples, sd, cfg)
    with torch.no_grad():
        # the reference samples the DA ROIs from the ALREADY SUBSAMPLED proposals (box_head.py:102-104)
        sub = [(s["boxes"], None) for s in samples]
        da_samples = subsample(sub, gts, cfg, sample_for_da=True, draws=draws)
    da_feat = roi_feature(feat, da_samples, sd, cfg)
    return dict(loss_classifier=cls_loss, loss_box_reg=box_loss), da_feat, dom, samples, da_samples


def training_losses(sd, cfg, images, gts, state=None, intermediates=None, selection_maps=None, draws=None,
                    grad_probe=False, selection_proposals=None):
    """GeneralizedRCNN.forward in training mode (modeling/detector/generalized_rcnn.py:61-153).
    images [N,3,H,W] (already padded), gts: list of dict(boxes [G,4], labels [G], is_source [G] bool).
    selection_maps=(objectness, deltas): tests may feed the proposal SELECTION fixed RPN maps (e.g. the golden
    reference ones) so index-valued results do not depend on this machine's fp32 GEMM rounding; maps of fewer than N
    images cover the leading ones.  selection_proposals: the proposal lists themselves (see below)."""
    N, _, H, W = images.shape
    image_sizes = [(H, W)] * N
    feat = backbone_c4(images, sd)
    objectness, deltas = rpn_head(feat, sd)
    rpn = cfg.MODEL.RPN
    anchors = grid_anchors(feat.shape[2], feat.shape[3], rpn.ANCHOR_STRIDE[0],
                           cell_anchors(rpn.ANCHOR_STRIDE[0], rpn.ANCHOR_SIZES, rpn.ASPECT_RATIOS))
    with torch.no_grad():
        sel_obj, sel_del = selection_maps if selection_maps is not None else (objectness, deltas)
        if sel_obj.shape[0] < N:
            # maps given for the leading images only (the product evaluates the RPN head on the images whose proposals
            # are read): the others are selected from this restatement's own maps, as the reference would
            k = sel_obj.shape[0]
            sel_obj = torch.cat([sel_obj.to(objectness.dtype), objectness[k:].detach()], dim=0)
            sel_del = torch.cat([sel_del.to(deltas.dtype), deltas[k:].detach()], dim=0)
        proposals = rpn_proposals(sel_obj, sel_del, anchors, image_sizes, gts, cfg, True)
        if selection_proposals is not None:
            # proposal lists handed in (boxes [P,4], objectness [P]) per image, None = keep this restatement's own.  The
            # ranking behind them sorts SIGMOID scores: thousands of anchors share a saturated fp32 value and are ordered
            # by index, and which logits collapse onto one value differs by an ulp between two sigmoid implementations
            # (and entirely between float32 and float64) — two devices then agree on the proposal SET but may swap
            # neighbours, and an index into the list means another box ("near-tied scores")
            proposals = [p if q is None else (q[0].to(p[0].dtype), q[1].to(p[1].dtype))
                         for p, q in zip(proposals, list(selection_proposals) + [None] * (N - len(selection_proposals)))]
    obj_loss, rpn_box_loss = rpn_losses(objectness, deltas, anchors, image_sizes, gts, cfg, draws, intermediates)
    img_labels = torch.tensor([1.0 if g["is_source"].any() else 0.0 for g in gts])
    losses = {}
    if cfg.MODEL.DA_HEADS.TRIPLET_USE:
        det, da_feat, dom, samples, da_samples = box_head_pass(feat[0:2], proposals[0:2], gts[0:2], sd, cfg, draws=draws)
        ins_set = None
        if cfg.MODEL.DA_HEADS.ALIGNMENT:
            ins_set = []
            for k in range(3):
                _, f_k, _, _, _ = box_head_pass(feat[k:k + 1], [proposals[1]], [gts[k]], sd, cfg, draws=draws)
                ins_set.append(f_k)
        da = da_losses_triplet(feat[0:2], da_feat, dom, img_labels[0:2], feat, ins_set,
                               state if state is not None else {}, sd, cfg, draws)
    else:
        det, da_feat, dom, samples, da_samples = box_head_pass(feat, proposals, gts, sd, cfg, draws=draws)
        da = da_losses_plain(feat, da_feat, dom, img_labels, sd, cfg, draws)
    losses.update(det)
    losses.update({"loss_objectness": obj_loss, "loss_rpn_box_reg": rpn_box_loss})
    losses.update(da)
    if intermediates is not None:
        intermediates.update(feat=feat.detach(), objectness=objectness.detach(), deltas=deltas.detach(),
                             proposals=[(b.clone(), s.clone()) for b, s in proposals],
                             sampled_idx=[s["idx"] for s in samples], da_sampled_idx=[s["idx"] for s in da_samples],
                             da_feat=da_feat.detach())
        if grad_probe:   # graph-attached tensors for torch.autograd.grad(loss, tensor) probes (AdvGRL fixtures)
            intermediates.update(feat_graph=feat, ins_feat_graph=da_feat)
    return losses


# ---------------------------------------------------------------------------------------------- evaluation
def post_process(logits, reg, proposals, image_sizes, cfg):
    """PostProcessor.forward / filter_results (roi_heads/box_head/inference.py:43-150): softmax, per-class decode
    with BBOX_REG_WEIGHTS, clip, score threshold, per-class NMS, top DETECTIONS_PER_IMG by kthvalue threshold.
    -> per image dict(boxes [D,4], scores [D], labels [D] int64), classes concatenated in ascending order"""
    rh = cfg.MODEL.ROI_HEADS
    prob = F.softmax(logits, -1)
    counts = [len(b) for b, _ in proposals]
    boxes = decode_multi(reg.reshape(sum(counts), -1), torch.cat([b for b, _ in proposals], 0), rh.BBOX_REG_WEIGHTS)
    ncls = prob.shape[1]
    out = []
    for p_i, b_i, (h, w) in zip(prob.split(counts, 0), boxes.split(counts, 0), image_sizes):
        b_i = b_i.reshape(-1, 4)
        b_i = torch.stack([b_i[:, 0].clamp(0, w - 1), b_i[:, 1].clamp(0, h - 1), b_i[:, 2].clamp(0, w - 1),
                           b_i[:, 3].clamp(0, h - 1)], 1).reshape(-1, ncls * 4)
        rb, rs, rl = [], [], []
        for j in range(1, ncls):
            inds = torch.nonzero(p_i[:, j] > rh.SCORE_THRESH).squeeze(1)
            bj, sj = b_i[inds, 4 * j:4 * j + 4], p_i[inds, j]
            keep = torch.from_numpy(O.nms(bj.numpy(), sj.numpy(), rh.NMS, 0))
            rb.append(bj[keep]), rs.append(sj[keep]), rl.append(torch.full((len(keep),), j, dtype=torch.int64))
        rb, rs, rl = torch.cat(rb), torch.cat(rs), torch.cat(rl)
        if len(rs) > rh.DETECTIONS_PER_IMG > 0:
            thr, _ = torch.kthvalue(rs, len(rs) - rh.DETECTIONS_PER_IMG + 1)
            keep = torch.nonzero(rs >= thr.item()).squeeze(1)
            rb, rs, rl = rb[keep], rs[keep], rl[keep]
        out.append(dict(boxes=rb, scores=rs, labels=rl))
    return out


def inference(sd, cfg, images, intermediates=None, selection_maps=None):
    """GeneralizedRCNN.forward in eval mode (generalized_rcnn.py:61-70,145-156; box_head.py:58-67 eval branch):
    backbone -> RPN (PRE/POST_NMS_TOP_N_TEST, no GT appended) -> ROIAlign + res5 on every proposal -> predictor
    -> PostProcessor."""
    with torch.no_grad():
        N, _, H, W = images.shape
        image_sizes = [(H, W)] * N
        feat = backbone_c4(images, sd)
        objectness, deltas = rpn_head(feat, sd)
        rpn = cfg.MODEL.RPN
        anchors = grid_anchors(feat.shape[2], feat.shape[3], rpn.ANCHOR_STRIDE[0],
                               cell_anchors(rpn.ANCHOR_STRIDE[0], rpn.ANCHOR_SIZES, rpn.ASPECT_RATIOS))
        sel_obj, sel_del = selection_maps if selection_maps is not None else (objectness, deltas)
        proposals = rpn_proposals(sel_obj, sel_del, anchors, image_sizes, None, cfg, False)
        x = roi_feature(feat, [dict(boxes=b) for b, _ in proposals], sd, cfg)
        v = F.avg_pool2d(x, 7).flatten(1)
        logits = F.linear(v, sd["roi_heads.box.predictor.cls_score.weight"],
                          sd["roi_heads.box.predictor.cls_score.bias"])
        reg = F.linear(v, sd["roi_heads.box.predictor.bbox_pred.weight"],
                       sd["roi_heads.box.predictor.bbox_pred.bias"])
        if intermediates is not None:
            intermediates.update(objectness=objectness, deltas=deltas, proposals=proposals, class_logits=logits,
                                 box_regression=reg)
        return post_process(logits, reg, proposals, image_sizes, cfg)


# ------------------------------------------------------------------------------------------- FPN variant
def backbone_fpn(images, sd, blocks=(3, 4, 6, 3)):
    """R-50-FPN body + FPN (backbone.py:21-42, fpn.py:43-79): C2..C5 -> (P2, P3, P4, P5, P6)"""
    p = "backbone.body"
    x = F.relu(frozen_bn(F.conv2d(images, sd[p + ".stem.conv1.weight"], None, 2, 3), sd, p + ".stem.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    cs = []
    for i, nb in enumerate(blocks):
        x = stage(x, sd, "%s.layer%d" % (p, i + 1), nb, 1 if i == 0 else 2)
        cs.append(x)
    f = "backbone.fpn.fpn_"
    conv = lambda name, t, pad: F.conv2d(t, sd[f + name + ".weight"], sd[f + name + ".bias"], 1, pad)
    last = conv("inner4", cs[3], 0)
    out = [conv("layer4", last, 1)]
    for lvl in (3, 2, 1):
        last = conv("inner%d" % lvl, cs[lvl - 1], 0) + F.interpolate(last, scale_factor=2, mode="nearest")
        out.insert(0, conv("layer%d" % lvl, last, 1))
    out.append(F.max_pool2d(out[-1], 1, 2, 0))
    return out


def fpn_anchors(objectness, cfg):
    """per-level anchor grids of a pyramid (anchor_generator.py:73-125: one size per level, all aspect ratios)"""
    rpn = cfg.MODEL.RPN
    return [grid_anchors(o.shape[2], o.shape[3], rpn.ANCHOR_STRIDE[l],
                         cell_anchors(rpn.ANCHOR_STRIDE[l], (rpn.ANCHOR_SIZES[l],), rpn.ASPECT_RATIOS))
            for l, o in enumerate(objectness)]


def rpn_proposals_fpn_train(objectness, deltas, image_sizes, gts, cfg):
    """training-mode RPNPostProcessor over pyramid levels (rpn/inference.py:124-181): per level and image top
    PRE_NMS_TOP_N_TRAIN -> decode/clip -> NMS -> first POST_NMS_TOP_N_TRAIN; levels concatenated per image; then
    FPN_POST_NMS_TOP_N_TRAIN best scores over the WHOLE BATCH (:161-172 — a mask, so every image keeps its survivors
    in their level-major order); ground-truth boxes appended with objectness 1 (:51-74).  Ties: lower position first."""
    rpn = cfg.MODEL.RPN
    N = objectness[0].shape[0]
    per_image = [[] for _ in range(N)]
    for lvl, (obj, dlt, anchors) in enumerate(zip(objectness, deltas, fpn_anchors(objectness, cfg))):
        scores = flatten_hwa(obj, 1).reshape(N, -1).sigmoid()
        d = flatten_hwa(dlt, 4)
        pre = min(rpn.PRE_NMS_TOP_N_TRAIN, scores.shape[1])
        for i in range(N):
            s, order = torch.sort(scores[i], descending=True, stable=True)
            s, order = s[:pre], order[:pre]
            h, w = image_sizes[i]
            boxes = torch.from_numpy(O.decode_clip(d[i][order].numpy(), anchors[order].numpy(),
                                                   (1.0, 1.0, 1.0, 1.0), math.log(1000.0 / 16), w, h))
            keep = torch.from_numpy(O.nms(boxes.numpy(), s.numpy(), rpn.NMS_THRESH, 0))[:rpn.POST_NMS_TOP_N_TRAIN]
            per_image[i].append((boxes[keep], s[keep]))
    cat = [(torch.cat([b for b, _ in lv]), torch.cat([x for _, x in lv])) for lv in per_image]
    all_scores = torch.cat([s for _, s in cat])
    k = min(rpn.FPN_POST_NMS_TOP_N_TRAIN, all_scores.numel())
    _, inds = torch.sort(all_scores, descending=True, stable=True)
    mask = torch.zeros_like(all_scores, dtype=torch.bool)
    mask[inds[:k]] = True
    out = []
    for (boxes, s), m, gt in zip(cat, mask.split([len(s) for _, s in cat]), gts):
        boxes, s = boxes[m], s[m]
        boxes = torch.cat([boxes, gt["boxes"]], 0)
        s = torch.cat([s, torch.ones(len(gt["boxes"]))], 0)
        out.append((boxes, s))
    return out


def rpn_proposals_fpn(objectness, deltas, image_sizes, cfg):
    """test-mode RPNPostProcessor over pyramid levels (rpn/inference.py:76-181): per level top PRE_NMS_TOP_N_TEST
    -> decode/clip -> NMS -> first POST_NMS_TOP_N_TEST; concatenate levels; per image top FPN_POST_NMS_TOP_N_TEST
    by objectness (ties: ascending position, as a stable sort)"""
    rpn = cfg.MODEL.RPN
    N = objectness[0].shape[0]
    per_image = [[] for _ in range(N)]
    for lvl, (obj, dlt) in enumerate(zip(objectness, deltas)):
        stride = rpn.ANCHOR_STRIDE[lvl]
        anchors = grid_anchors(obj.shape[2], obj.shape[3], stride,
                               cell_anchors(stride, (rpn.ANCHOR_SIZES[lvl],), rpn.ASPECT_RATIOS))
        scores = flatten_hwa(obj, 1).reshape(N, -1).sigmoid()
        d = flatten_hwa(dlt, 4)
        pre = min(rpn.PRE_NMS_TOP_N_TEST, scores.shape[1])
        for i in range(N):
            s, order = torch.sort(scores[i], descending=True, stable=True)
            s, order = s[:pre], order[:pre]
            h, w = image_sizes[i]
            boxes = torch.from_numpy(O.decode_clip(d[i][order].numpy(), anchors[order].numpy(),
                                                   (1.0, 1.0, 1.0, 1.0), math.log(1000.0 / 16), w, h))
            keep = torch.from_numpy(O.nms(boxes.numpy(), s.numpy(), rpn.NMS_THRESH, 0))[:rpn.POST_NMS_TOP_N_TEST]
            per_image[i].append((boxes[keep], s[keep]))
    out = []
    for lv in per_image:
        boxes, s = torch.cat([b for b, _ in lv]), torch.cat([x for _, x in lv])
        k = min(rpn.FPN_POST_NMS_TOP_N_TEST, len(s))
        _, inds = torch.sort(s, descending=True, stable=True)
        out.append((boxes[inds[:k]], s[inds[:k]]))
    return out


def pool_fpn(feats, proposals, cfg):
    """Pooler with LevelMapper (poolers.py:11-42, 91-121): level = floor(4 + log2(sqrt(area)/224 + 1e-6)) clamped"""
    bh = cfg.MODEL.ROI_BOX_HEAD
    scales = bh.POOLER_SCALES
    k_min, k_max = -math.log2(scales[0]), -math.log2(scales[-1])
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, (b, _) in enumerate(proposals)], 0)
    s = torch.sqrt(box_area(rois[:, 1:]))
    lvl = torch.clamp(torch.floor(4 + torch.log2(s / 224 + 1e-6)), min=k_min, max=k_max).to(torch.int64) - int(k_min)
    res = bh.POOLER_RESOLUTION
    out = torch.zeros((len(rois), feats[0].shape[1], res, res))
    for l, (feat, sc) in enumerate(zip(feats, scales)):
        idx = torch.nonzero(lvl == l).squeeze(1)
        if len(idx):
            out[idx] = roi_align(feat, rois[idx], sc, res, res, bh.POOLER_SAMPLING_RATIO)
    return out, lvl


def inference_fpn(sd, cfg, images, intermediates=None, selection_maps=None):
    """eval forward of the FPN Faster R-CNN (configs/e2e_faster_rcnn_R_50_FPN_1x.yaml): FPN backbone, shared RPN head
    on 5 levels, level-mapped ROIAlign, FPN2MLP head (roi_box_feature_extractors.py:48-79), FPNPredictor"""
    with torch.no_grad():
        N, _, H, W = images.shape
        image_sizes = [(H, W)] * N
        feats = backbone_fpn(images, sd)
        maps = [rpn_head(f, sd) for f in feats]
        objectness, deltas = [m[0] for m in maps], [m[1] for m in maps]
        sel_obj, sel_del = selection_maps if selection_maps is not None else (objectness, deltas)
        proposals = rpn_proposals_fpn(sel_obj, sel_del, image_sizes, cfg)
        pooled, lvl = pool_fpn(feats[:len(cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES)], proposals, cfg)
        fe = "roi_heads.box.feature_extractor."
        x = F.relu(F.linear(pooled.flatten(1), sd[fe + "fc6.weight"], sd[fe + "fc6.bias"]))
        x = F.relu(F.linear(x, sd[fe + "fc7.weight"], sd[fe + "fc7.bias"]))
        logits = F.linear(x, sd["roi_heads.box.predictor.cls_score.weight"],
                          sd["roi_heads.box.predictor.cls_score.bias"])
        reg = F.linear(x, sd["roi_heads.box.predictor.bbox_pred.weight"],
                       sd["roi_heads.box.predictor.bbox_pred.bias"])
        if intermediates is not None:
            intermediates.update(features=feats, objectness=objectness, deltas=deltas, proposals=proposals,
                                 levels=lvl, class_logits=logits, box_regression=reg)
        return post_process(logits, reg, proposals, image_sizes, cfg)


# ------------------------------------------------------------------------------------------ CPU baseline
def targets_to_dicts(targets):
    return [dict(boxes=t.bbox.cpu(), labels=t.get_field("labels").cpu(), is_source=t.get_field("is_source").cpu())
            for t in targets]


def timed_training_sample(cfg_path, seed, height, width, images_per_step, init_fn, budget_s=90.0):
    """cpu_baseline leg of bench.py: one full training step (forward + backward + SGD) of this restatement on
    the host cores, on the GPU workload's own batch (same config, same 1024x2048 images, same ROI counts)."""
    from da_detect_amd.config import cfg as base_cfg
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model

    cfg = base_cfg.clone()
    cfg.merge_from_file(cfg_path)
    torch.manual_seed(seed)
    model = build_detection_model(cfg)  # parameter container only (names / shapes / trainable flags)
    init_fn(model, seed)
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    params = []
    for k in trainable:
        sd[k].requires_grad_(True)
        params.append(sd[k])
    opt = torch.optim.SGD(params, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM,
                          weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    # oneDNN scales poorly past ~64 threads on these layer sizes; the count actually used is what is reported
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)

    def one_step(h, w):
        images, targets = make_batch(cfg, images_per_step, h, w, seed=seed, device=torch.device("cpu"))
        gts = targets_to_dicts(targets)
        t0 = time.perf_counter()
        losses = training_losses(sd, cfg, images.tensors, gts)
        total = sum(losses.values())
        opt.zero_grad()
        total.backward()
        opt.step()
        return time.perf_counter() - t0

    # One warm-up step on 1/16 of the pixels (allocator, oneDNN primitive caches, thread pool), then the GPU workload's own
    # batch if the estimate fits the budget (measured on the GPU box's EPYC 9575F, 64 threads: 2.2 s and 21 s).  If it
    # does not fit, `value` stays None — an images/s figure on other images is not the baseline — and the per-size
    # samples plus a pixel-proportional extrapolation are reported instead.
    samples = []
    dt = one_step(height // 4, width // 4)
    samples.append({"image_hw": [height // 4, width // 4], "seconds": round(dt, 3)})
    spent = dt
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    base = {"unit": "images/s", "cores": cores, "host_threads_visible": os.cpu_count(), "cpu_model": cpu_model,
            "kind": "port"}
    what = ("1 training step (fwd+bwd+SGD) of oracle/model_ref.py; torch CPU fp32 convs on %d host threads, "
            "single-thread C NMS/ROIAlign like the reference's CPU operators" % cores)
    if spent < budget_s / 3.0:      # the warm-up step is dominated by first-use costs (~10 s): it predicts nothing
        full = one_step(height, width)
        samples.append({"image_hw": [height, width], "seconds": round(full, 3)})
        base.update(value=round(images_per_step / full, 4), seconds=round(full, 2), samples=samples,
                    sample="%s on the GPU workload's own batch: %d images of %dx%d (after one warm-up step at %dx%d)"
                           % (what, images_per_step, height, width, height // 4, width // 4))
        return base
    div = 4
    while div > 2 and spent + dt * 4.5 < budget_s:
        div //= 2
        dt = one_step(height // div, width // div)
        spent += dt
        samples.append({"image_hw": [height // div, width // div], "seconds": round(dt, 3)})
    base.update(value=None, samples=samples,
                extrapolated_images_per_s=round(images_per_step / (dt * div * div), 4),
                sample="%s; the full %dx%d batch did not fit the %.0f s budget: per-size samples and a pixel-proportional "
                       "extrapolation from %dx%d (an upper bound on the time: the ROI work does not grow with the "
                       "image)" % (what, height, width, budget_s, height // div, width // div))
    return base
