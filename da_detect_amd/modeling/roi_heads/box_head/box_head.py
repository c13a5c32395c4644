"""Box head with the extra domain-adaptation ROI pass
(reference: maskrcnn_benchmark/modeling/roi_heads/box_head/box_head.py:22-118)."""
import logging
import os

import torch

from ....structures.bounding_box import is_source_image
from ....utils import rng
from ...elision import elision_enabled, leading_source_images
from ....utils.streams import side_section
from .inference import make_roi_box_post_processor
from .loss import make_roi_box_loss_evaluator
from .roi_box_feature_extractors import make_roi_box_feature_extractor
from .roi_box_predictors import make_roi_box_predictor


_NO_DEDUP = os.environ.get("DADET_NO_ROI_DEDUP", "0") == "1"
# queue the pooler + res5 head before the sampled counts have reached the host (ROIBoxHead.forward); 0: after them
_SPECULATE = os.environ.get("DADET_ROI_SPECULATE", "1") == "1"


class ROIBoxHead(torch.nn.Module):
    def __init__(self, cfg):
        super(ROIBoxHead, self).__init__()
        self.feature_extractor = make_roi_box_feature_extractor(cfg)
        self.predictor = make_roi_box_predictor(cfg)
        self.post_processor = make_roi_box_post_processor(cfg)
        self.loss_evaluator = make_roi_box_loss_evaluator(cfg)
        self.proposals_ready = None   # optional event of the compute stream: the proposals passed to forward exist
        # set by the caller for ONE call: nothing will read the instance-level features of this call (see forward)
        self.ins_features_unused = False
        self.speculate = True
        self.speculation = {"kept": 0, "dropped": 0}

    def _note_speculation(self, kept):
        """bookkeeping of the early-queued box head (forward): a dropped result is a whole pooler + res5 forward thrown away.
        A configuration that rarely fills BATCH_SIZE_PER_IMAGE (a low post-NMS top-n, few proposals early in training) would
        pay for it every step — after 16 steps with more results dropped than kept the early queue switches itself off for
        this module and says so once (`speculation` holds the counts; DADET_ROI_SPECULATE=0 never starts it)."""
        st = self.speculation
        st["kept" if kept else "dropped"] += 1
        if not kept and st["dropped"] == 1:
            logging.getLogger("maskrcnn_benchmark.trainer").info(
                "box head: an early-queued pooler + res5 pass was dropped (an image filled fewer than "
                "BATCH_SIZE_PER_IMAGE rows); counting")
        if st["kept"] + st["dropped"] >= 16 and st["dropped"] > st["kept"] and self.speculate:
            self.speculate = False
            logging.getLogger("maskrcnn_benchmark.trainer").warning(
                "box head: early queue switched off (%d of %d results dropped)" % (st["dropped"], st["kept"] + st["dropped"]))

    def forward(self, features, proposals, targets=None):
        """-> (x, proposals | detections, losses, da_ins_feas, da_ins_labels).  Training runs two passes of
        pooler + res5 + predictor: the sampled detection ROIs, then BATCH_SIZE_PER_IMAGE uniformly sampled
        ROIs per image whose features / domain labels feed the instance-level domain classifier."""
        burn = 0
        if self.training:
            unused, self.ins_features_unused = self.ins_features_unused, False
            # Rows of target-domain images: the detection losses mask them out (box_head/loss.py:193-198) and their only
            # other reader is the instance-level domain classifier.  When that one has no loss either (instance and
            # consistency weights 0: the reference evaluates it and drops the result, da_heads.py:402-439; SURVEY.md
            # appendix A "may elide"), sampling them, pooling them and running res5 + predictor forward and backward
            # over them changes no loss and no gradient — every one of their gradient rows is exactly 0.  They are left
            # out: this call then sees the leading source-domain images only (source images come first in a batch,
            # engine/trainer.py:215-224; the pooler's batch index is the list position).  One sampler seed per skipped
            # image is still drawn, so that the images that ARE sampled get the seeds they would get anyway.
            # Not with the reference's random stream (utils.rng.use_cpu_stream): there every draw keeps its size.
            if unused and elision_enabled():
                n_src = leading_source_images(targets)
                if 0 < n_src < len(targets):
                    burn = len(targets) - n_src
                    proposals, targets = proposals[:n_src], targets[:n_src]
            # sampling runs on the side stream: its host round trips then do not wait for the RPN-head backward
            # queued on the compute stream just before (RPNModule.early_backward)
            after, self.proposals_ready = self.proposals_ready, None
            dev = features[0].device      # (not proposals[0].bbox: that would materialise a device-resident list)
            le = self.loss_evaluator
            state = x = None
            with side_section(dev, after=after) as done, torch.no_grad():
                if _SPECULATE and self.speculate and dev.type == "cuda":
                    state = le.subsample_launch(proposals, targets)
                if state is not None:
                    done(state["buf"], state["counts"])
                else:
                    proposals = le.subsample(proposals, targets)
                    for _ in range(burn):
                        rng.next_seed(dev)
                    # the reference draws the DA ROI sample after the detection losses (box_head.py:102-104); nothing
                    # between the two draws from the random stream, so drawing it here is the same sample — and it keeps
                    # every host synchronisation of the box head in front of the res5 head instead of behind it
                    da_proposals = le.subsample_for_da(proposals, targets)
                    done(proposals, da_proposals, le._proposals, le._loss_prep)
            if state is not None:
                # The sample's size reaches the host ~0.2 ms after the sampler ends (copy, wake-up, the Python between here
                # and the first launch of the res5 head), and the GPU has nothing else to run by then (tools/probes/
                # host_lead.py: lead 0 at this point, 3 - 6 ms everywhere else).  Every image fills its BATCH_SIZE_PER_IMAGE
                # rows in all but degenerate steps, so the pooler and the res5 head are QUEUED NOW on that assumption —
                # behind the sampler on the GPU, before the host knows the counts — and kept only if the counts confirm it;
                # otherwise the result is dropped (nothing but a tensor was produced) and the exact lists are pooled below.
                if state["speculative"] is not None:
                    x = self.feature_extractor(features, state["speculative"])
                with torch.no_grad():
                    proposals = le.subsample_finish(state)      # the host waits for the counts' copy, nothing else
                    for _ in range(burn):
                        rng.next_seed(dev)
                    # the sampled lists hold at most BATCH_SIZE_PER_IMAGE rows each: the DA sample takes them all without a
                    # host round trip (loss.py: _subsample_for_da_fused, n <= cap) — it must, the res5 head is queued in front
                    assert all(len(p) <= state["cap"] for p in proposals)
                    da_proposals = le.subsample_for_da(proposals, targets)
                if state["speculative"] is not None:
                    self._note_speculation(bool(state["exact"]))
                if not state["exact"]:
                    x = None
        else:
            x = None
        if x is None:
            x = self.feature_extractor(features, proposals)
        class_logits, box_regression = self.predictor(x)
        if not self.training:
            return x, self.post_processor((class_logits, box_regression), proposals), {}, x, None
        loss_classifier, loss_box_reg, _ = self.loss_evaluator([class_logits], [box_regression])
        # The reference samples the DA ROIs FROM THE ALREADY SUBSAMPLED `proposals` (box_head.py:50-104: the name
        # is rebound by subsample()).  With every label forced to 0 the sampler takes min(n, BATCH_SIZE_PER_IMAGE)
        # "negatives"; n <= BATCH_SIZE_PER_IMAGE here, so it takes ALL of them, in ascending index order: the DA
        # ROI set is the detection ROI set, and the reference's second pooler + res5 pass recomputes `x` bit for
        # bit.  The identical sub-expression is evaluated once; its two consumers' gradients add up in autograd
        # exactly as the two passes' parameter gradients would.  (subsample_for_da still runs: it draws from the
        # random stream.)  Set DADET_NO_ROI_DEDUP=1 to execute the redundant pass.
        if burn:
            return (x, proposals, dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg), None, None)
        limit = self.loss_evaluator.fg_bg_sampler.batch_size_per_image
        same_set = all(len(p) <= limit for p in proposals) and not _NO_DEDUP
        if same_set:
            assert all(len(a) == len(b) for a, b in zip(da_proposals, proposals))
            da_ins_feas = x
        else:
            da_ins_feas = self.feature_extractor(features, da_proposals)
        # the reference runs the predictor on the DA features too and only keeps the domain mask, which
        # depends on self._proposals alone (box_head.py:107-110); the dead predictor call is skipped
        da_ins_labels = torch.cat([p.get_field("domain_labels") for p in self.loss_evaluator._proposals], dim=0)
        # number of source-domain rows, known on the host: spares consistency_loss its nonzero() round trip
        da_ins_labels._n_src_host = sum(len(p) for p, t in zip(self.loss_evaluator._proposals, targets)
                                        if is_source_image(t))
        return (x, proposals, dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg), da_ins_feas,
                da_ins_labels)


def build_roi_box_head(cfg):
    return ROIBoxHead(cfg)
