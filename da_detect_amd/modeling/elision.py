"""Work of a training step whose results nothing reads, and that is therefore left out (same losses, same gradients).

The reference evaluates every head on every image and then masks or drops what a recipe does not use:
  * detection losses take source-domain rows only (roi_heads/box_head/loss.py:193-198), so ROIs of target-domain images
    matter only as input of the instance-level domain classifier — and not at all when its loss weights are 0
    (da_heads.py:402-439 computes the head and discards it; SURVEY.md appendix A: "may elide");
  * RPN losses label anchors of source-domain images only (rpn/loss.py:57-98), so the gradient of every other image's
    objectness / regression map is identically zero;
  * a triplet batch's auxiliary image never reaches the box head (generalized_rcnn.py:100: proposals[0:2]).
ROIBoxHead.forward, RPNModule._forward_train_overlapped and GeneralizedRCNN._images_with_read_proposals act on this.
DADET_DEAD_ROI_ROWS=1 evaluates everything, like the reference; so does utils.rng.use_cpu_stream (the reference's random
stream: every draw keeps its size).  tests/test_default_path_gpu.py compares both ways with each other and the lean one
with the oracle, which always evaluates everything.
"""
import os

from ..structures.bounding_box import is_source_image
from ..utils import rng

_KEEP_DEAD_ROWS = os.environ.get("DADET_DEAD_ROI_ROWS", "0") == "1"


def elision_enabled():
    return not _KEEP_DEAD_ROWS and not rng.cpu_stream_enabled()


def leading_source_images(targets):
    """number of source-domain images at the front of the batch (engine/trainer.py:215-224 puts them first); 0 when they
    are not a prefix"""
    src = [is_source_image(t) for t in targets]
    n = sum(src)
    return n if all(src[:n]) else 0
