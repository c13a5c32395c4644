"""is the res5 head bit-reproducible when it runs several times back to back (the aligned-triplet passes)?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402
from da_detect_amd.config import cfg as base  # noqa: E402
from da_detect_amd.modeling.roi_heads.box_head import roi_box_feature_extractors as fe  # noqa: E402

dev = torch.device("cuda", 0)
c = base.clone()
c.merge_from_file(os.path.join(ROOT, "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_img_only.yaml"))
torch.manual_seed(3)
ext = fe.make_roi_box_feature_extractor(c).to(dev)
g = torch.Generator().manual_seed(5)
for R in (256, 512):
    xs = [torch.randn((R, 1024, 7, 7), generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    with torch.no_grad():
        ref = [ext.head(x, first_stride=1).clone() for x in xs]
        torch.cuda.synchronize()
        bad = 0
        for rep in range(20):
            outs = [ext.head(x, first_stride=1) for x in xs]      # three different inputs back to back, like the aligned passes
            for o, r in zip(outs, ref):
                if not torch.equal(o, r):
                    bad += 1
                    d = (o - r).abs()
                    rows = (d.flatten(1).max(dim=1).values > 0).nonzero().flatten()
                    print("   R=%d rep %d: %d rows differ, max abs %.3e, first rows %s" % (R, rep, rows.numel(), float(d.max()), rows[:8].tolist()))
        print("R=%d STREAMK_SMALL=%s STREAMK=%s: %d of 60 outputs differ from the first evaluation" % (
            R, os.environ.get("DADET_STREAMK_SMALL", "1"), os.environ.get("DADET_STREAMK", "1"), bad))
