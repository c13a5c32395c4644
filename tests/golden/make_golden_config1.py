"""Golden vectors for BASELINE.json configs[0] — e2e_faster_rcnn_R_50_C4_1x.yaml (plain Faster R-CNN R-50-C4, 81 classes,
no DA heads, DATALOADER.SIZE_DIVISIBILITY 0) — runs ONLY in the authoring container (needs /root/reference).

The reference yields eval-mode detections for this configuration (generalized_rcnn.py:134-136; training raises
UnboundLocalError).  Two fixtures from the imported reference on the CPU:
  eval_c4_plain_small.npz   2 x 200 x 333 images: stride-16 map 13 x 21 — odd sizes at every stage, nothing divisible by 32
  eval_c4_plain_full.npz    2 x 800 x 1333 images, the configuration's own size: C4 map 50 x 84
Same contents and the same oracle pinning as make_golden_eval.py (whose run_case does the work).

    python tests/golden/make_golden_config1.py [small|full]
"""
import sys

import make_golden_eval as E
from oracle import model_ref

YAML = "/root/reference/configs/e2e_faster_rcnn_R_50_C4_1x.yaml"
SIZES = {"small": (200, 333), "full": (800, 1333)}


def main():
    for name in (sys.argv[1:] or list(SIZES)):
        H, W = SIZES[name]
        for seed in range(8):
            try:
                E.run_case("eval_c4_plain_" + name, YAML, model_ref.inference, H=H, W=W, nimg=2, seed=seed)
                break
            except AssertionError as e:
                # exactly tied scores (two sliver ROIs on the image border pool identical features): the reference's
                # unstable sorts leave their order unspecified -> next seed
                if str(e) not in ("proposals", "class logits", "detections"):
                    raise
                print("%s: seed %d has tied proposal scores (%s) -> next seed" % (name, seed, e))
        else:
            raise SystemExit("no tie-free seed for " + name)


if __name__ == "__main__":
    main()
