#!/usr/bin/env python
"""Evaluation entry point with the flow of the reference's tools/test_net.py on this package: build the detector, load
the weights (`--ckpt`, else cfg.MODEL.WEIGHT) through the suffix-matching checkpointer, run every test image through
the eval path (backbone -> RPN test-mode selection -> box head -> per-class NMS) and write the COCO-style detection
records (`bbox.json`, `predictions.pth`) to the output folder.  One process per GPU:

    python tools/test_net_da.py --config-file configs/da_faster_rcnn/<yaml> --dataset ann.json,imgdir \\
        [--ckpt model_final.pth] [--output-dir out] [KEY VALUE ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/test_net_da.py ...

Differences to the reference script, on purpose: datasets are named by an (annotation file, image root) pair instead
of the path catalog, and scoring (pycocotools mAP — a third-party package outside this repository) is left to the
caller: the records written here are exactly what `COCO.loadRes` consumes (engine/inference.py)."""
import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from da_detect_amd.config import cfg  # noqa: E402
from da_detect_amd.data.build import make_test_data_loader  # noqa: E402
from da_detect_amd.data.datasets import COCODataset  # noqa: E402
from da_detect_amd.data.transforms import build_transforms  # noqa: E402
from da_detect_amd.engine.inference import inference  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.utils.checkpoint import DetectronCheckpointer  # noqa: E402
from da_detect_amd.utils.comm import get_rank, synchronize  # noqa: E402


def _pair(text):
    ann, root = text.split(",")
    return ann, root


def main():
    ap = argparse.ArgumentParser(description="DA Faster R-CNN evaluation on MI355X")
    ap.add_argument("--config-file", required=True)
    ap.add_argument("--dataset", type=_pair, required=True, help="annotation.json,image_root of the test set")
    ap.add_argument("--ckpt", default=None, help="checkpoint to evaluate (default: cfg.MODEL.WEIGHT)")
    ap.add_argument("--output-dir", default=None)
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE overrides of the yaml")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("test_net_da.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")   # RCCL on ROCm
        synchronize()

    c = cfg.clone()
    c.merge_from_file(args.config_file)
    c.merge_from_list(args.opts)
    c.freeze()
    logging.basicConfig(level=logging.INFO if get_rank() == 0 else logging.WARNING)
    log = logging.getLogger("maskrcnn_benchmark.test_net")
    device = torch.device("cuda", local_rank)

    model = build_detection_model(c).to(device)
    output_dir = args.output_dir or c.MODEL.OUTPUT_DIR      # the tree has no top-level OUTPUT_DIR (defaults.py:427)
    checkpointer = DetectronCheckpointer(c, model, save_dir=output_dir)
    checkpointer.load(args.ckpt if args.ckpt else c.MODEL.WEIGHT)
    model.eval()

    ann, root = args.dataset
    dataset = COCODataset(ann, root, remove_images_without_annotations=False,
                          transforms=build_transforms(c, is_train=False))
    loader = make_test_data_loader(c, dataset, is_distributed=world > 1)
    folder = None
    if output_dir:
        folder = os.path.join(output_dir, "inference", os.path.splitext(os.path.basename(ann))[0])
        os.makedirs(folder, exist_ok=True)
    iou_types = ("bbox",)
    records = inference(model, loader, dataset_name=os.path.basename(ann), iou_types=iou_types,
                        box_only=c.MODEL.RPN_ONLY, device=device, expected_results=c.TEST.EXPECTED_RESULTS,
                        expected_results_sigma_tol=c.TEST.EXPECTED_RESULTS_SIGMA_TOL, output_folder=folder)
    synchronize()
    if records is not None:
        log.info("%d detections on %d images%s", len(records), len(dataset),
                 " -> %s" % os.path.join(folder, "bbox.json") if folder else "")


if __name__ == "__main__":
    main()
