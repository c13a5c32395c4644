"""Evaluation loop and the detection-record boundary (reference: maskrcnn_benchmark/engine/inference.py:18-129,
data/datasets/evaluation/coco/coco_eval.py:81-112).

The model's eval forward runs on the HIP kernels; everything after it is host-side bookkeeping.  COCO mAP itself is
pycocotools arithmetic (parity unpinned, third party) — this module stops at the `bbox.json` records,
which is the boundary the reference hands to pycocotools.  A dataset only needs `id_to_img_map`, `get_img_info(i)`
and `contiguous_category_id_to_json_id` (the three members prepare_for_coco_detection touches)."""
import datetime
import json
import logging
import os
import time

import torch

from ..utils.comm import all_gather, get_world_size, is_main_process, synchronize


def compute_on_dataset(model, data_loader, device):
    """engine/inference.py:18-52: batches are (images, targets, image_ids) -> {image_id: BoxList on CPU}"""
    model.eval()
    results = {}
    cpu = torch.device("cpu")
    for batch in data_loader:
        images, _, image_ids = batch
        images = images.to(device)
        with torch.no_grad():
            output = model(images)
        results.update({i: o.to(cpu) for i, o in zip(image_ids, output)})
    return results


def _accumulate_predictions_from_multiple_gpus(predictions_per_gpu):
    """engine/inference.py:55-74"""
    all_predictions = all_gather(predictions_per_gpu)
    if not is_main_process():
        return None
    predictions = {}
    for p in all_predictions:
        predictions.update(p)
    image_ids = sorted(predictions.keys())
    if len(image_ids) != image_ids[-1] + 1:
        logging.getLogger("maskrcnn_benchmark.inference").warning(
            "Number of images that were gathered from multiple processes is not a contiguous set. "
            "Some images might be missing from the evaluation")
    return [predictions[i] for i in image_ids]


def prepare_for_coco_detection(predictions, dataset):
    """coco_eval.py:81-112: one record per detection, box resized to the original image and given as xywh"""
    records = []
    for image_id, prediction in enumerate(predictions):
        original_id = dataset.id_to_img_map[image_id]
        if len(prediction) == 0:
            continue
        info = dataset.get_img_info(image_id)
        prediction = prediction.resize((info["width"], info["height"])).convert("xywh")
        boxes = prediction.bbox.tolist()
        scores = prediction.get_field("scores").tolist()
        labels = [dataset.contiguous_category_id_to_json_id[i] for i in prediction.get_field("labels").tolist()]
        records.extend({"image_id": original_id, "category_id": labels[k], "bbox": box, "score": scores[k]}
                       for k, box in enumerate(boxes))
    return records


def inference(model, data_loader, dataset_name, iou_types=("bbox",), box_only=False, device="cuda",
              expected_results=(), expected_results_sigma_tol=4, output_folder=None, evaluate=None):
    """engine/inference.py:76-129.  `evaluate(dataset, predictions, output_folder, **extra)` is the dataset-specific
    scorer (pycocotools in the reference); when None the bbox records are written / returned instead."""
    device = torch.device(device)
    num_devices = get_world_size()
    logger = logging.getLogger("maskrcnn_benchmark.inference")
    dataset = data_loader.dataset
    logger.info("Start evaluation on %s dataset(%d images).", dataset_name, len(dataset))
    start = time.time()
    predictions = compute_on_dataset(model, data_loader, device)
    synchronize()
    total = time.time() - start
    logger.info("Total inference time: %s (%s s / img per device, on %d devices)",
                str(datetime.timedelta(seconds=total)), total * num_devices / max(len(dataset), 1), num_devices)
    predictions = _accumulate_predictions_from_multiple_gpus(predictions)
    if not is_main_process():
        return None
    if output_folder:
        torch.save(predictions, os.path.join(output_folder, "predictions.pth"))
    if evaluate is not None:
        return evaluate(dataset=dataset, predictions=predictions, output_folder=output_folder, box_only=box_only,
                        iou_types=iou_types, expected_results=expected_results,
                        expected_results_sigma_tol=expected_results_sigma_tol)
    records = prepare_for_coco_detection(predictions, dataset)
    if output_folder:
        with open(os.path.join(output_folder, "bbox.json"), "w") as f:
            json.dump(records, f)
    return records
