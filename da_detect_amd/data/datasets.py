"""COCO-style detection datasets with a domain flag (reference: maskrcnn_benchmark/data/datasets/coco.py:44-124 and
data/build.py:23-63 `Dataset_triplet`).  The annotation file is read with the json module (pycocotools /
torchvision.datasets.CocoDetection are not needed for bounding boxes); segmentation masks and keypoints are outside the
DA Faster R-CNN path."""
import copy
import json
import os

import numpy as np
import torch

from ..structures.bounding_box import BoxList

MIN_KEYPOINTS_PER_IMAGE = 10


def has_valid_annotation(anno):
    """coco.py:21-41 for box-only annotations: at least one box that is not (nearly) degenerate"""
    if len(anno) == 0:
        return False
    if all(any(o <= 1 for o in obj["bbox"][2:]) for obj in anno):
        return False
    return True


class COCODataset(torch.utils.data.Dataset):
    def __init__(self, ann_file, root, remove_images_without_annotations, transforms=None, is_source=True,
                 decode_to_tensor=False):
        with open(ann_file) as f:
            data = json.load(f)
        self.root = root
        self.imgs = {im["id"]: im for im in data["images"]}
        self.anns_of = {}
        for a in data.get("annotations", []):
            self.anns_of.setdefault(a["image_id"], []).append(a)
        self.ids = sorted(self.imgs.keys())
        if remove_images_without_annotations:
            self.ids = [i for i in self.ids if has_valid_annotation(self.anns_of.get(i, []))]
        cat_ids = sorted(c["id"] for c in data.get("categories", []))
        self.json_category_id_to_contiguous_id = {v: i + 1 for i, v in enumerate(cat_ids)}
        self.contiguous_category_id_to_json_id = {v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self.id_to_img_map = {k: v for k, v in enumerate(self.ids)}
        self._transforms = transforms
        self.is_source = is_source
        self.decode_to_tensor = decode_to_tensor     # True: return uint8 [H,W,3] for data/device_prep.py

    def __len__(self):
        return len(self.ids)

    def _load_image(self, img_id):
        from PIL import Image

        return Image.open(os.path.join(self.root, self.imgs[img_id]["file_name"])).convert("RGB")

    def __getitem__(self, idx):
        img_id = self.ids[idx]
        img = self._load_image(img_id)
        anno = [o for o in self.anns_of.get(img_id, []) if o.get("iscrowd", 0) == 0]
        boxes = torch.as_tensor([o["bbox"] for o in anno], dtype=torch.float32).reshape(-1, 4)
        target = BoxList(boxes, img.size, mode="xywh").convert("xyxy")
        classes = torch.tensor([self.json_category_id_to_contiguous_id[o["category_id"]] for o in anno],
                               dtype=torch.int64)
        target.add_field("labels", classes)
        target.add_field("is_source", torch.full_like(classes, bool(self.is_source), dtype=torch.bool))
        target = target.clip_to_image(remove_empty=True)
        if self.decode_to_tensor:
            return torch.from_numpy(np.array(img, dtype=np.uint8)), target, idx
        if self._transforms is not None:
            img, target = self._transforms(img, target)
        return img, target, idx

    def get_img_info(self, index):
        return self.imgs[self.id_to_img_map[index]]


class ConcatDataset(torch.utils.data.ConcatDataset):
    """concatenation that still answers get_img_info (reference: data/datasets/concat_dataset.py:7-23)"""

    def get_img_info(self, idx):
        import bisect

        which = bisect.bisect_right(self.cumulative_sizes, idx)
        return self.datasets[which].get_img_info(idx if which == 0 else idx - self.cumulative_sizes[which - 1])


class TripletDataset(torch.utils.data.Dataset):
    """index-aligned (source, target, auxiliary) samples; the target / auxiliary images are paired with a COPY of the
    source annotations carrying their own domain flag — build.py:34-46 (the datasets are renderings of the same
    scenes: Cityscapes, its foggy and its rainy version)"""

    def __init__(self, datasets):
        assert len(datasets) == 3
        self.dataset_s, self.dataset_p, self.dataset_n = datasets

    def __len__(self):
        return len(self.dataset_s)

    @staticmethod
    def _with_domain(target_s, target_other):
        t = copy.deepcopy(target_s)
        t.add_field("is_source", copy.deepcopy(target_other.get_field("is_source")))
        return t

    def __getitem__(self, index):
        img_s, target_s, i1 = self.dataset_s[index]
        img_p, target_p, i2 = self.dataset_p[index]
        img_n, target_n, i3 = self.dataset_n[index]
        return (img_s, target_s, img_p, self._with_domain(target_s, target_p), img_n,
                self._with_domain(target_s, target_n), i1, i2, i3)

    def get_img_info(self, index):
        return self.dataset_s.get_img_info(index)


Dataset_triplet = TripletDataset      # the reference's name (data/build.py:23)
