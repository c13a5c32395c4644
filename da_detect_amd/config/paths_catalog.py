"""Dataset / pretrained-model catalog (reference: maskrcnn_benchmark/config/paths_catalog.py:7-300).

`cfg.PATHS_CATALOG` names a python file that defines `DatasetCatalog` (and `ModelCatalog`); data/build.py loads it ONCE
per file (utils.imports.load_paths_catalog: run-time `register` calls and edits of the tables stay visible) and calls `DatasetCatalog.get(name)` -> dict(factory=<class name in data.datasets>,
args=dict(root=..., ann_file=...)).  The reference's own file hard-codes its author's directories; a site keeps
pointing PATHS_CATALOG at its copy of that file (it is plain data and loads unchanged), or fills this one:

    DADET_DATA_DIR=/data python tools/train_net_da.py ...            # relative entries resolve under DATA_DIR
    DatasetCatalog.register("my_train_cocostyle", "my/images", "my/annotations.json")

Entries below are the dataset NAMES the shipped yamls refer to, laid out under DATA_DIR the way the Cityscapes /
BDD100K conversion scripts of the reference write them.
"""
import os


class DatasetCatalog(object):
    DATA_DIR = os.environ.get("DADET_DATA_DIR", "datasets")
    DATASETS = {
        "coco_2014_train": {"img_dir": "coco/train2014", "ann_file": "coco/annotations/instances_train2014.json"},
        "coco_2014_val": {"img_dir": "coco/val2014", "ann_file": "coco/annotations/instances_val2014.json"},
        "coco_2014_minival": {"img_dir": "coco/val2014", "ann_file": "coco/annotations/instances_minival2014.json"},
        "coco_2014_valminusminival": {"img_dir": "coco/val2014",
                                      "ann_file": "coco/annotations/instances_valminusminival2014.json"},
        "cityscapes_fine_instanceonly_seg_train_cocostyle": {
            "img_dir": "cityscapes/leftImg8bit/train",
            "ann_file": "cityscapes/annotations/instancesonly_filtered_gtFine_train.json"},
        "cityscapes_fine_instanceonly_seg_val_cocostyle": {
            "img_dir": "cityscapes/leftImg8bit/val",
            "ann_file": "cityscapes/annotations/instancesonly_filtered_gtFine_val.json"},
        "foggy_cityscapes_fine_instanceonly_seg_train_cocostyle": {
            "img_dir": "cityscapes/leftImg8bit_foggy/train",
            "ann_file": "cityscapes/annotations/instancesonly_filtered_gtFine_train.json"},
        "foggy_cityscapes_fine_instanceonly_seg_val_cocostyle": {
            "img_dir": "cityscapes/leftImg8bit_foggy/val",
            "ann_file": "cityscapes/annotations/instancesonly_filtered_gtFine_val.json"},
        "rainy_cityscapes_fine_instanceonly_seg_train_cocostyle": {
            "img_dir": "cityscapes/leftImg8bit_rain/train",
            "ann_file": "cityscapes/annotations/instancesonly_filtered_gtFine_train.json"},
        "rainy_cityscapes_fine_instanceonly_seg_val_cocostyle": {
            "img_dir": "cityscapes/leftImg8bit_rain/val",
            "ann_file": "cityscapes/annotations/instancesonly_filtered_gtFine_val.json"},
        "bdd100k_daytime_clear_city_street_train_cocostyle": {
            "img_dir": "bdd100k/daytime_clear_city_street_coco/train",
            "ann_file": "bdd100k/daytime_clear_city_street_coco/train_bdd100k_coco.json"},
        "rainy_bdd100k_daytime_clear_city_street_train_cocostyle": {
            "img_dir": "bdd100k/overcast_rainy/train",
            "ann_file": "bdd100k/daytime_clear_city_street_coco/train_bdd100k_coco.json"},
        "bdd100k_daytime_clear_city_street_val_cocostyle": {
            "img_dir": "bdd100k/daytime_clear_city_street_coco/val",
            "ann_file": "bdd100k/daytime_clear_city_street_coco/val_bdd100k_coco.json"},
    }

    @classmethod
    def register(cls, name, img_dir, ann_file):
        """add / replace a COCO-style entry (absolute paths are kept, relative ones resolve under DATA_DIR)"""
        cls.DATASETS[name] = {"img_dir": img_dir, "ann_file": ann_file}

    @staticmethod
    def get(name):
        if "coco" in name:      # every name above contains "coco" (".._cocostyle"), the reference's dispatch rule
            if name not in DatasetCatalog.DATASETS:
                raise RuntimeError("Dataset not available: {}".format(name))
            attrs = DatasetCatalog.DATASETS[name]
            data_dir = os.environ.get("DADET_DATA_DIR", DatasetCatalog.DATA_DIR)     # the environment wins, at call time
            return dict(factory="COCODataset", args=dict(root=os.path.join(data_dir, attrs["img_dir"]),
                                                         ann_file=os.path.join(data_dir, attrs["ann_file"])))
        raise RuntimeError("Dataset not available: {}".format(name))


class ModelCatalog(object):
    """`catalog://` weights are downloads in the reference (paths_catalog.py:254-300); there is no network here, so a
    catalog name resolves to a local file under MODEL_DIR (DADET_MODEL_DIR) with the reference's file names."""
    MODEL_DIR = os.environ.get("DADET_MODEL_DIR", "pretrained")
    C2_IMAGENET_MODELS = {"MSRA/R-50": "ImageNetPretrained/MSRA/R-50.pkl", "MSRA/R-101": "ImageNetPretrained/MSRA/R-101.pkl",
                          "FAIR/20171220/X-101-32x8d": "ImageNetPretrained/20171220/X-101-32x8d.pkl"}

    @staticmethod
    def get(name):
        if name.startswith("ImageNetPretrained/"):
            key = name[len("ImageNetPretrained/"):]
            if key in ModelCatalog.C2_IMAGENET_MODELS:
                return os.path.join(os.environ.get("DADET_MODEL_DIR", ModelCatalog.MODEL_DIR),
                                    ModelCatalog.C2_IMAGENET_MODELS[key])
        raise RuntimeError("model not present in the catalog {}".format(name))
