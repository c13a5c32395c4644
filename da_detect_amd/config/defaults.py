"""Default configuration tree.

Key names and default values are the reference's (reference: maskrcnn_benchmark/config/defaults.py:21-430,
DA knobs :259-283) so that every `configs/da_faster_rcnn/*.yaml` and `configs/e2e_faster_rcnn_R_50_C4_1x.yaml`
of the reference merges unchanged.  Expressed as one nested literal instead of attribute assignments.
"""
import os

from .cfgnode import CfgNode

_POOL16 = (1.0 / 16,)

_DEFAULTS = {
    "MODEL": {
        "RPN_ONLY": False, "MASK_ON": False, "DOMAIN_ADAPTATION_ON": False, "RETINANET_ON": False,
        "KEYPOINT_ON": False, "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN",
        "CLS_AGNOSTIC_BBOX_REG": False, "WEIGHT": "",
        "EVAL_USE_IN_TRAINING": True, "OUTPUT_DIR": "./", "SAVE_DIR": "./", "OUTPUT_SAVE_NAME": "output",
        "BACKBONE": {"CONV_BODY": "R-50-C4", "FREEZE_CONV_BODY_AT": 2, "OUT_CHANNELS": 256 * 4, "USE_GN": False},
        "FPN": {"USE_GN": False, "USE_RELU": False},
        "GROUP_NORM": {"DIM_PER_GP": -1, "NUM_GROUPS": 32, "EPSILON": 1e-5},
        "RPN": {
            "USE_FPN": False, "ANCHOR_SIZES": (32, 64, 128, 256, 512), "ANCHOR_STRIDE": (16,),
            "ASPECT_RATIOS": (0.5, 1.0, 2.0), "STRADDLE_THRESH": 0, "FG_IOU_THRESHOLD": 0.7,
            "BG_IOU_THRESHOLD": 0.3, "BATCH_SIZE_PER_IMAGE": 256, "POSITIVE_FRACTION": 0.5,
            "PRE_NMS_TOP_N_TRAIN": 12000, "PRE_NMS_TOP_N_TEST": 6000, "POST_NMS_TOP_N_TRAIN": 2000,
            "POST_NMS_TOP_N_TEST": 1000, "NMS_THRESH": 0.7, "MIN_SIZE": 0,
            "FPN_POST_NMS_TOP_N_TRAIN": 2000, "FPN_POST_NMS_TOP_N_TEST": 2000, "RPN_HEAD": "SingleConvRPNHead",
        },
        "ROI_HEADS": {
            "USE_FPN": False, "FG_IOU_THRESHOLD": 0.5, "BG_IOU_THRESHOLD": 0.5,
            "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25,
            "SCORE_THRESH": 0.05, "NMS": 0.5, "DETECTIONS_PER_IMG": 100,
        },
        "ROI_BOX_HEAD": {
            "FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor", "PREDICTOR": "FastRCNNPredictor",
            "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_SCALES": _POOL16, "NUM_CLASSES": 81,
            "MLP_HEAD_DIM": 1024, "USE_GN": False, "DILATION": 1, "CONV_HEAD_DIM": 256, "NUM_STACKED_CONVS": 4,
        },
        "ROI_MASK_HEAD": {
            "FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor", "PREDICTOR": "MaskRCNNC4Predictor",
            "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_SCALES": _POOL16, "MLP_HEAD_DIM": 1024,
            "CONV_LAYERS": (256, 256, 256, 256), "RESOLUTION": 14, "SHARE_BOX_FEATURE_EXTRACTOR": True,
            "POSTPROCESS_MASKS": False, "POSTPROCESS_MASKS_THRESHOLD": 0.5, "DILATION": 1, "USE_GN": False,
        },
        "ROI_KEYPOINT_HEAD": {
            "FEATURE_EXTRACTOR": "KeypointRCNNFeatureExtractor", "PREDICTOR": "KeypointRCNNPredictor",
            "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_SCALES": _POOL16, "MLP_HEAD_DIM": 1024,
            "CONV_LAYERS": tuple(512 for _ in range(8)), "RESOLUTION": 14, "NUM_CLASSES": 17,
            "SHARE_BOX_FEATURE_EXTRACTOR": True,
        },
        "DA_HEADS": {
            "DA_IMG_GRL_WEIGHT": 0.1, "DA_INS_GRL_WEIGHT": 0.1, "DA_IMG_LOSS_WEIGHT": 1.0,
            "DA_INS_LOSS_WEIGHT": 1.0, "DA_CST_LOSS_WEIGHT": 0.1, "DA_TRIPLET_INS_WEIGHT": 1.0,
            "DA_TRIPLET_IMG_WEIGHT": 1.0, "DA_ADV_GRL": True, "DA_ADV_GRL_THRESHOLD": 30, "ALIGNMENT": True,
            "TRIPLET_USE": True, "TRIPLET_MARGIN": 1.0, "TRIPLET_MAX_MARGIN": 1.0, "TRIPLET_MARGIN_INS": 1.0,
            "TRIPLET_MARGIN_IMG": 1.0, "DA_IMG_advGRL_WEIGHT": 0.1, "DA_INS_advGRL_WEIGHT": 0.1,
        },
        "RESNETS": {
            "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True,
            "TRANS_FUNC": "BottleneckWithFixedBatchNorm", "STEM_FUNC": "StemWithFixedBatchNorm",
            "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64,
            # vendored tree (tools/cityscapes/maskrcnn_benchmark/config/defaults.py:287-289)
            "STAGE_WITH_DCN": (False, False, False, False), "WITH_MODULATED_DCN": False, "DEFORMABLE_GROUPS": 1,
        },
        "RETINANET": {
            "NUM_CLASSES": 81, "ANCHOR_SIZES": (32, 64, 128, 256, 512), "ASPECT_RATIOS": (0.5, 1.0, 2.0),
            "ANCHOR_STRIDES": (8, 16, 32, 64, 128), "STRADDLE_THRESH": 0, "OCTAVE": 2.0, "SCALES_PER_OCTAVE": 3,
            "USE_C5": True, "NUM_CONVS": 4, "BBOX_REG_WEIGHT": 4.0, "BBOX_REG_BETA": 0.11, "PRE_NMS_TOP_N": 1000,
            "FG_IOU_THRESHOLD": 0.5, "BG_IOU_THRESHOLD": 0.4, "LOSS_ALPHA": 0.25, "LOSS_GAMMA": 2.0,
            "PRIOR_PROB": 0.01, "INFERENCE_TH": 0.05, "NMS_TH": 0.4,
        },
    },
    "INPUT": {
        "MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333,
        "PIXEL_MEAN": [102.9801, 115.9465, 122.7717], "PIXEL_STD": [1.0, 1.0, 1.0], "TO_BGR255": True,
    },
    "DATASETS": {"TRAIN": (), "SOURCE_TRAIN": (), "TARGET_TRAIN": (), "TARGET_TRAIN_negative": (), "TEST": (),
                 "TEST_SOURCE": ()},
    "DATALOADER": {"NUM_WORKERS": 4, "SIZE_DIVISIBILITY": 0, "ASPECT_RATIO_GROUPING": True},
    "SOLVER": {
        "MAX_ITER": 40000, "BASE_LR": 0.0001, "BIAS_LR_FACTOR": 2, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005,
        "WEIGHT_DECAY_BIAS": 0, "GAMMA": 0.1, "STEPS": (30000,), "WARMUP_FACTOR": 1.0 / 3, "WARMUP_ITERS": 500,
        "WARMUP_METHOD": "linear", "WARMUP_LR": 0.0001, "LR_MIN": 0.000001, "CHECKPOINT_PERIOD": 2500,
        "IMS_PER_BATCH": 16,
    },
    "TEST": {"EXPECTED_RESULTS": [], "EXPECTED_RESULTS_SIGMA_TOL": 4, "IMS_PER_BATCH": 8, "DETECTIONS_PER_IMG": 100},
    "TENSORBOARD_EXPERIMENT": "logs/maskrcnn-benchmark",
    "PATHS_CATALOG": os.path.join(os.path.dirname(__file__), "paths_catalog.py"),
}

_C = CfgNode(_DEFAULTS)


def get_cfg_defaults():
    """a fresh copy of the default tree"""
    return _C.clone()
