"""Flattening helpers for the RPN prediction maps (reference: maskrcnn_benchmark/modeling/rpn/utils.py)."""
from ..utils import cat


def permute_and_flatten(layer, N, A, C, H, W):
    """[N, A*C, H, W] -> [N, H*W*A, C]; on channels_last maps the permute is a plain view"""
    layer = layer.reshape(N, -1, C, H, W)
    layer = layer.permute(0, 3, 4, 1, 2)
    return layer.reshape(N, -1, C)


def concat_box_prediction_layers(box_cls, box_regression, masks=None):
    cls_flat, reg_flat = [], []
    for cls_lvl, reg_lvl in zip(box_cls, box_regression):
        if masks is not None:
            cls_lvl, reg_lvl = cls_lvl[masks, :], reg_lvl[masks, :]
        N, AxC, H, W = cls_lvl.shape
        A = reg_lvl.shape[1] // 4
        C = AxC // A
        cls_flat.append(permute_and_flatten(cls_lvl, N, A, C, H, W))
        reg_flat.append(permute_and_flatten(reg_lvl, N, A, 4, H, W))
    return cat(cls_flat, dim=1).reshape(-1, C), cat(reg_flat, dim=1).reshape(-1, 4)
