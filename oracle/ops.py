"""ctypes + numpy front-end of the CPU oracle (oracle/dadet_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by da_detect_amd/.  All functions take and return numpy arrays in the REFERENCE's layouts (NCHW).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdadet_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_nms.restype = ctypes.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def nms(boxes, scores, thresh, tie_rule=0):
    """kept original indices, ascending (reference: csrc/cpu/nms_cpu.cpp:6-65)"""
    boxes, scores = _f(boxes).reshape(-1, 4), _f(scores).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = lib().oracle_nms(_fp(boxes), _fp(scores), ctypes.c_int(n), ctypes.c_float(thresh),
                         ctypes.c_int(tie_rule), _fp(keep))
    return keep[:k].copy()


def roi_align_forward(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    inp, rois = _f(inp), _f(rois).reshape(-1, 5)
    B, C, H, W = inp.shape
    R = rois.shape[0]
    out = np.empty((R, C, ph, pw), dtype=np.float32)
    lib().oracle_roi_align_forward(_fp(inp), _fp(rois), _fp(out), B, C, H, W, R, ph, pw,
                                   ctypes.c_float(spatial_scale), sampling_ratio)
    return out


def roi_align_backward(grad, rois, spatial_scale, ph, pw, B, C, H, W, sampling_ratio):
    grad, rois = _f(grad), _f(rois).reshape(-1, 5)
    R = rois.shape[0]
    gin = np.empty((B, C, H, W), dtype=np.float32)
    lib().oracle_roi_align_backward(_fp(grad), _fp(rois), _fp(gin), B, C, H, W, R, ph, pw,
                                    ctypes.c_float(spatial_scale), sampling_ratio)
    return gin


def sigmoid_focal_loss_forward(logits, targets, gamma, alpha):
    logits = _f(logits)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    N, C = logits.shape
    out = np.empty_like(logits)
    lib().oracle_sigmoid_focal_loss_forward(_fp(logits), _fp(targets), _fp(out), N, C,
                                            ctypes.c_float(gamma), ctypes.c_float(alpha))
    return out


def sigmoid_focal_loss_backward(logits, targets, d_losses, gamma, alpha):
    logits, d_losses = _f(logits), _f(d_losses)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    N, C = logits.shape
    out = np.empty_like(logits)
    lib().oracle_sigmoid_focal_loss_backward(_fp(logits), _fp(targets), _fp(d_losses), _fp(out), N, C,
                                             ctypes.c_float(gamma), ctypes.c_float(alpha))
    return out


def decode_clip(deltas, anchors, weights, xform_clip, im_w, im_h):
    """BoxCoder.decode + clip_to_image of K (delta, anchor) pairs (box_coder.py:52-95, bounding_box.py:214-224)"""
    deltas, anchors = _f(deltas).reshape(-1, 4), _f(anchors).reshape(-1, 4)
    K = deltas.shape[0]
    out = np.empty((K, 4), dtype=np.float32)
    wx, wy, ww, wh = weights
    lib().oracle_decode_clip(_fp(deltas), _fp(anchors), K, ctypes.c_float(wx), ctypes.c_float(wy),
                             ctypes.c_float(ww), ctypes.c_float(wh), ctypes.c_float(xform_clip),
                             ctypes.c_float(im_w), ctypes.c_float(im_h), _fp(out))
    return out


def roi_pool_forward(inp, rois, spatial_scale, ph, pw):
    inp, rois = _f(inp), _f(rois).reshape(-1, 5)
    B, C, H, W = inp.shape
    R = rois.shape[0]
    out = np.empty((R, C, ph, pw), dtype=np.float32)
    arg = np.empty((R, C, ph, pw), dtype=np.int32)
    lib().oracle_roi_pool_forward(_fp(inp), _fp(rois), _fp(out), _fp(arg), B, C, H, W, R, ph, pw,
                                  ctypes.c_float(spatial_scale))
    return out, arg


def roi_pool_backward(grad, argmax, rois, B, C, H, W):
    grad, rois = _f(grad), _f(rois).reshape(-1, 5)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    R, _, ph, pw = grad.shape
    gin = np.empty((B, C, H, W), dtype=np.float32)
    lib().oracle_roi_pool_backward(_fp(grad), _fp(argmax), _fp(rois), _fp(gin), B, C, H, W, R, ph, pw)
    return gin
