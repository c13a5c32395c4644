"""ctypes binding of libdadet_hip.so (C-ABI in include/dadet.h).

The library is the product: there is no CPU / eager fallback.  Importing this module never fails (so the
host-side logic can be unit-tested without a GPU), but the first operator call raises loudly when the
shared object has not been built or no HIP device is visible.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DADET_LIB: another build of the same library (A/B runs of build-time switches on one box)
LIB_PATH = os.environ.get("DADET_LIB") or os.path.join(_HERE, "libdadet_hip.so")
_lib = None
DEFAULT_GEMM_MODE = "4"


class ConvDesc(ctypes.Structure):
    """mirror of dadet_conv_desc (include/dadet.h)"""

    _fields_ = [(n, c_int) for n in (
        "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "Ho", "Wo", "OutH", "OutW",
        "out_spatial_stride", "relu_mode")]


class WgradPending(ctypes.Structure):
    """mirror of dadet_wgrad_pending (include/dadet.h)"""

    _fields_ = [("partials", c_void_p), ("out_scale", c_void_p), ("dw", c_void_p), ("count", ctypes.c_longlong),
                ("K", c_int), ("splits", c_int), ("accumulate", c_int)]


class TransposeItem(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("scale", c_void_p), ("wt", c_void_p), ("Cout", c_int), ("KH", c_int), ("KW", c_int),
                ("Cin", c_int), ("first_block", c_int), ("blocks_ci", c_int), ("blocks_co", c_int), ("cout_pad", c_int)]


class AmaxItem(ctypes.Structure):
    """mirror of dadet_amax_item (include/dadet.h)"""

    _fields_ = [("x", c_void_p), ("slot", c_void_p), ("n", ctypes.c_longlong), ("first_block", c_int), ("blocks", c_int)]


class MergeEntry(ctypes.Structure):
    """mirror of dadet_merge_entry (include/dadet.h)"""

    _fields_ = [("boxes", c_void_p), ("scores", c_void_p), ("keep", c_void_p), ("count", c_void_p), ("boxes_out", c_void_p),
                ("scores_out", c_void_p), ("n", c_int), ("cap", c_int)]


class TopkRow(ctypes.Structure):
    """mirror of dadet_topk_row (include/dadet.h)"""

    _fields_ = [("scores", c_void_p), ("out_scores", c_void_p), ("out_idx", c_void_p), ("n", c_int), ("k", c_int)]


class SgdEntry(ctypes.Structure):
    """mirror of dadet_sgd_entry (include/dadet.h)"""

    _fields_ = [("p", c_void_p), ("g", c_void_p), ("buf", c_void_p), ("numel", c_int64),
                ("lr", c_float), ("weight_decay", c_float)]


# name -> argtypes ; every function returns int (status) unless listed in _RESTYPES
_P = c_void_p
_SIGNATURES = {
    "dadet_version": [],
    "dadet_device_info": [POINTER(c_int), POINTER(c_int), POINTER(c_size_t), c_char_p, c_int],
    "dadet_nms_workspace_bytes": [c_int, POINTER(c_size_t)],
    "dadet_nms": [_P, _P, c_int, c_float, c_int, c_int, _P, c_size_t, _P, _P, _P],
    "dadet_roi_align_forward": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P],
    "dadet_roi_align_backward": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P],
    "dadet_roi_align_workspace_bytes": [c_int, c_int, c_int, c_int, POINTER(c_size_t)],
    "dadet_roi_align_forward_ws": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_size_t, _P],
    "dadet_roi_align_backward_atomic": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P],
    "dadet_roi_align_forward_sub": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                    _P, c_size_t, _P],
    "dadet_roi_align_forward_level": [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P,
                                      c_size_t, _P],
    "dadet_roi_align_backward_level": [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                       _P],
    "dadet_roi_align_backward_sub": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                     _P],
    "dadet_sigmoid_focal_loss_forward": [_P, _P, _P, c_int, c_int, c_float, c_float, _P],
    "dadet_sigmoid_focal_loss_backward": [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P],
    "dadet_conv_forward": [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P],
    "dadet_conv_forward_scaled": [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "dadet_conv_wgrad_scaled": [POINTER(ConvDesc), _P, _P, c_int, _P, _P, c_int, _P, c_size_t, POINTER(WgradPending), _P, _P,
                                _P],
    "dadet_amax": [_P, ctypes.c_longlong, _P, _P],
    "dadet_amax_batch": [_P, c_int, c_int, _P],
    "dadet_conv_forward_variant": [POINTER(ConvDesc)],
    "dadet_set_gemm_mode": [c_int],
    "dadet_get_gemm_mode": [],
    "dadet_conv_wgrad_variant": [POINTER(ConvDesc)],
    "dadet_conv_wgrad_group_plan": [POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_size_t)],
    "dadet_conv_wgrad_group": [POINTER(ConvDesc), c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                               POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_size_t),
                               POINTER(WgradPending), POINTER(c_void_p), POINTER(c_void_p), _P],
    "dadet_nonfinite_poll": [c_char_p, c_int],
    "dadet_set_big_gemm": [c_int],
    "dadet_get_big_gemm": [],
    "dadet_conv_wgrad_workspace_bytes": [POINTER(ConvDesc), POINTER(c_size_t)],
    "dadet_conv_wgrad": [POINTER(ConvDesc), _P, _P, _P, _P, c_int, _P, c_size_t, _P],
    "dadet_conv_wgrad_partials": [POINTER(ConvDesc), _P, _P, _P, _P, c_int, _P, c_size_t, POINTER(WgradPending), _P],
    "dadet_conv_wgrad_reduce_batch": [POINTER(WgradPending), c_int, _P],
    "dadet_conv_weight_transpose": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "dadet_conv_weight_transpose_padded": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "dadet_conv_wgrad_partials_ld": [POINTER(ConvDesc), _P, _P, c_int, _P, _P, c_int, _P, c_size_t, POINTER(WgradPending),
                                     _P],
    "dadet_conv_weight_transpose_batch": [_P, c_int, c_int, _P],
    "dadet_deform_sample_forward": [_P, _P, _P, _P] + [c_int] * 12 + [_P],
    "dadet_deform_sample_backward": [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 12 + [_P],
    "dadet_deform_sample_forward_ld": [_P, _P, c_int, _P, c_int, c_int, _P] + [c_int] * 12 + [_P],
    "dadet_deform_sample_backward_ld": [_P, _P, c_int, _P, c_int, c_int, _P, _P, _P, c_int, _P, c_int] + [c_int] * 12
                                       + [_P, c_size_t, _P],
    "dadet_deform_sample_backward_ld_m": [_P, _P, c_int, _P, c_int, c_int, _P, _P, _P, c_int, _P, c_int] + [c_int] * 12
                                         + [_P, c_size_t, _P, ctypes.c_longlong, _P],
    "dadet_deform_sample_backward_workspace_bytes": [c_int, c_int, c_int, c_int, POINTER(c_size_t)],
    "dadet_image_resample_h": [_P, c_int, c_int, _P, _P, c_int, c_int, _P, _P],
    "dadet_image_resample_v_normalize": [_P, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, POINTER(c_float),
                                         POINTER(c_float), _P, c_int, _P],
    "dadet_rpn_loss": [_P, _P, _P, _P, c_int, _P, _P, c_int, c_float, _P, _P, _P, _P],
    "dadet_fpn_merge_levels": [POINTER(MergeEntry), c_int, _P],
    "dadet_rpn_loss_rows": [_P, _P, _P, _P, c_int, c_int, _P, c_int, c_float, _P, _P, c_int, _P, _P],
    "dadet_rpn_loss_rows_level": [_P, _P, _P, _P, c_int, c_int, _P, c_int, c_float, ctypes.c_longlong, ctypes.c_longlong,
                                  ctypes.c_longlong, c_int, _P, c_int, _P, _P, c_int, _P, _P],
    "dadet_gather_pixel_taps_level": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "dadet_scatter_pixel_taps_add_level": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                           _P],
    "dadet_gather_pixel_taps": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "dadet_scatter_pixel_taps_add": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "dadet_fast_rcnn_loss": [_P, _P, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_int, _P, _P, _P, _P],
    "dadet_fast_rcnn_loss_rows": [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P],
    "dadet_sample_rois": [_P, _P, _P, c_int, c_int, c_int, c_uint64, c_int, _P, _P, _P, _P, _P, _P, _P, _P],
    "dadet_proposals_sample": [_P, _P, _P, _P, c_int, _P, c_int, _P, _P, c_int, c_float, c_float, c_float, c_float, c_float,
                               c_float, c_int, c_int, c_uint64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "dadet_sample_anchors": [_P, _P, c_int, c_int, c_int, c_uint64, c_int64, _P, _P, _P, _P, _P],
    "dadet_topk_sorted": [_P, c_int, c_int, c_int64, c_int, _P, _P, _P],
    "dadet_topk_sorted_rows_workspace_bytes": [c_int, c_int, POINTER(c_size_t)],
    "dadet_topk_sorted_rows": [POINTER(TopkRow), c_int, _P, c_size_t, _P],
    "dadet_rpn_anchor_targets": [_P, _P, c_int, _P, c_int, c_float, c_float, _P, _P, _P, _P],
    "dadet_box_match_encode": [_P, c_int, _P, _P, c_int, c_float, c_float, c_float, c_float, c_float, c_float, _P, _P,
                               _P, _P],
    "dadet_roi_pool_forward": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    "dadet_roi_pool_backward": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "dadet_deform_psroi_pool_forward": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                        c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P],
    "dadet_deform_psroi_pool_backward": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_float, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P],
    "dadet_relu_bn_backward": [_P, _P, _P, _P, _P, c_int64, c_int, _P],
    "dadet_relu_bn_backward_m": [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P],
    "dadet_colsum_workspace_bytes": [c_int64, c_int, POINTER(c_size_t)],
    "dadet_colsum": [_P, _P, c_int64, c_int, _P, c_size_t, _P],
    "dadet_colsum_ld": [_P, c_int, _P, c_int64, c_int, c_int, _P, c_size_t, _P],
    "dadet_channel_affine": [_P, _P, _P, _P, c_int64, c_int, c_int, _P],
    "dadet_maxpool3x3s2_forward": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "dadet_avgpool_forward": [_P, _P, c_int, c_int, c_int, _P],
    "dadet_avgpool_backward": [_P, _P, c_int, c_int, c_int, _P],
    "dadet_nchw3_to_nhwc4": [_P, _P, c_int, c_int, c_int, _P],
    "dadet_rpn_decode_clip": [_P, _P, _P, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_float, _P, _P],
    "dadet_da_img_head_loss_forward": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "dadet_da_img_head_loss_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "dadet_da_img_head_loss_backward_g": [_P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, c_int, c_int, c_int,
                                          _P],
    "dadet_da_img_head_loss_backward_gm": [_P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, c_int, c_int, c_int,
                                          _P, _P, _P],
    "dadet_da_ins_tail_forward": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "dadet_da_ins_tail_backward": [_P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "dadet_da_ins_dropout_rows": [_P, _P, _P, c_int64, c_int, _P],
    "dadet_da_ins_merge": [_P, _P, _P, _P, _P, _P, c_int64, c_int, _P],
    "dadet_triplet_w_forward": [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P],
    "dadet_triplet_w_backward": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P],
    "dadet_sgd_step": [_P, c_int, c_int64, c_float, c_int, c_float, _P],
}
_RESTYPES = {"dadet_last_error": c_char_p}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["dadet_last_error"])


class DadetError(RuntimeError):
    pass


def load():
    """dlopen the library and attach prototypes; raises DadetError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DadetError(
            "libdadet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C da_detect_amd/csrc`. There is no CPU fallback for the product path." % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64, and libdadet_hip.so must bind to THAT copy (same SONAME, already
    # loaded).  Loaded the other way round the process ends up with two HIP runtimes (/opt/rocm's for this library,
    # torch's for torch) and the second one to initialise reports "no ROCm-capable device is detected" — seen when
    # __graft_entry__.build() (which loads the library without touching torch) and smoke() ran in one process.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.dadet_last_error.argtypes = []
    lib.dadet_last_error.restype = c_char_p
    _lib = lib
    # contraction mode of the GEMM kernels: DADET_GEMM_MODE = 4 (default: fp32-class two-term fp16 split under per-tensor
    # scales), 3 (fp32-class three-term bf16 split), 0 (exact fp32 MFMA) or 2 (two bf16 terms, ~2^-16 products); see
    # include/dadet.h
    mode = int(os.environ.get("DADET_GEMM_MODE", DEFAULT_GEMM_MODE))
    if lib.dadet_set_gemm_mode(mode) != 0:
        raise DadetError("DADET_GEMM_MODE=%d is not a valid contraction mode (0, 2, 3 or 4)" % mode)
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dadet_last_error()
        raise DadetError("%s failed (status %d): %s" % (what or "dadet call", rc, (msg or b"").decode()))


def call(name, *args):
    check(getattr(load(), name)(*args), name)
