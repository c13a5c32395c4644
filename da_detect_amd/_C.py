"""Tensor-level entry points over the C-ABI library — the stand-in for `maskrcnn_benchmark._C`.

The first block mirrors the reference's pybind surface one-to-one (reference:
maskrcnn_benchmark/csrc/vision.cpp:7-15): `nms`, `roi_align_forward`, `roi_align_backward`,
`sigmoid_focalloss_forward`, `sigmoid_focalloss_backward` with the same argument order and meaning.
The second block exposes the kernels the reference gets from ATen (conv / pooling / SGD) and the fused DA
head tails.  PyTorch is used here only for device memory and the current HIP stream.

Layout contract: 4-D activations are logical NCHW tensors in torch.channels_last memory format (physical
NHWC), weights are logical [Cout,Cin,KH,KW] in channels_last (physical [Cout,KH,KW,Cin]).  Inputs in any
other layout are converted (one copy); outputs are always channels_last.
"""
import contextlib
import os
import ctypes
import time

import torch

from . import _lib
from . import amax as _amax
from ._lib import ConvDesc

CL = torch.channels_last

# NMS tie rule used by default: the reference's CPU build suppresses on IoU >= thr, its CUDA build on
# IoU > thr (csrc/cpu/nms_cpu.cpp:60 vs csrc/cuda/nms.cu:60).  The oracle available without CUDA is the CPU
# rule, so that is the default; set to 1 to reproduce the CUDA build.
NMS_TIE_RULE = 0

# optional per-launch timing of the GEMM kernels (bench.py sets this to a KernelProfiler; None = off)
PROFILER = None
_KNAME_CACHE = {}   # (M, Cout, K) -> kernel name of the forward GEMM variant (profiling only)


class KernelProfiler(object):
    """HIP-event bracket around kernel launches on the CURRENT stream (the stream the kernels are launched
    on); elapsed times are read after a device synchronise, never inside the timed region."""

    def __init__(self, pool=0, only=None):
        self.only = only    # optional substring: bracket only the kernels whose name contains it
        # events are created (and once recorded, which is what instantiates the HIP object) up front so that the
        # timed region only pays for hipEventRecord
        self._pool = []
        for _ in range(pool):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._pool.append(ev)
        self.records = {}   # name -> list of (start_event, end_event, algorithmic_flops)
        self.bytes = {}     # name -> summed algorithmic bytes (operands read once + result written once)
        self.origin = torch.cuda.Event(enable_timing=True)   # common time origin for the union of busy intervals
        torch.cuda.synchronize()
        self.origin.record()
        self.host_origin = time.perf_counter()   # the same instant on the host clock (idle device: recorded at once)
        self.host_times = {}                     # name -> host time (ms since origin) of every bracketed launch

    class _Span(object):
        def __init__(self, prof, name, work):
            self.prof, self.name, self.work = prof, name, work

        def __enter__(self):
            if getattr(self.prof, "detail", False):
                self.prof.host_times.setdefault(self.name, []).append(
                    (time.perf_counter() - self.prof.host_origin) * 1e3)
            pool = self.prof._pool
            self.s = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
            self.e = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
            self.s.record()

        def __exit__(self, *exc):
            self.e.record()
            self.prof.records.setdefault(self.name, []).append((self.s, self.e, self.work))
            self.prof.bytes[self.name] = self.prof.bytes.get(self.name, 0.0) + self.nbytes

    def span(self, name, work, nbytes=0.0):
        if self.only is not None and self.only not in name:
            return contextlib.nullcontext()
        sp = KernelProfiler._Span(self, name, work)
        sp.nbytes = nbytes
        return sp

    def summary(self):
        """name -> dict(launches, total_ms, avg_ms, work_per_launch, achieved = work / time per second)"""
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            work = sum(w for _, _, w in recs)
            out[name] = dict(launches=len(recs), total_ms=ms, avg_ms=ms / len(recs),
                             bytes_per_launch=self.bytes.get(name, 0.0) / len(recs),
                             work_per_launch=work / len(recs), achieved=work / (ms * 1e-3) if ms > 0 else 0.0)
        return out

    def union(self):
        """(total work, milliseconds during which AT LEAST ONE bracketed kernel was running): kernels of the
        data-gradient and weight-gradient streams overlap, so summed durations count that time twice"""
        torch.cuda.synchronize()
        spans, work = [], 0.0
        for recs in self.records.values():
            for s, e, w in recs:
                spans.append((self.origin.elapsed_time(s), self.origin.elapsed_time(e)))
                work += w
        spans.sort()
        busy, cur_s, cur_e = 0.0, None, None
        for s, e in spans:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        return work, busy


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.DadetError(
            "%s must be a tensor on the HIP device: the da_detect_amd operators have no CPU path" % name)
    if t.dtype != torch.float32:
        raise _lib.DadetError("%s must be float32, got %s" % (name, t.dtype))
    return t


def _nhwc(t):
    """physical NHWC view of a logical NCHW tensor (copies only when the layout differs)"""
    return t.contiguous(memory_format=CL)


_MODE = None   # cached contraction mode (set_gemm_mode keeps it current)


def _mode4():
    """True when the GEMMs run the two-term fp16 split, whose kernels want the operands' largest magnitudes (amax.py)"""
    global _MODE
    if _MODE is None:
        _MODE = _lib.load().dadet_get_gemm_mode()
    return _MODE == 4


def _weight_amax(w):
    """slot of the B operand of a GEMM: parameters (and views of them) and the cached transposed weights live in the
    persistent per-epoch table, anything else is treated like an activation"""
    like = w.__dict__.get("_dadet_amax_like")     # a padded copy of a parameter (zero rows appended): the parameter's maximum
    if like is not None and like() is not None:
        w = like()
    base = w._base if w._base is not None else w
    if isinstance(base, torch.nn.Parameter) or w.__dict__.get("_dadet_persistent"):
        return _amax.WEIGHTS.ptr(w, _TRANSPOSES.epoch)
    return _amax.ptr(w)


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# reference `_C` surface
# ------------------------------------------------------------------------------------------------
def nms_with_count(dets, scores, threshold, max_keep=-1, tie_rule=None):
    """device-resident result: (keep_buffer int64[n], num_keep int32[1]); no host sync.
    scores=None: `dets` are already ranked best-first (the internal ranking pass is skipped)."""
    _dev(dets, "dets")
    n = dets.shape[0]
    dets = dets.contiguous()
    if scores is not None:
        _dev(scores, "scores")
        scores = scores.contiguous()
    keep = torch.empty(n, dtype=torch.int64, device=dets.device)
    count = torch.empty(1, dtype=torch.int32, device=dets.device)
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_nms_workspace_bytes", n, ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, dets.device)
    rule = NMS_TIE_RULE if tie_rule is None else tie_rule
    _lib.call("dadet_nms", _p(dets), _p(scores), n, float(threshold), int(rule), int(max_keep), _p(ws),
              ctypes.c_size_t(ws.numel()), _p(keep), _p(count), _stream())
    return keep, count


def nms(dets, scores, threshold):
    """_C.nms(dets[N,4], scores[N], thr) -> int64[K] kept original indices, ascending (nms.h:10-28)."""
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    keep, count = nms_with_count(dets, scores, threshold)
    return keep[: int(count.item())]


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, bin_stride=1):
    """_C.roi_align_forward -> [R,C,ph,pw] (ROIAlign.h:11-25).  bin_stride s > 1: only the bins (i * s, j * s) of the
    ph x pw grid, as a compact [R, C, ceil(ph / s), ceil(pw / s)] tensor (dadet_roi_align_forward_sub)."""
    _dev(input, "input"), _dev(rois, "rois")
    B, C, H, W = input.shape
    R = rois.shape[0]
    x = _nhwc(input)
    rois = rois.contiguous()
    if bin_stride != 1:
        s = int(bin_stride)
        out = torch.empty((R, C, -(-pooled_height // s), -(-pooled_width // s)), dtype=torch.float32,
                          device=input.device, memory_format=CL)
        ws = _roi_workspace(B, H, W, R, input.device) if ROI_ALIGN_WORKSPACE else None
        _lib.call("dadet_roi_align_forward_sub", _p(x), _p(rois), _p(out), B, C, H, W, R, pooled_height, pooled_width,
                  float(spatial_scale), int(sampling_ratio), s, _p(ws) if ws is not None else None,
                  ctypes.c_size_t(ws.numel() if ws is not None else 0), _stream())
        return _amax.carry(out, input)     # averages of bilinear samples: bounded by the map's largest magnitude
    out = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.float32, device=input.device,
                      memory_format=CL)
    if ROI_ALIGN_WORKSPACE:
        ws = _roi_workspace(B, H, W, R, input.device)
        _lib.call("dadet_roi_align_forward_ws", _p(x), _p(rois), _p(out), B, C, H, W, R, pooled_height, pooled_width,
                  float(spatial_scale), int(sampling_ratio), _p(ws), ctypes.c_size_t(ws.numel()), _stream())
    else:
        _lib.call("dadet_roi_align_forward", _p(x), _p(rois), _p(out), B, C, H, W, R, pooled_height,
                  pooled_width, float(spatial_scale), int(sampling_ratio), _stream())
    return _amax.carry(out, input)


# forward: ROIs processed in spatial (Z-order) order through a scratch buffer; False = the plain entry point
ROI_ALIGN_WORKSPACE = True


def _roi_workspace(B, H, W, R, device):
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_roi_align_workspace_bytes", B, H, W, R, ctypes.byref(nbytes))
    return _workspace(nbytes.value, device)


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                       height, width, sampling_ratio, atomic=False, bin_stride=1, live_images=None):
    """_C.roi_align_backward -> [B,C,H,W] (ROIAlign.h:27-45).  bin_stride s > 1: `grad` is the compact gradient of the
    bins (i * s, j * s) (roi_align_forward with the same bin_stride)."""
    _dev(grad, "grad"), _dev(rois, "rois")
    g = _nhwc(grad)
    rois = rois.contiguous()
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device,
                      memory_format=CL)
    # live_images: only the leading images are referenced by any ROI (the caller knows: it built the batch indices); the
    # gather kernel then sweeps those images' pixel tiles only and the others' gradient is a plain zero fill
    B = batch_size
    if live_images is not None and 0 < live_images < batch_size and not atomic and rois.shape[0] > 0:
        B = int(live_images)
        gin[B:].zero_()
    if bin_stride != 1:
        if atomic:
            raise ValueError("roi_align_backward: bin_stride needs the gather form")
        _lib.call("dadet_roi_align_backward_sub", _p(g), _p(rois), _p(gin), B, channels, height, width,
                  rois.shape[0], pooled_height, pooled_width, float(spatial_scale), int(sampling_ratio),
                  int(bin_stride), _stream())
        return gin
    batch_size = B
    if atomic:
        gin.zero_()
    _lib.call("dadet_roi_align_backward_atomic" if atomic else "dadet_roi_align_backward", _p(g), _p(rois),
              _p(gin), batch_size, channels, height, width, rois.shape[0], pooled_height, pooled_width,
              float(spatial_scale), int(sampling_ratio), _stream())
    return gin


def roi_align_forward_level(input, rois, levels, level, out, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """one pyramid level of a multi-level ROIAlign (dadet_roi_align_forward_level): the ROIs with levels[r] == level are
    pooled from `input` into their rows of `out` [R,C,ph,pw] (channels_last); the other rows are not touched"""
    _dev(input, "input"), _dev(rois, "rois")
    assert levels.is_cuda and levels.dtype == torch.int64 and levels.is_contiguous() and levels.numel() == rois.shape[0]
    assert out.is_contiguous(memory_format=CL) and out.shape[0] == rois.shape[0]
    B, C, H, W = input.shape
    x = _nhwc(input)
    rois = rois.contiguous()
    ws = _roi_workspace(B, H, W, rois.shape[0], input.device)
    _lib.call("dadet_roi_align_forward_level", _p(x), _p(rois), _p(levels), int(level), _p(out), B, C, H, W, rois.shape[0],
              pooled_height, pooled_width, float(spatial_scale), int(sampling_ratio), _p(ws), ctypes.c_size_t(ws.numel()),
              _stream())
    return out


def roi_align_backward_level(grad, rois, levels, level, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                             height, width, sampling_ratio):
    """gradient of one level's map from the ROIs of that level (dadet_roi_align_backward_level) -> [B,C,H,W]"""
    _dev(grad, "grad"), _dev(rois, "rois")
    assert levels.is_cuda and levels.dtype == torch.int64 and levels.is_contiguous() and levels.numel() == rois.shape[0]
    g = _nhwc(grad)
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device, memory_format=CL)
    _lib.call("dadet_roi_align_backward_level", _p(g), _p(rois.contiguous()), _p(levels), int(level), _p(gin), batch_size,
              channels, height, width, rois.shape[0], pooled_height, pooled_width, float(spatial_scale), int(sampling_ratio),
              _stream())
    return gin


def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    """_C.sigmoid_focalloss_forward(logits[N,C], targets[N] int32, C, gamma, alpha) -> [N,C]"""
    _dev(logits, "logits")
    logits = logits.contiguous()
    targets = targets.contiguous().to(torch.int32)
    out = torch.empty_like(logits)
    _lib.call("dadet_sigmoid_focal_loss_forward", _p(logits), _p(targets), _p(out), logits.shape[0],
              int(num_classes), float(gamma), float(alpha), _stream())
    return out


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    _dev(logits, "logits")
    logits = logits.contiguous()
    targets = targets.contiguous().to(torch.int32)
    d_losses = d_losses.contiguous()
    out = torch.empty_like(logits)
    _lib.call("dadet_sigmoid_focal_loss_backward", _p(logits), _p(targets), _p(d_losses), _p(out),
              logits.shape[0], int(num_classes), float(gamma), float(alpha), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# convolution family
# ------------------------------------------------------------------------------------------------
def _desc(N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, OutH=None, OutW=None, os=1, relu_mode=0):
    return ConvDesc(N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, Ho if OutH is None else OutH,
                    Wo if OutW is None else OutW, os, relu_mode)


def conv_out_size(H, W, KH, KW, stride, pad):
    return (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1


def conv_forward(x, w, scale=None, bias=None, addend=None, mask_ref=None, stride=1, pad=0, relu_mode=0,
                 out=None, out_spatial_stride=1, out_hw=None, out_size=None):
    """y = act(conv(x, w) * scale + bias + addend); see dadet_conv_forward in include/dadet.h.

    x [N,Cin,H,W] channels_last, w [Cout,Cin,KH,KW] channels_last.  `out_size` overrides (Ho, Wo) (used
    by the stem whose kernel is zero-padded to 7x8).  With out_spatial_stride > 1 the result is scattered
    into `out` (or a fresh zero tensor) of spatial size out_hw.
    """
    _dev(x, "x"), _dev(w, "w")
    N, Cin, H, W = x.shape
    Cout, Cin_w, KH, KW = w.shape
    assert Cin == Cin_w, "conv_forward: channel mismatch %d vs %d" % (Cin, Cin_w)
    x = _nhwc(x)
    w = _nhwc(w)
    Ho, Wo = out_size if out_size is not None else conv_out_size(H, W, KH, KW, stride, pad)
    if out_spatial_stride > 1:
        OutH, OutW = out_hw
        if out is None:
            out = torch.empty((N, Cout, OutH, OutW), dtype=torch.float32, device=x.device,
                              memory_format=CL).zero_()
    else:
        OutH, OutW = Ho, Wo
        if out is None:
            out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=CL)
    if addend is not None:
        addend = _nhwc(addend) if addend is not out else addend
    if mask_ref is not None:
        mask_ref = _nhwc(mask_ref)
    d = _desc(N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, OutH, OutW, out_spatial_stride, relu_mode)
    ax = aw = None
    if _mode4():
        # measured here (outside a profiler span) when the producer left none
        ax, aw = _amax.ptr(x), _weight_amax(w)
    if PROFILER is not None:
        key = (N * Ho * Wo, Cout, Cin * KH * KW, KH, pad, out_spatial_stride)
        kname = _KNAME_CACHE.get(key)
        if kname is None:
            variant = _lib.load().dadet_conv_forward_variant(ctypes.byref(d))
            mode = get_gemm_mode()
            if variant == 4:
                kname = "conv_big_kernel<256>"
            elif variant == 5:
                kname = "conv_big128_kernel"
            elif variant == 3:
                # panel width as csrc/conv_ws.hip's launch_fwd_ws picks it: K = 256 runs 128-column panels on layers of
                # at least 256 columns (mode 4, unless DADET_WS_K256_BN=64), 64-column panels otherwise
                bn = 128
                if Cin == 256:
                    wide = mode == 4 and Cout >= 256 and os.environ.get("DADET_WS_K256_BN", "") != "64"
                    bn = 128 if wide else 64
                kname = "conv1x1_ws_kernel<%d,%d,%d>" % (Cin, bn, mode)
            else:
                kname = ("conv_fwd_kernel<%s>" if mode == 0 else ("conv_fwd_split_kernel<%%s,%d>" % mode)) % \
                    ("2,2", "2,1", "1,1")[variant]
            _KNAME_CACHE[key] = kname
        if getattr(PROFILER, "detail", False):   # tools/gemm_table.py: one row per problem shape
            # rows issued on a side stream share the GPU with the compute stream's kernels: their rates are not the kernel's
            side = x.is_cuda and torch.cuda.current_stream(x.device) != torch.cuda.default_stream(x.device)
            kname = "%s|M=%d N=%d K=%d k%dx%d s%d%s%s" % (kname, N * Ho * Wo, Cout, Cin * KH * KW, KH, KW, stride,
                                                          " +add" if addend is not None else "", " [side]" if side else "")
        # a strided 1x1 layer reads only the pixels it samples (whole rows of Cin floats): not all of x
        x_read = N * Ho * Wo * Cin if (KH == 1 and KW == 1 and stride > 1) else x.numel()
        with PROFILER.span(kname,
                           2.0 * N * Ho * Wo * Cout * Cin * KH * KW,
                           4.0 * (x_read + w.numel() + out.numel() + (addend.numel() if addend is not None else 0)
                                  + (mask_ref.numel() if mask_ref is not None else 0))):
            _conv_forward_call(d, x, w, scale, bias, addend, mask_ref, out, ax, aw)
        return out
    _conv_forward_call(d, x, w, scale, bias, addend, mask_ref, out, ax, aw)
    return out


def _conv_forward_call(d, x, w, scale, bias, addend, mask_ref, out, ax, aw):
    if ax is None:
        _lib.call("dadet_conv_forward", ctypes.byref(d), _p(x), _p(w), _p(scale), _p(bias), _p(addend),
                  _p(mask_ref), _p(out), _stream())
        return
    # mode 4: the operands' maxima go in, the epilogue leaves max|out| in a fresh slot that travels with `out`
    slot = _amax.new_slot(out.device)
    _lib.call("dadet_conv_forward_scaled", ctypes.byref(d), _p(x), _p(w), _p(scale), _p(bias), _p(addend),
              _p(mask_ref), _p(out), ax, aw, ctypes.c_void_p(slot[0]), _stream())
    _amax.attach(out, slot)


def _transpose_now(w, scale, wt):
    Cout, Cin, KH, KW = w.shape
    cout_pad = wt.shape[1]          # rows of the result: >= Cout (zero columns behind the weight's own)
    _lib.call("dadet_conv_weight_transpose_padded", _p(w), _p(scale), _p(wt), Cout, KH, KW, Cin, cout_pad, _stream())
    return wt


class _TransposeCache(object):
    """Transposed (tap-flipped, FrozenBN-folded) weights for the data-gradient GEMMs, refreshed for ALL registered weights
    by ONE launch per optimizer step instead of one launch in front of every data-gradient GEMM.

    An entry (weight storage, shape, scale storage) owns a persistent output buffer.  It is valid while (a) the weight's
    autograd version is the one it was computed from and (b) the weight epoch is — the epoch is what in-place updates
    through raw pointers bump (FusedSGD.step -> bump_weight_epoch(); ATen's in-place ops bump the version themselves).
    The first request of an epoch recomputes every entry in one batched launch on the current stream; a weight seen for
    the first time is transposed on the spot and joins the table.  `enabled = False`: one launch per request."""

    def __init__(self):
        self.entries = {}
        self.epoch = 0
        self.batched_epoch = -1
        self.table = None          # (device tensor, n, total_blocks, keys)
        self.ready = None          # (stream, event) of the last batched refresh
        self.enabled = True
        self._from_bump = False
        self._trained_only = False

    def bump(self, device=None, trained_only=False):
        """new weight epoch; with a device: refresh every entry at once, on the caller's stream (the optimizer's, behind
        the update and behind everything that read the old buffers).  trained_only: the caller changed trainable parameters
        only (the optimizer): frozen weights' maxima are left alone (amax.WeightSlots.refresh)"""
        self.epoch += 1
        self._trained_only = bool(trained_only)
        if self.enabled and device is not None and self.entries:
            self._from_bump = True
            try:
                self._refresh_all(device)
            finally:
                self._from_bump = False
        elif device is not None and _mode4():
            # no transposed copies registered: the weights' maxima alone
            _amax.WEIGHTS.refresh(device, self.epoch, trained_only=self._trained_only)
        self._trained_only = False

    def _single(self, e, w, scale):
        """one entry on the current stream, remembered with an event for readers on other streams"""
        _transpose_now(w, scale, e["wt"])
        _amax.WEIGHTS.invalidate(e["wt"])
        e["epoch"] = self.epoch
        if w.is_cuda:
            st = torch.cuda.current_stream(w.device)
            e["ev"] = (st, st.record_event())
        return e["wt"]

    def get(self, w, scale, cout_pad=0):
        Cout, Cin, KH, KW = w.shape
        key = (w.data_ptr(), Cout, Cin, KH, KW, scale.data_ptr() if scale is not None else 0, w.device.index, cout_pad)
        e = self.entries.get(key)
        if e is None:
            wt = torch.empty((Cin, max(Cout, cout_pad), KH, KW), dtype=torch.float32, device=w.device, memory_format=CL)
            # the entry keeps w / scale alive: their storage addresses are in the device table
            wt._dadet_persistent = True    # its largest magnitude lives in amax.WEIGHTS (mode 4)
            e = self.entries[key] = dict(w=w, scale=scale, wt=wt, version=w._version, epoch=self.epoch, used=self.epoch,
                                         ev=None)
            self.table = None
            return self._single(e, w, scale)
        e["used"] = self.epoch
        if e["version"] != w._version:
            e["w"], e["version"] = w, w._version
            return self._single(e, w, scale)
        if e["epoch"] != self.epoch:
            if self.batched_epoch != self.epoch:
                self._refresh_all(w.device)
            if e["epoch"] != self.epoch:      # joined the table after this epoch's launch
                return self._single(e, w, scale)
        # produced on ONE stream (the optimizer's for the batched refresh): a reader on another stream waits for it
        src = e["ev"] if e["ev"] is not None else self.ready
        if src is not None and w.is_cuda and torch.cuda.current_stream(w.device) != src[0]:
            torch.cuda.current_stream(w.device).wait_event(src[1])
        return e["wt"]

    def _refresh_all(self, device):
        # weights nobody asked for during the last two epochs belong to a model that is gone: dropped (with their buffers)
        for k in [k for k, e in self.entries.items() if e["used"] < self.epoch - 2]:
            del self.entries[k]
        live = [(k, e) for k, e in self.entries.items() if k[6] == device.index and e["version"] == e["w"]._version]
        if not live:
            self.batched_epoch = self.epoch
            return
        if self.table is None or self.table[3] != [k for k, _ in live]:
            arr = (_lib.TransposeItem * len(live))()
            blocks = 0
            for i, (k, e) in enumerate(live):
                _, Cout, Cin, KH, KW, _, _, cout_pad = k
                it = arr[i]
                it.w, it.scale, it.wt = e["w"].data_ptr(), (e["scale"].data_ptr() if e["scale"] is not None else None), \
                    e["wt"].data_ptr()
                it.Cout, it.KH, it.KW, it.Cin = Cout, KH, KW, Cin
                it.cout_pad = cout_pad
                it.blocks_ci, it.blocks_co = (Cin + 31) // 32, (max(Cout, cout_pad) + 31) // 32
                it.first_block = blocks
                blocks += it.blocks_ci * it.blocks_co * KH * KW
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self.table = (host.to(device), len(live), blocks, [k for k, _ in live])
        dev_t, n, blocks, _ = self.table
        _lib.call("dadet_conv_weight_transpose_batch", _p(dev_t), n, blocks, _stream())
        for _, e in live:
            e["epoch"] = self.epoch
            e["ev"] = None
        self.batched_epoch = self.epoch
        if _mode4():
            # parameters and the transposed copies just written: every persistent operand's maximum in one launch.  Behind
            # the optimizer (bump with a device) no other stream reads the slots; from the middle of a step they may.
            _amax.WEIGHTS.refresh(device, self.epoch, sync=not self._from_bump,
                                  trained_only=self._from_bump and self._trained_only)
        if device.type == "cuda":
            st = torch.cuda.current_stream(device)
            self.ready = (st, st.record_event())


_TRANSPOSES = _TransposeCache()


def weight_epoch():
    """counts the in-place weight updates made through raw pointers (see bump_weight_epoch)"""
    return _TRANSPOSES.epoch


def bump_weight_epoch(device=None, trained_only=False):
    """weights were updated in place through raw pointers (the fused SGD kernel) or through `.data` (a broadcast, a
    checkpoint load, an EMA, a manual `p.data.copy_`): cached derived forms are stale.  REQUIRED after any weight write
    that does not go through autograd's version counter — the data-gradient GEMMs otherwise keep using the transposed /
    padded copies of the old values, and (contraction mode 4) the old largest magnitude of the weight: a weight that grew
    past twice its recorded maximum overflows fp16's range inside the GEMM and the step ends in inf / nan, loudly.
    FusedSGD.step, BucketedGradReducer.broadcast_parameters and Checkpointer.load call it themselves.
    trained_only=True (the optimizer): only tensors with requires_grad changed — frozen weights keep their maxima."""
    _TRANSPOSES.bump(device, trained_only)


def conv_weight_transpose(w, scale=None, cout_pad=0):
    """[Cout,Cin,KH,KW] -> data-gradient weights [Cin,Cout,KH,KW] (flipped taps, `scale[cout]` folded in).  The result
    may be a cached buffer shared by later calls with the same weight: treat it as read-only.
    cout_pad > Cout: [Cin,cout_pad,KH,KW] with zero columns behind the weight's own — for an output gradient whose rows
    are padded to cout_pad channels (the offset branch of a deformable block)."""
    _dev(w, "w")
    Cout, Cin, KH, KW = w.shape
    cout_pad = cout_pad if cout_pad > Cout else 0
    w = _nhwc(w)
    # cached only for storage that persists across steps — a trainable parameter or a view of one (a temporary's address may
    # be handed to another tensor by the allocator, and the cache is keyed by address)
    base = w._base if w._base is not None else w
    if _TRANSPOSES.enabled and base.is_leaf and base.requires_grad:
        return _TRANSPOSES.get(w, scale, cout_pad)
    wt = torch.empty((Cin, max(Cout, cout_pad), KH, KW), dtype=torch.float32, device=w.device, memory_format=CL)
    return _transpose_now(w, scale, wt)


class WgradBatch(list):
    """weight gradients whose reduction pass over the split partial results is still to come
    (conv_wgrad(..., pending=batch) ... conv_wgrad_reduce_batch(batch)); keeps their workspaces alive until then"""


def _wgrad_kname(d, dense_rows):
    """the kernel a weight-gradient launch of this shape runs (profiling labels)"""
    mode = get_gemm_mode()
    if mode == 0:
        return "conv_wgrad_kernel"
    if dense_rows and _lib.load().dadet_conv_wgrad_variant(ctypes.byref(d)) == 1:
        return "conv_wgrad_big_kernel"
    return "conv_wgrad_split_kernel<%d>" % mode


def conv_wgrad_reduce_batch(batch):
    """one launch for the reduction passes collected in `batch` (dadet_conv_wgrad_reduce_batch); same stream as the GEMMs"""
    if not batch:
        return
    arr = (_lib.WgradPending * len(batch))(*[item[0] for item in batch])
    _lib.call("dadet_conv_wgrad_reduce_batch", arr, len(batch), _stream())
    del batch[:]


WGRAD_GROUP_MAX = 4


def conv_wgrad_group(requests, pending):
    """the weight gradients of ONE backward node in one launch (dadet_conv_wgrad_group; contraction mode 4).
    requests: 1 - 4 dicts(x, gy, weight_shape, stride, pad, out_scale, dw, accumulate) as conv_wgrad's arguments, dw given
    (the parameters' gradient buffers, all different); pending: the WgradBatch that takes their reduction passes.
    -> True when the launch was made; False when some request does not qualify (nothing was launched: call conv_wgrad per
    request)."""
    n = len(requests)
    if not (1 <= n <= WGRAD_GROUP_MAX) or not _mode4() or pending is None:
        return False
    descs = (_lib.ConvDesc * n)()
    xs, gys = [], []
    for i, r in enumerate(requests):
        x, gy = _nhwc(_dev(r["x"], "x")), _nhwc(_dev(r["gy"], "gy"))
        N, Cin, H, W = x.shape
        Cout, Cin_w, KH, KW = r["weight_shape"]
        if Cin != Cin_w or gy.shape[1] != Cout or r["dw"] is None:
            return False
        descs[i] = _desc(N, H, W, Cin, Cout, KH, KW, r.get("stride", 1), r.get("pad", 0), gy.shape[2], gy.shape[3])
        xs.append(x)
        gys.append(gy)
    splits = (ctypes.c_int * n)()
    nbytes = (ctypes.c_size_t * n)()
    kind = _lib.load().dadet_conv_wgrad_group_plan(descs, n, splits, nbytes)
    if not kind:
        return False
    wss = [torch.empty(max(int(b), 16), dtype=torch.uint8, device=xs[0].device) for b in nbytes]
    vp = ctypes.c_void_p * n

    def ptrs(ts):
        return vp(*[(t.data_ptr() if t is not None else None) for t in ts])

    items = (_lib.WgradPending * n)()
    ax = vp(*[_amax.ptr(t).value for t in xs])
    ag = vp(*[_amax.ptr(t).value for t in gys])
    acc = (ctypes.c_int * n)(*[1 if r.get("accumulate", False) else 0 for r in requests])
    sizes = (ctypes.c_size_t * n)(*[w.numel() for w in wss])

    def launch():
        _lib.call("dadet_conv_wgrad_group", descs, n, ptrs(xs), ptrs(gys), ptrs([r.get("out_scale") for r in requests]),
                  ptrs([r["dw"] for r in requests]), acc, ptrs(wss), sizes, items, ax, ag, _stream())

    if PROFILER is not None:
        flops = sum(2.0 * d.N * d.Ho * d.Wo * d.Cout * d.Cin * d.KH * d.KW for d in descs)
        nbytes_alg = sum(4.0 * (x.numel() + g.numel() + r["dw"].numel()) for x, g, r in zip(xs, gys, requests))
        kname = "conv_wgrad_big_group_kernel" if kind == 256 else "conv_wgrad_split_group_kernel"
        if getattr(PROFILER, "detail", False):
            kname += "|" + " + ".join("M=%d N=%d K=%d k%dx%d s%d" % (d.N * d.Ho * d.Wo, d.Cout, d.Cin * d.KH * d.KW, d.KH,
                                                                     d.KW, d.stride) for d in descs)
        with PROFILER.span(kname, flops, nbytes_alg):
            launch()
    else:
        launch()
    for i, r in enumerate(requests):
        if items[i].splits > 1:
            it = _lib.WgradPending()
            ctypes.memmove(ctypes.byref(it), ctypes.byref(items[i]), ctypes.sizeof(it))
            pending.append((it, wss[i], r["dw"], r.get("out_scale")))
    return True


def conv_wgrad(x, gy, weight_shape, stride=1, pad=0, out_scale=None, dw=None, accumulate=False, pending=None):
    """dw[co,ci,r,s] = out_scale[co] * sum_m gy[m,co] * x[gather(m,r,s),ci]  (channels_last weight layout).
    pending (a WgradBatch): the reduction over split partial results is left to conv_wgrad_reduce_batch(pending) — dw is
    complete only after that call."""
    _dev(x, "x"), _dev(gy, "gy")
    N, Cin, H, W = x.shape
    Cout, Cin_w, KH, KW = weight_shape
    assert Cin == Cin_w
    x = _nhwc(x)
    gy = _nhwc(gy)
    Ho, Wo = gy.shape[2], gy.shape[3]
    # rows of gy padded to a multiple of four channels (Cout itself need not be one): dadet_conv_wgrad_partials_ld
    gy_ld = gy.shape[1]
    if gy_ld != Cout and pending is None:
        own = WgradBatch()
        dw = conv_wgrad(x, gy, weight_shape, stride, pad, out_scale, dw, accumulate, pending=own)
        conv_wgrad_reduce_batch(own)
        return dw
    if dw is None:
        dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=x.device, memory_format=CL)
        accumulate = False
    d = _desc(N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo)
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_conv_wgrad_workspace_bytes", ctypes.byref(d), ctypes.byref(nbytes))
    m4 = _mode4()
    if m4:
        ax, ag = _amax.ptr(x), _amax.ptr(gy)
    if pending is not None:
        # own workspace (the shared one is overwritten by the next weight gradient), alive until the batched pass
        ws = torch.empty(max(nbytes.value, 16), dtype=torch.uint8, device=x.device)
        item = _lib.WgradPending()

        def launch():
            if m4:
                _lib.call("dadet_conv_wgrad_scaled", ctypes.byref(d), _p(x), _p(gy), gy_ld, _p(out_scale), _p(dw),
                          1 if accumulate else 0, _p(ws), ctypes.c_size_t(ws.numel()), ctypes.byref(item), ax, ag,
                          _stream())
                return
            _lib.call("dadet_conv_wgrad_partials_ld", ctypes.byref(d), _p(x), _p(gy), gy_ld, _p(out_scale), _p(dw),
                      1 if accumulate else 0, _p(ws), ctypes.c_size_t(ws.numel()), ctypes.byref(item), _stream())

        if PROFILER is not None:
            kname = _wgrad_kname(d, gy_ld == Cout)
            if getattr(PROFILER, "detail", False):
                kname = "%s|M=%d N=%d K=%d k%dx%d s%d" % (kname, N * Ho * Wo, Cout, Cin * KH * KW, KH, KW, stride)
            with PROFILER.span(kname, 2.0 * N * Ho * Wo * Cout * Cin * KH * KW,
                               4.0 * (x.numel() + gy.numel() + dw.numel())):
                launch()
        else:
            launch()
        if item.splits > 1:
            pending.append((item, ws, dw, out_scale))
        return dw
    ws = _workspace(nbytes.value, x.device)
    if PROFILER is not None:
        kname = _wgrad_kname(d, gy_ld == Cout)
        if getattr(PROFILER, "detail", False):
            kname = "%s|M=%d N=%d K=%d k%dx%d s%d" % (kname, N * Ho * Wo, Cout, Cin * KH * KW, KH, KW, stride)
        with PROFILER.span(kname,
                           2.0 * N * Ho * Wo * Cout * Cin * KH * KW,
                           4.0 * (x.numel() + gy.numel() + dw.numel())):
            _wgrad_call(d, x, gy, gy_ld, out_scale, dw, accumulate, ws, ax if m4 else None, ag if m4 else None)
        return dw
    _wgrad_call(d, x, gy, gy_ld, out_scale, dw, accumulate, ws, ax if m4 else None, ag if m4 else None)
    return dw


def _wgrad_call(d, x, gy, gy_ld, out_scale, dw, accumulate, ws, ax, ag):
    if ax is not None:
        _lib.call("dadet_conv_wgrad_scaled", ctypes.byref(d), _p(x), _p(gy), gy_ld, _p(out_scale), _p(dw),
                  1 if accumulate else 0, _p(ws), ctypes.c_size_t(ws.numel()), None, ax, ag, _stream())
        return
    _lib.call("dadet_conv_wgrad", ctypes.byref(d), _p(x), _p(gy), _p(out_scale), _p(dw),
              1 if accumulate else 0, _p(ws), ctypes.c_size_t(ws.numel()), _stream())


def relu_bn_backward(g, y=None, scale=None, want_unscaled=False):
    """(g * (y > 0)) and optionally that times scale[c]; returns (g_masked or None, g_scaled)."""
    _dev(g, "g")
    g = _nhwc(g) if g.dim() == 4 else g.contiguous()
    C = g.shape[1]
    rows = g.numel() // C
    yy = None
    if y is not None:
        yy = _nhwc(y) if y.dim() == 4 else y.contiguous()
    g_out = torch.empty_like(g) if want_unscaled else None
    g_scaled = torch.empty_like(g)
    if not _mode4():
        _lib.call("dadet_relu_bn_backward", _p(g), _p(yy), _p(scale), _p(g_out), _p(g_scaled), rows, C, _stream())
        return g_out, g_scaled
    # contraction mode 4: both outputs feed GEMMs.  A gate only removes values, so the input's largest magnitude bounds
    # the gated copy when the input carries one; otherwise (and for the scaled copy) the kernel leaves the exact maxima
    so = ss = None
    if g_out is not None and _amax.slot_of(_amax.carry(g_out, g)) is None:
        so = _amax.new_slot(g.device)
    if scale is not None or _amax.slot_of(_amax.carry(g_scaled, g)) is None:
        ss = _amax.new_slot(g.device)
    _lib.call("dadet_relu_bn_backward_m", _p(g), _p(yy), _p(scale), _p(g_out), _p(g_scaled), rows, C,
              ctypes.c_void_p(so[0]) if so else None, ctypes.c_void_p(ss[0]) if ss else None, _stream())
    if so:
        _amax.attach(g_out, so)
    if ss:
        _amax.attach(g_scaled, ss)
    return g_out, g_scaled


def colsum(g, out=None, accumulate=False, cols=None):
    """sum over every axis but channels of a channels_last / [M,C] tensor -> [C].  cols: only the first `cols` channels
    (rows stay C floats apart); out + accumulate: added into an existing [cols] buffer (a bias gradient's slot in the
    flat gradient bucket)"""
    _dev(g, "g")
    g = _nhwc(g) if g.dim() == 4 else g.contiguous()
    ld = g.shape[1]
    C = ld if cols is None else int(cols)
    rows = g.numel() // ld
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=g.device)
        accumulate = False
    assert out.numel() == C and out.is_contiguous() and out.dtype == torch.float32
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_colsum_workspace_bytes", rows, C, ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, g.device)
    _lib.call("dadet_colsum_ld", _p(g), ld, _p(out), rows, C, 1 if accumulate else 0, _p(ws), ctypes.c_size_t(ws.numel()),
              _stream())
    return out


def channel_affine(x, scale, bias, relu=False):
    _dev(x, "x")
    x = _nhwc(x)
    C = x.shape[1]
    y = torch.empty_like(x)
    _lib.call("dadet_channel_affine", _p(x), _p(scale), _p(bias), _p(y), x.numel() // C, C,
              1 if relu else 0, _stream())
    return y


def maxpool3x3s2(x):
    _dev(x, "x")
    x = _nhwc(x)
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=CL)
    _lib.call("dadet_maxpool3x3s2_forward", _p(x), _p(y), N, H, W, C, Ho, Wo, _stream())
    return _amax.carry(y, x)


def avgpool_forward(x):
    """[R,C,h,w] channels_last -> [R,C] (mean over h*w)"""
    _dev(x, "x")
    x = _nhwc(x)
    R, C, h, w = x.shape
    y = torch.empty((R, C), dtype=torch.float32, device=x.device)
    _lib.call("dadet_avgpool_forward", _p(x), _p(y), R, h * w, C, _stream())
    return _amax.carry(y, x)


def avgpool_backward(gy, h, w):
    _dev(gy, "gy")
    gy = gy.contiguous()
    R, C = gy.shape
    gx = torch.empty((R, C, h, w), dtype=torch.float32, device=gy.device, memory_format=CL)
    _lib.call("dadet_avgpool_backward", _p(gy), _p(gx), R, h * w, C, _stream())
    return _amax.carry(gx, gy)     # gy / (h w): a bound (a few binades above the maximum cost nothing that matters)


def nchw3_to_nhwc4(x):
    """[N,3,H,W] contiguous NCHW image batch -> [N,4,H,W] channels_last (4th channel zero)"""
    _dev(x, "x")
    x = x.contiguous()
    N, C, H, W = x.shape
    assert C == 3
    y = torch.empty((N, 4, H, W), dtype=torch.float32, device=x.device, memory_format=CL)
    _lib.call("dadet_nchw3_to_nhwc4", _p(x), _p(y), N, H, W, _stream())
    if _mode4():
        _amax.ptr(x)              # three of the staged tensor's four channels (a batch tensor that is reused is measured once)
        _amax.carry(y, x)
    return y


def da_ins_tail_forward(h, w3, b3, labels, means, r_bce, r_cst, n_src):
    """h [r_bce + r_cst, C], w3 [C], b3 [1], labels float [r_bce] | None, means [L, 2] | None
    -> (logits [rows], sums [2] = (BCE sum, consistency sum))"""
    _dev(h, "h")
    rows, C = h.shape
    assert rows == r_bce + r_cst
    L = int(means.shape[0]) if means is not None else 0
    logits = torch.empty(rows, dtype=torch.float32, device=h.device)
    sums = torch.zeros(2, dtype=torch.float32, device=h.device)
    _lib.call("dadet_da_ins_tail_forward", _p(h), _p(w3), _p(b3), _p(labels), _p(means), _p(logits), _p(sums),
              int(r_bce), int(r_cst), int(n_src), L, C, _stream())
    return logits, sums


def da_ins_tail_backward(h, w3, logits, labels, means, coef, inv_keep, r_bce, r_cst, n_src):
    """-> (g_z [rows, C] gradient w.r.t. the last hidden layer's PRE-activation, g_w3 [C], g_b3 [1], g_means [L, 2] | None)"""
    rows, C = h.shape
    L = int(means.shape[0]) if means is not None else 0
    g_z = torch.empty_like(h)
    acc = torch.zeros(C + 1 + 2 * L, dtype=torch.float32, device=h.device)      # one zero-fill for the three sums
    g_w3, g_b3 = acc[:C], acc[C:C + 1]
    g_means = acc[C + 1:].view(L, 2) if L else None
    _lib.call("dadet_da_ins_tail_backward", _p(h), _p(w3), _p(logits), _p(labels), _p(means), _p(coef), float(inv_keep),
              _p(g_z), _p(g_w3), _p(g_b3), _p(g_means), int(r_bce), int(r_cst), int(n_src), L, C, _stream())
    return g_z, g_w3, g_b3, g_means


def da_ins_dropout_rows(h1, masks):
    """h1 [R, C], masks [P, R, C] -> [P * R, C]: the passes' dropped copies of the shared hidden layer"""
    P = masks.shape[0]
    out = torch.empty((P * h1.shape[0], h1.shape[1]), dtype=torch.float32, device=h1.device)
    _lib.call("dadet_da_ins_dropout_rows", _p(h1), _p(masks), _p(out), ctypes.c_int64(h1.numel()), int(P), _stream())
    return out


def da_ins_merge(g, masks, h1, grl, need_x=True):
    """g [P * R, C], masks [P, R, C], h1 [R, C], grl float [P] (device) -> (g_w [R, C], g_x [R, C] | None)"""
    P = masks.shape[0]
    g_w = torch.empty_like(h1)
    g_x = torch.empty_like(h1) if need_x else None
    _lib.call("dadet_da_ins_merge", _p(g), _p(masks), _p(h1), _p(grl), _p(g_w), _p(g_x), ctypes.c_int64(h1.numel()),
              int(P), _stream())
    return g_w, g_x


def rpn_decode_clip(deltas_nhwc, anchors, topk_idx, weights, xform_clip, im_w, im_h):
    """decode + clip the top-k anchors of ONE image.  deltas_nhwc: [H,W,A*4] (any view whose storage is
    that order), anchors [H*W*A,4], topk_idx int64[K] -> boxes [K,4]."""
    K = topk_idx.shape[0]
    out = torch.empty((K, 4), dtype=torch.float32, device=anchors.device)
    wx, wy, ww, wh = weights
    _lib.call("dadet_rpn_decode_clip", _p(deltas_nhwc), _p(anchors), _p(topk_idx), K, float(wx), float(wy),
              float(ww), float(wh), float(xform_clip), float(im_w), float(im_h), _p(out), _stream())
    return out


def da_img_head_loss_forward(t, w2, b2, labels, num_images, rows_per_image):
    """t [M,C1] (physical), w2 [C1], b2 [1], labels [num_images] -> (logits [M], sums [num_images,2])"""
    C1 = w2.numel()
    M = num_images * rows_per_image
    logits = torch.empty(M, dtype=torch.float32, device=t.device)
    sums = torch.zeros((num_images, 2), dtype=torch.float32, device=t.device)
    _lib.call("dadet_da_img_head_loss_forward", _p(t), _p(w2), _p(b2), _p(labels), _p(logits), _p(sums),
              num_images, rows_per_image, C1, _stream())
    return logits, sums


def da_img_head_loss_backward_g(t, w2, logits, labels, g_bce, g_mean_sig, w_adv, w_cst, num_images, rows_per_image,
                                need_x=True):
    """da_img_head_loss_backward with the coefficients formed in the kernel: g_bce 0-d / [1] tensor, g_mean_sig [N] tensor or
    None, w_adv float or 0-d device tensor, w_cst float -> (g_t_w, g_t_x | None, g_w2 [C1], g_b2 [1])"""
    C1 = w2.numel()
    g_t_w = torch.empty_like(t)
    g_t_x = torch.empty_like(t) if need_x else None
    acc = torch.zeros(C1 + 1, dtype=torch.float32, device=t.device)       # one zero fill for both sums
    g_w2, g_b2 = acc[:C1], acc[C1:]
    adv_dev = w_adv if isinstance(w_adv, torch.Tensor) else None
    # contraction mode 4: the two maps feed GEMMs; the kernel leaves their largest magnitudes in fresh slots
    sw = _amax.new_slot(t.device) if _mode4() else None
    sx = _amax.new_slot(t.device) if (sw is not None and need_x) else None
    _lib.call("dadet_da_img_head_loss_backward_gm", _p(t), _p(w2), _p(logits), _p(labels), _p(g_bce.contiguous()),
              _p(g_mean_sig.contiguous()) if g_mean_sig is not None else None,
              _p(adv_dev.reshape(1).to(torch.float32)) if adv_dev is not None else None,
              0.0 if adv_dev is not None else float(w_adv), float(w_cst), _p(g_t_w), _p(g_t_x), _p(g_w2), _p(g_b2),
              num_images, rows_per_image, C1, ctypes.c_void_p(sw[0]) if sw else None,
              ctypes.c_void_p(sx[0]) if sx else None, _stream())
    if sw:
        _amax.attach(g_t_w, sw)
    if sx:
        _amax.attach(g_t_x, sx)
    return g_t_w, g_t_x, g_w2, g_b2


def da_img_head_loss_backward(t, w2, logits, labels, coef, num_images, rows_per_image, need_x=True):
    C1 = w2.numel()
    g_t_w = torch.empty_like(t)
    g_t_x = torch.empty_like(t) if need_x else None
    g_w2 = torch.zeros(C1, dtype=torch.float32, device=t.device)
    g_b2 = torch.zeros(1, dtype=torch.float32, device=t.device)
    _lib.call("dadet_da_img_head_loss_backward", _p(t), _p(w2), _p(logits), _p(labels), _p(coef),
              _p(g_t_w), _p(g_t_x), _p(g_w2), _p(g_b2), num_images, rows_per_image, C1, _stream())
    return g_t_w, g_t_x, g_w2, g_b2


def triplet_w_forward(a, p, n, margin, eps=1e-6):
    """a, p, n: [1,C,H,W] channels_last maps -> (loss mean over C*H, dist [H*C,2])"""
    a, p, n = _nhwc(a), _nhwc(p), _nhwc(n)
    _, C, H, W = a.shape
    dist = torch.empty((H * C, 2), dtype=torch.float32, device=a.device)
    loss_sum = torch.zeros(1, dtype=torch.float32, device=a.device)
    _lib.call("dadet_triplet_w_forward", _p(a), _p(p), _p(n), H, W, C, float(margin), float(eps), _p(dist),
              _p(loss_sum), _stream())
    return loss_sum / float(H * C), dist


def triplet_w_backward(a, p, n, dist, g_scale, margin, eps=1e-6, need=(True, True, True)):
    a, p, n = _nhwc(a), _nhwc(p), _nhwc(n)
    _, C, H, W = a.shape
    ga = torch.empty_like(a) if need[0] else None
    gp = torch.empty_like(p) if need[1] else None
    gn = torch.empty_like(n) if need[2] else None
    _lib.call("dadet_triplet_w_backward", _p(a), _p(p), _p(n), _p(dist), _p(g_scale), H, W, C, float(margin),
              float(eps), _p(ga), _p(gp), _p(gn), _stream())
    return ga, gp, gn


def deform_sample_forward(x, offset, mask, kh, kw, stride, pad, dil, dg):
    """x [N,C,H,W], offset [N,dg*2*kh*kw,Ho,Wo], mask [N,dg*kh*kw,Ho,Wo] | None (all channels_last)
    -> cols [N, kh*kw*C, Ho, Wo] channels_last (channel = tap*C + c)"""
    _dev(x, "x"), _dev(offset, "offset")
    x, offset = _nhwc(x), _nhwc(offset)
    mask = _nhwc(mask) if mask is not None else None
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[2], offset.shape[3]
    cols = torch.empty((N, kh * kw * C, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=CL)
    _lib.call("dadet_deform_sample_forward", _p(x), _p(offset), _p(mask), _p(cols), N, H, W, C, kh, kw, stride, pad,
              dil, dg, Ho, Wo, _stream())
    return cols


def deform_sample_backward(x, offset, mask, gcols, kh, kw, stride, pad, dil, dg, need_x=True):
    x, offset, gcols = _nhwc(x), _nhwc(offset), _nhwc(gcols)
    mask = _nhwc(mask) if mask is not None else None
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[2], offset.shape[3]
    T = kh * kw
    gx = torch.empty_like(x).zero_() if need_x else None
    goffset = torch.empty_like(offset).zero_()
    gmask = torch.empty_like(mask).zero_() if mask is not None else None
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_deform_sample_backward_workspace_bytes", N, H, W, dg, ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, x.device)      # per-cell sample lists of the gather form (csrc/deform.hip)
    _lib.call("dadet_deform_sample_backward_ld", _p(x), _p(offset), dg * 2 * T, _p(mask), dg * T, 0, _p(gcols), _p(gx),
              _p(goffset), dg * 2 * T, _p(gmask), dg * T, N, H, W, C, kh, kw, stride, pad, dil, dg, Ho, Wo, _p(ws),
              ctypes.c_size_t(ws.numel()), _stream())
    return gx, goffset, gmask


def _off_ptr(t, floats):
    return ctypes.c_void_p(t.data_ptr() + 4 * int(floats))


def deform_sample_forward_om(x, om, kh, kw, stride, pad, dil, dg, modulated):
    """deformable sampling that reads offsets (channels [0, 2T*dg)) and modulation LOGITS (channels [2T*dg, 3T*dg),
    sigmoid applied in the kernel) straight out of `om` [N, ld, Ho, Wo] channels_last — the offset-predicting conv's
    output, ld >= the channels used (padded to a multiple of 4).  -> cols [N, kh*kw*C, Ho, Wo]"""
    _dev(x, "x"), _dev(om, "om")
    x, om = _nhwc(x), _nhwc(om)
    N, C, H, W = x.shape
    ld, Ho, Wo = om.shape[1], om.shape[2], om.shape[3]
    T = kh * kw
    assert ld >= dg * T * (3 if modulated else 2)
    cols = torch.empty((N, T * C, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=CL)
    _lib.call("dadet_deform_sample_forward_ld", _p(x), _p(om), ld, _off_ptr(om, 2 * T * dg) if modulated else None, ld,
              1, _p(cols), N, H, W, C, kh, kw, stride, pad, dil, dg, Ho, Wo, _stream())
    # bilinear samples (weights summing to at most one) times a sigmoid: bounded by the map's largest magnitude
    return _amax.carry(cols, x)


def deform_sample_backward_om(x, om, gcols, kh, kw, stride, pad, dil, dg, modulated):
    """-> (gx [N,C,H,W], gom [N, ld, Ho, Wo]): gradients w.r.t. the sampled map and w.r.t. the offset conv's output
    (offset channels, modulation logits; the padding channels stay zero)"""
    x, om, gcols = _nhwc(x), _nhwc(om), _nhwc(gcols)
    N, C, H, W = x.shape
    ld, Ho, Wo = om.shape[1], om.shape[2], om.shape[3]
    T = kh * kw
    # both gradients in ONE zero-filled allocation (one fill launch per deformable block instead of two; x.numel() is a
    # multiple of 4 floats, so the second view stays 16-byte aligned)
    flat = torch.zeros(x.numel() + om.numel(), dtype=torch.float32, device=x.device)
    gx = flat[:x.numel()].view(N, H, W, C).permute(0, 3, 1, 2)
    gom = flat[x.numel():].view(N, Ho, Wo, ld).permute(0, 3, 1, 2)
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_deform_sample_backward_workspace_bytes", N, H, W, dg, ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, x.device)
    if _mode4():
        # gom is an operand of the offset conv's weight-gradient GEMM: the pass that stores it leaves its largest magnitude
        # (one dadet_amax pass + its launch gap less per deformable block: 30 per step of R-101-FPN-DCN)
        slot = _amax.new_slot(x.device)
        _lib.call("dadet_deform_sample_backward_ld_m", _p(x), _p(om), ld, _off_ptr(om, 2 * T * dg) if modulated else None,
                  ld, 1, _p(gcols), _p(gx), _p(gom), ld, _off_ptr(gom, 2 * T * dg) if modulated else None, ld,
                  N, H, W, C, kh, kw, stride, pad, dil, dg, Ho, Wo, _p(ws), ctypes.c_size_t(ws.numel()),
                  ctypes.c_void_p(slot[0]), ctypes.c_longlong(gom.numel()), _stream())
        _amax.attach(gom, slot)
        return gx, gom
    _lib.call("dadet_deform_sample_backward_ld", _p(x), _p(om), ld, _off_ptr(om, 2 * T * dg) if modulated else None, ld,
              1, _p(gcols), _p(gx), _p(gom), ld, _off_ptr(gom, 2 * T * dg) if modulated else None, ld,
              N, H, W, C, kh, kw, stride, pad, dil, dg, Ho, Wo, _p(ws), ctypes.c_size_t(ws.numel()), _stream())
    return gx, gom


def rpn_loss(objectness, box_regression, sampled_inds, labels_sampled, pos_inds, targets_pos, beta):
    """single-level RPN losses + their gradients in one launch (dadet_rpn_loss).  objectness [N,A,H,W] and
    box_regression [N,4A,H,W] channels_last -> (losses [2], grad_objectness, grad_box_regression)"""
    _dev(objectness, "objectness"), _dev(box_regression, "box_regression")
    obj, reg = _nhwc(objectness), _nhwc(box_regression)
    losses = torch.empty(2, dtype=torch.float32, device=obj.device)
    g_obj = torch.empty_like(obj).zero_()
    g_reg = torch.empty_like(reg).zero_()
    _lib.call("dadet_rpn_loss", _p(obj), _p(reg), _p(sampled_inds.contiguous()), _p(labels_sampled.contiguous()),
              int(sampled_inds.numel()), _p(pos_inds.contiguous()), _p(targets_pos.contiguous()),
              int(pos_inds.numel()), float(beta), _p(losses), _p(g_obj), _p(g_reg), _stream())
    return losses, g_obj, g_reg


def rpn_loss_rows(objectness, box_regression, sampled_inds, labels_sampled, num_pos, targets_pos, beta, level=None,
                  shared=None):
    """RPN losses with the gradient in row form (dadet_rpn_loss_rows): -> (losses [2], grad rows [S, ldg] with ldg = 5A
    rounded up to a multiple of 4, pixel index of every row int32 [S]).
    level = (anchors per image over all levels, this level's offset in an image, this level's anchors H*W*A): the maps are
    ONE level of a pyramid, sampled_inds index the concatenation over levels; rows of other levels stay zero, pixel -1
    (dadet_rpn_loss_rows_level).
    shared = (level id, rows [S, ldg] zero-filled once by the caller, pixels int32 [S], row_level int32 [S]): the buffers
    of all levels' launches — this one writes the rows of its own anchors and tags them; -> (losses [2], rows, pixels)"""
    _dev(objectness, "objectness"), _dev(box_regression, "box_regression")
    obj, reg = _nhwc(objectness), _nhwc(box_regression)
    A = obj.shape[1]
    S = int(sampled_inds.numel())
    ldg = (5 * A + 3) // 4 * 4
    losses = torch.empty(2, dtype=torch.float32, device=obj.device)
    if shared is not None:
        level_id, rows, pixels, row_level = shared
        assert rows.shape == (S, ldg) and rows.is_contiguous() and pixels.dtype == row_level.dtype == torch.int32
    else:
        level_id, row_level = 0, None
        rows = torch.empty((S, ldg), dtype=torch.float32, device=obj.device)
        pixels = torch.empty(S, dtype=torch.int32, device=obj.device)
    if level is not None:
        per_image, off, cnt = (int(v) for v in level)
        assert cnt == obj.shape[1] * obj.shape[2] * obj.shape[3]
        _lib.call("dadet_rpn_loss_rows_level", _p(obj), _p(reg), _p(sampled_inds.contiguous()),
                  _p(labels_sampled.contiguous()), S, int(num_pos), _p(targets_pos.contiguous()), A, float(beta), per_image,
                  off, cnt, int(level_id), _p(row_level), 1 if shared is not None else 0, _p(losses), _p(rows), ldg,
                  _p(pixels), _stream())
        return losses, rows, pixels
    _lib.call("dadet_rpn_loss_rows", _p(obj), _p(reg), _p(sampled_inds.contiguous()), _p(labels_sampled.contiguous()),
              S, int(num_pos), _p(targets_pos.contiguous()), A, float(beta), _p(losses), _p(rows), ldg, _p(pixels),
              _stream())
    return losses, rows, pixels


def gather_pixel_taps(x, pixels, ksize=1, pad=0, row_level=None, level=0, out=None):
    """x [N,C,H,W] channels_last, pixels int32 [S] -> [S, ksize*ksize, C]: the rows x[pixel + tap offset] (zero outside).
    row_level / level / out: only the rows tagged `level` are written, into the shared buffer `out`"""
    _dev(x, "x")
    assert pixels.is_cuda and pixels.dtype == torch.int32
    N, C, H, W = x.shape
    x = _nhwc(x)
    S = int(pixels.numel())
    if out is None:
        out = torch.empty((S, ksize * ksize, C), dtype=torch.float32, device=x.device)
    if row_level is not None:
        _lib.call("dadet_gather_pixel_taps_level", _p(x), _p(pixels), _p(row_level), int(level), S, N, H, W, C, ksize,
                  ksize, pad, _p(out), _stream())
        return out
    _lib.call("dadet_gather_pixel_taps", _p(x), _p(pixels), S, N, H, W, C, ksize, ksize, pad, _p(out), _stream())
    return _amax.carry(out, x)


def scatter_pixel_taps_add(y, pixels, shape, ksize=1, pad=0, row_level=None, level=0):
    """y [S, ksize*ksize, C] -> zero [N,C,H,W] channels_last map with y[r, tap] added at pixel_r + tap offset
    (row_level / level: only the rows tagged `level`)"""
    _dev(y, "y")
    assert pixels.is_cuda and pixels.dtype == torch.int32
    N, C, H, W = shape
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=y.device, memory_format=CL).zero_()
    if row_level is not None:
        _lib.call("dadet_scatter_pixel_taps_add_level", _p(y.contiguous()), _p(pixels), _p(row_level), int(level),
                  int(pixels.numel()), N, H, W, C, ksize, ksize, pad, _p(dx), _stream())
        return dx
    _lib.call("dadet_scatter_pixel_taps_add", _p(y.contiguous()), _p(pixels), int(pixels.numel()), N, H, W, C, ksize,
              ksize, pad, _p(dx), _stream())
    return dx


def fpn_merge_levels(per_image, post_n):
    """per_image[i] = [(boxes [n,4] in score order, scores [n], keep int64 [n], count int32 [1]) per level] -> per image
    (boxes [cap_i, 4], scores [cap_i]) with the levels' NMS results laid end to end, cap = min(n, post_n) slots per level,
    score -1 behind a level's kept count (dadet_fpn_merge_levels: one launch, no count read back)"""
    entries, outs, keepalive = [], [], []
    for levels in per_image:
        caps = [min(int(b.shape[0]), int(post_n)) for b, _, _, _ in levels]
        dev = levels[0][0].device
        boxes_out = torch.empty((sum(caps), 4), dtype=torch.float32, device=dev)
        scores_out = torch.empty(sum(caps), dtype=torch.float32, device=dev)
        off = 0
        for (b, sc, keep, count), cap in zip(levels, caps):
            if cap == 0:
                continue
            b, sc = b.contiguous(), sc.contiguous()
            keepalive.append((b, sc))
            e = _lib.MergeEntry()
            e.boxes, e.scores, e.keep, e.count = b.data_ptr(), sc.data_ptr(), keep.data_ptr(), count.data_ptr()
            e.boxes_out, e.scores_out = boxes_out[off:].data_ptr(), scores_out[off:].data_ptr()
            e.n, e.cap = int(b.shape[0]), cap
            entries.append(e)
            off += cap
        outs.append((boxes_out, scores_out))
    for i in range(0, len(entries), 24):
        chunk = entries[i:i + 24]
        arr = (_lib.MergeEntry * len(chunk))(*chunk)
        _lib.call("dadet_fpn_merge_levels", arr, len(chunk), _stream())
    return outs


def fast_rcnn_loss(class_logits, box_regression, src, labels_src, rows_pos, map_inds, targets_pos):
    """Fast R-CNN losses + gradients in one launch (dadet_fast_rcnn_loss) -> (losses [2], g_cls, g_reg)"""
    _dev(class_logits, "class_logits"), _dev(box_regression, "box_regression")
    cls, reg = class_logits.contiguous(), box_regression.contiguous()
    losses = torch.empty(2, dtype=torch.float32, device=cls.device)
    g_cls, g_reg = torch.zeros_like(cls), torch.zeros_like(reg)
    _lib.call("dadet_fast_rcnn_loss", _p(cls), _p(reg), cls.shape[1], reg.shape[1], _p(src.contiguous()),
              _p(labels_src.contiguous()), int(src.numel()), _p(rows_pos.contiguous()), _p(map_inds.contiguous()),
              _p(targets_pos.contiguous()), int(rows_pos.numel()), _p(losses), _p(g_cls), _p(g_reg), _stream())
    return losses, g_cls, g_reg


def fast_rcnn_loss_rows(class_logits, box_regression, loss_labels, regression_targets):
    """Fast R-CNN losses + gradients from per-row targets (dadet_fast_rcnn_loss_rows) -> (losses [2], g_cls, g_reg);
    loss_labels int64 [R] (< 0: row outside the losses), regression_targets [R,4]"""
    _dev(class_logits, "class_logits"), _dev(box_regression, "box_regression")
    cls, reg = class_logits.contiguous(), box_regression.contiguous()
    R = cls.shape[0]
    if loss_labels.numel() != R or regression_targets.numel() != 4 * R or reg.shape[0] != R:
        raise _lib.DadetError("fast_rcnn_loss_rows: %d logit rows, %d labels, %d target values" % (
            R, loss_labels.numel(), regression_targets.numel()))
    losses = torch.empty(2, dtype=torch.float32, device=cls.device)
    g_cls, g_reg = torch.zeros_like(cls), torch.zeros_like(reg)
    _lib.call("dadet_fast_rcnn_loss_rows", _p(cls), _p(reg), R, cls.shape[1], reg.shape[1],
              _p(loss_labels.contiguous().to(torch.int64)), _p(_dev(regression_targets, "targets").contiguous()),
              _p(losses), _p(g_cls), _p(g_reg), _stream())
    return losses, g_cls, g_reg


SAMPLE_ROIS_MAX = 4096
PROPOSALS_SAMPLE_MAX_GT = 1024      # ground-truth boxes per image dadet_proposals_sample holds in LDS (csrc/sampling.hip)


def sample_rois_buffers(rows, device):
    """output buffers of sample_rois for `rows` rows (several images can share one set, each writing its own slice)"""
    return dict(idx=torch.empty(rows, dtype=torch.int64, device=device),
                objectness=torch.empty(rows, dtype=torch.float32, device=device),
                boxes=torch.empty((rows, 4), dtype=torch.float32, device=device),
                labels=torch.empty(rows, dtype=torch.int64, device=device),
                regression_targets=torch.empty((rows, 4), dtype=torch.float32, device=device),
                loss_labels=torch.empty(rows, dtype=torch.int64, device=device),
                domain=torch.empty(rows, dtype=torch.bool, device=device))


def sample_rois(boxes, labels, regression_targets, cap, max_pos, seed, is_source, counts_out, out=None):
    """one image's box-head sample in one launch (dadet_sample_rois) -> dict(idx, boxes, labels, regression_targets,
    loss_labels, domain) of `cap` rows each; counts_out: int32 [2] device tensor receiving (rows taken, positives).
    labels / regression_targets None = every proposal is a negative with zero targets (target-domain images).
    `out`: contiguous `cap`-row slices of sample_rois_buffers to write into."""
    _dev(boxes, "boxes")
    boxes = boxes.contiguous()
    n, dev = boxes.shape[0], boxes.device
    if labels is not None:
        labels = labels.contiguous()
        assert labels.dtype == torch.int64 and labels.numel() == n
    if regression_targets is not None:
        regression_targets = _dev(regression_targets, "regression_targets").contiguous()
    assert counts_out.dtype == torch.int32 and counts_out.numel() == 2 and counts_out.is_contiguous()
    if out is None:
        out = sample_rois_buffers(cap, dev)
    assert all(v.shape[0] == cap and v.is_contiguous() for v in out.values())
    _lib.call("dadet_sample_rois", _p(boxes), _p(labels), _p(regression_targets), n, int(cap), int(max_pos),
              ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), 1 if is_source else 0, _p(out["idx"]), _p(out["boxes"]),
              _p(out["labels"]), _p(out["regression_targets"]), _p(out["loss_labels"]), _p(out["domain"]),
              _p(counts_out), _stream())
    return out


def proposals_sample(pending, gt_boxes, gt_labels, high, low, weights, cap, max_pos, seed, is_source, counts_out,
                     out=None):
    """box-head sample of one image straight from the RPN's NMS result on the device (dadet_proposals_sample).
    pending: structures.bounding_box.PendingProposals.pending (sorted_boxes, sorted_scores, keep, count_dev, post_n, gt).
    -> (out dict as sample_rois + "objectness", prop_boxes [post_n + G', 4], prop_scores, n_props int32 [1])"""
    sb, ss = pending["sorted_boxes"].contiguous(), pending["sorted_scores"].contiguous()
    keep, count = pending["keep"], pending["count_dev"]
    post_n = int(pending["post_n"])
    app = pending["gt"].bbox.contiguous() if pending["gt"] is not None else None
    n_app = int(app.shape[0]) if app is not None else 0
    dev = sb.device
    if out is None:
        out = sample_rois_buffers(cap, dev)
    if "objectness" not in out:
        out["objectness"] = torch.empty(cap, dtype=torch.float32, device=dev)
    G = int(gt_boxes.shape[0]) if gt_boxes is not None else 0
    prop_boxes = torch.empty((post_n + n_app, 4), dtype=torch.float32, device=dev)
    prop_scores = torch.empty(post_n + n_app, dtype=torch.float32, device=dev)
    n_props = torch.empty(1, dtype=torch.int32, device=dev)
    wx, wy, ww, wh = weights
    _lib.call("dadet_proposals_sample", _p(sb), _p(ss), _p(keep), _p(count), post_n, _p(app), n_app,
              _p(gt_boxes.contiguous()) if G else None, _p(gt_labels.contiguous()) if G else None, G, float(high),
              float(low), float(wx), float(wy), float(ww), float(wh), int(cap), int(max_pos),
              ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), 1 if is_source else 0, _p(prop_boxes), _p(prop_scores),
              _p(n_props), _p(out["idx"]), _p(out["boxes"]), _p(out["labels"]), _p(out["regression_targets"]),
              _p(out["loss_labels"]), _p(out["domain"]), _p(out["objectness"]), _p(counts_out), _stream())
    return out, prop_boxes, prop_scores, n_props


SAMPLE_ANCHORS_MAX_CAP = 1024


def sample_anchors(labels, regression_targets, cap, max_pos, seed, index_offset, counts_out, out=None):
    """one image's RPN anchor sample in one launch (dadet_sample_anchors) -> dict(pos, neg: int64 [cap] anchor indices
    + index_offset, ascending; regression_targets_pos [cap,4]); counts_out: int32 [2] device tensor receiving
    (positives, negatives).  `out`: cap-row slices of buffers shared by several images."""
    _dev(labels, "labels"), _dev(regression_targets, "regression_targets")
    labels, regression_targets = labels.contiguous(), regression_targets.contiguous()
    A, dev = labels.numel(), labels.device
    assert regression_targets.numel() == 4 * A and 0 < cap <= SAMPLE_ANCHORS_MAX_CAP
    assert counts_out.dtype == torch.int32 and counts_out.numel() == 2 and counts_out.is_contiguous()
    if out is None:
        out = dict(pos=torch.empty(cap, dtype=torch.int64, device=dev),
                   neg=torch.empty(cap, dtype=torch.int64, device=dev),
                   regression_targets_pos=torch.empty((cap, 4), dtype=torch.float32, device=dev))
    assert all(v.shape[0] == cap and v.is_contiguous() for v in out.values())
    _lib.call("dadet_sample_anchors", _p(labels), _p(regression_targets), A, int(cap), int(max_pos),
              ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), ctypes.c_int64(int(index_offset)), _p(out["pos"]),
              _p(out["neg"]), _p(out["regression_targets_pos"]), _p(counts_out), _stream())
    return out


TOPK_SORTED_MAX = 16384


def topk_sorted(scores, k):
    """scores [rows, n] -> (values [rows, k], indices [rows, k] int64): the k largest per row, descending, equal scores
    by ascending index — torch.sort(scores, dim=1, descending=True, stable=True) cut to k columns, in ONE launch
    (dadet_topk_sorted)"""
    _dev(scores, "scores")
    assert scores.dim() == 2 and scores.stride(1) == 1 and 0 < k <= min(scores.shape[1], TOPK_SORTED_MAX)
    rows, n = scores.shape
    vals = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    _lib.call("dadet_topk_sorted", _p(scores), rows, n, ctypes.c_int64(scores.stride(0)), int(k), _p(vals), _p(idx),
              _stream())
    return vals, idx


TOPK_ROWS_MAX = 16


def topk_sorted_rows(score_rows, ks):
    """sorted top-k of several 1-D score tensors of different lengths in one call (dadet_topk_sorted_rows: five launches
    for all of them) -> [(scores [k_i] descending, indices [k_i] int64, ties by ascending index)]"""
    assert 0 < len(score_rows) <= TOPK_ROWS_MAX and len(ks) == len(score_rows)
    dev = score_rows[0].device
    rows, outs = (_lib.TopkRow * len(score_rows))(), []
    for i, (s, k) in enumerate(zip(score_rows, ks)):
        _dev(s, "scores")
        assert s.dim() == 1 and s.is_contiguous() and 0 < k <= min(s.numel(), TOPK_SORTED_MAX)
        o_s = torch.empty(k, dtype=torch.float32, device=dev)
        o_i = torch.empty(k, dtype=torch.int64, device=dev)
        rows[i] = _lib.TopkRow(s.data_ptr(), o_s.data_ptr(), o_i.data_ptr(), s.numel(), int(k))
        outs.append((o_s, o_i))
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_topk_sorted_rows_workspace_bytes", len(score_rows), int(max(ks)), ctypes.byref(nbytes))
    ws = _workspace(nbytes.value, dev)
    _lib.call("dadet_topk_sorted_rows", rows, len(score_rows), _p(ws), ctypes.c_size_t(ws.numel()), _stream())
    return outs


def rpn_anchor_targets(anchors, visible, gt_boxes, high_threshold, low_threshold):
    """-> (labels float [A] in {1, 0, -1}, regression_targets [A,4]); see dadet_rpn_anchor_targets"""
    _dev(anchors, "anchors"), _dev(gt_boxes, "gt_boxes")
    anchors = anchors.contiguous().float()
    gt_boxes = gt_boxes.contiguous().float()
    vis = visible.contiguous().to(torch.uint8) if visible.dtype != torch.bool else visible.contiguous().view(torch.uint8)
    A, G = anchors.shape[0], gt_boxes.shape[0]
    ws = torch.empty(G, dtype=torch.int32, device=anchors.device)
    labels = torch.empty(A, dtype=torch.float32, device=anchors.device)
    targets = torch.empty((A, 4), dtype=torch.float32, device=anchors.device)
    _lib.call("dadet_rpn_anchor_targets", _p(anchors), _p(vis), A, _p(gt_boxes), G, float(high_threshold),
              float(low_threshold), _p(ws), _p(labels), _p(targets), _stream())
    return labels, targets


def box_match_encode(proposals, gt_boxes, gt_labels, high_threshold, low_threshold, weights):
    """-> (matched_idxs int64 [P], labels int64 [P], regression_targets [P,4]); see dadet_box_match_encode"""
    _dev(proposals, "proposals"), _dev(gt_boxes, "gt_boxes")
    proposals = proposals.contiguous().float()
    gt_boxes = gt_boxes.contiguous().float()
    gt_labels = gt_labels.contiguous().to(torch.int64)
    P, G = proposals.shape[0], gt_boxes.shape[0]
    matched = torch.empty(P, dtype=torch.int64, device=proposals.device)
    labels = torch.empty(P, dtype=torch.int64, device=proposals.device)
    targets = torch.empty((P, 4), dtype=torch.float32, device=proposals.device)
    wx, wy, ww, wh = [float(v) for v in weights]
    _lib.call("dadet_box_match_encode", _p(proposals), P, _p(gt_boxes), _p(gt_labels), G, float(high_threshold),
              float(low_threshold), wx, wy, ww, wh, _p(matched), _p(labels), _p(targets), _stream())
    return matched, labels, targets


def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    """-> (output, argmax int32) [R, C, ph, pw] channels_last; reference ROIPool.h:11-24"""
    _dev(input, "input"), _dev(rois, "rois")
    x = _nhwc(input)
    rois = rois.contiguous().float()
    B, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.float32, device=x.device, memory_format=CL)
    arg = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.int32, device=x.device, memory_format=CL)
    _lib.call("dadet_roi_pool_forward", _p(x), _p(rois), _p(out), _p(arg), B, C, H, W, R, pooled_height,
              pooled_width, float(spatial_scale), _stream())
    return out, arg


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                      height, width):
    """argument order of ROIPool.h:26-46 (`input` and `spatial_scale` are unused, as in the reference kernel)"""
    _dev(grad, "grad")
    grad = _nhwc(grad)
    argmax = argmax.contiguous(memory_format=CL)
    rois = rois.contiguous().float()
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device,
                      memory_format=CL)
    _lib.call("dadet_roi_pool_backward", _p(grad), _p(argmax), _p(rois), _p(gin), batch_size, channels, height, width,
              rois.shape[0], pooled_height, pooled_width, _stream())
    return gin


def _psroi_args(data, rois, offset, out_size, out_channels, no_trans, group_size, part_size, sample_per_part):
    B, C, H, W = data.shape
    R = rois.shape[0]
    num_classes = 1 if no_trans else offset.shape[1] // 2
    return B, H, W, C, R, num_classes


def deform_psroi_pool_forward(data, rois, offset, no_trans, spatial_scale, out_channels, group_size, out_size,
                              part_size, sample_per_part, trans_std):
    """data [B, out_channels*group_size^2, H, W] (channels_last), rois [R,5], offset [R, 2*num_classes, part, part]
    (ignored with no_trans) -> (output, output_count) [R, out_channels, out_size, out_size] channels_last.
    Reference: tools/cityscapes/maskrcnn_benchmark/layers/dcn/deform_pool_func.py:30-35."""
    _dev(data, "data"), _dev(rois, "rois")
    data = _nhwc(data)
    rois = rois.contiguous().float()
    offset = None if no_trans else offset.contiguous().float()
    B, H, W, C, R, ncls = _psroi_args(data, rois, offset, out_size, out_channels, no_trans, group_size, part_size,
                                      sample_per_part)
    out = torch.empty((R, out_channels, out_size, out_size), dtype=torch.float32, device=data.device,
                      memory_format=CL)
    cnt = torch.empty_like(out)
    _lib.call("dadet_deform_psroi_pool_forward", _p(data), _p(rois), _p(offset), _p(out), _p(cnt), B, H, W, C, R,
              int(bool(no_trans)), float(spatial_scale), out_channels, group_size, out_size, part_size,
              sample_per_part, float(trans_std), ncls, _stream())
    return out, cnt


def deform_psroi_pool_backward(grad_out, data, rois, offset, count, no_trans, spatial_scale, out_channels,
                               group_size, out_size, part_size, sample_per_part, trans_std):
    """-> (grad_data channels_last, grad_offset | None); reference deform_pool_func.py:52-60"""
    data, grad_out, count = _nhwc(data), _nhwc(grad_out), _nhwc(count)
    rois = rois.contiguous().float()
    offset = None if no_trans else offset.contiguous().float()
    B, H, W, C, R, ncls = _psroi_args(data, rois, offset, out_size, out_channels, no_trans, group_size, part_size,
                                      sample_per_part)
    gdata = torch.empty_like(data).zero_()
    goffset = None if no_trans else torch.zeros_like(offset)
    _lib.call("dadet_deform_psroi_pool_backward", _p(grad_out), _p(count), _p(data), _p(rois), _p(offset), _p(gdata),
              _p(goffset), B, H, W, C, R, int(bool(no_trans)), float(spatial_scale), out_channels, group_size,
              out_size, part_size, sample_per_part, float(trans_std), ncls, _stream())
    return gdata, goffset


# ---- the vendored tree's deformable entry points, by the reference's names and signatures ---------------------------------
# tools/cityscapes/maskrcnn_benchmark/csrc/vision.cpp:17-23 (= csrc/deform_conv.h, csrc/deform_pool.h; CUDA side
# csrc/cuda/vision.h:61-113), called from layers/dcn/deform_conv_func.py:49,87,110,182,216 and deform_pool_func.py:41,81 with
# CALLER-ALLOCATED results (`output`, `grad_*`) and scratch (`columns`, `ones`).  Composed from the sampling kernels
# (csrc/deform.hip) and the GEMM entry points: the reference's Python keeps working with only the native module replaced
# (option B).  `columns` / `ones` are accepted and left alone — the reference rebinds its by-value copies
# (`columns = at::zeros(...)`, deform_conv_cuda.cu:197), the caller's tensors never change there either.  groups == 1 and
# equal vertical / horizontal stride, padding, dilation only (every DCN config of the reference); anything else raises.
def _dc_geometry(what, kw, kh, dw, dh, pw, ph, lw, lh, group, weight):
    if group != 1:
        raise NotImplementedError("%s: groups > 1 is not on the HIP path" % what)
    if dw != dh or pw != ph or lw != lh:
        raise NotImplementedError("%s: stride / padding / dilation must be the same in both dimensions" % what)
    if tuple(weight.shape[2:]) != (kh, kw):
        raise ValueError("%s: kernel %dx%d does not match the weight %s" % (what, kh, kw, tuple(weight.shape)))
    return int(dw), int(pw), int(lw)


def _w_as_1x1(weight):
    """[Cout,Cin,kh,kw] -> [Cout, kh*kw*Cin, 1, 1] with K = (tap, ci), the column order of the sampling kernels"""
    cout, cin, kh, kw = weight.shape
    return weight.contiguous(memory_format=CL).permute(0, 2, 3, 1).reshape(cout, kh * kw * cin, 1, 1)


def _into(dst, src, what):
    if tuple(dst.shape) != tuple(src.shape):
        raise ValueError("%s: caller's tensor is %s, the result %s" % (what, tuple(dst.shape), tuple(src.shape)))
    dst.copy_(src)


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH,
                        group, deformable_group, im2col_step):
    """DCNv1 forward into `output` [N,Cout,Ho,Wo] (deform_conv.h:10-43, deform_conv_cuda.cu:158-242) -> 1"""
    stride, pad, dil = _dc_geometry("deform_conv_forward", kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, weight)
    cols = deform_sample_forward(input, offset, None, kH, kW, stride, pad, dil, deformable_group)
    _into(output, conv_forward(cols, _w_as_1x1(weight)), "deform_conv_forward: output")
    return 1


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW, padH,
                               dilationW, dilationH, group, deformable_group, im2col_step):
    """DCNv1 gradients w.r.t. the sampled map and the offsets into `gradInput`, `gradOffset` (deform_conv.h:46-83) -> 1"""
    stride, pad, dil = _dc_geometry("deform_conv_backward_input", kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                                    weight)
    gcols = conv_forward(_nhwc(gradOutput), conv_weight_transpose(_w_as_1x1(weight)))
    gx, goffset, _ = deform_sample_backward(input, offset, None, gcols, kH, kW, stride, pad, dil, deformable_group)
    _into(gradInput, gx, "deform_conv_backward_input: gradInput")
    _into(gradOffset, goffset, "deform_conv_backward_input: gradOffset")
    return 1


def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                    dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """gradWeight += scale * (gradOutput x columns^T) (deform_conv.h:86-124; the reference's addmm_ accumulates too) -> 1"""
    if group != 1 or dW != dH or padW != padH or dilationW != dilationH:
        raise NotImplementedError("deform_conv_backward_parameters: groups == 1 and equal stride / padding / dilation only")
    cout, cin = gradWeight.shape[0], gradWeight.shape[1]
    cols = deform_sample_forward(input, offset, None, kH, kW, int(dW), int(padW), int(dilationW), deformable_group)
    gw = conv_wgrad(cols, _nhwc(gradOutput), (cout, kH * kW * cin, 1, 1))
    gradWeight.add_(gw.reshape(cout, kH, kW, cin).permute(0, 3, 1, 2), alpha=float(scale))
    return 1


def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h,
                                  stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    """DCNv2 forward into `output` (deform_conv.h:127-159, deform_conv_cuda.cu:489-573); `bias` is read only with_bias"""
    stride, pad, dil = _dc_geometry("modulated_deform_conv_forward", kernel_w, kernel_h, stride_w, stride_h, pad_w, pad_h,
                                    dilation_w, dilation_h, group, weight)
    cols = deform_sample_forward(input, offset, mask, kernel_h, kernel_w, stride, pad, dil, deformable_group)
    _into(output, conv_forward(cols, _w_as_1x1(weight), None, bias if with_bias else None),
          "modulated_deform_conv_forward: output")


def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight, grad_bias,
                                   grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                   dilation_h, dilation_w, group, deformable_group, with_bias):
    """DCNv2 backward (deform_conv.h:162-201, deform_conv_cuda.cu:575-691): grad_input / grad_offset / grad_mask are written,
    grad_weight and (with_bias) grad_bias accumulated, as the reference's addmm_ / addmv_ do on the caller's zeros"""
    stride, pad, dil = _dc_geometry("modulated_deform_conv_backward", kernel_w, kernel_h, stride_w, stride_h, pad_w, pad_h,
                                    dilation_w, dilation_h, group, weight)
    cout, cin = weight.shape[0], weight.shape[1]
    gy = _nhwc(grad_output)
    w1 = _w_as_1x1(weight)
    gcols = conv_forward(gy, conv_weight_transpose(w1))
    gx, goffset, gmask = deform_sample_backward(input, offset, mask, gcols, kernel_h, kernel_w, stride, pad, dil,
                                                deformable_group)
    _into(grad_input, gx, "modulated_deform_conv_backward: grad_input")
    _into(grad_offset, goffset, "modulated_deform_conv_backward: grad_offset")
    _into(grad_mask, gmask, "modulated_deform_conv_backward: grad_mask")
    cols = deform_sample_forward(input, offset, mask, kernel_h, kernel_w, stride, pad, dil, deformable_group)
    gw = conv_wgrad(cols, gy, (cout, kernel_h * kernel_w * cin, 1, 1))
    grad_weight.add_(gw.reshape(cout, kernel_h, kernel_w, cin).permute(0, 3, 1, 2))
    if with_bias:
        grad_bias.add_(colsum(gy))


def deform_psroi_pooling_forward(input, bbox, trans, out, top_count, no_trans, spatial_scale, output_dim, group_size,
                                 pooled_size, part_size, sample_per_part, trans_std):
    """position-sensitive deformable ROI pooling into `out`, `top_count` [R, output_dim, pooled, pooled]
    (deform_pool.h:10-35; called from deform_pool_func.py:41-55)"""
    o, c = deform_psroi_pool_forward(input, bbox, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size,
                                     part_size, sample_per_part, trans_std)
    _into(out, o, "deform_psroi_pooling_forward: out")
    _into(top_count, c, "deform_psroi_pooling_forward: top_count")


def deform_psroi_pooling_backward(out_grad, input, bbox, trans, top_count, input_grad, trans_grad, no_trans, spatial_scale,
                                  output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std):
    """gradients into `input_grad` and (unless no_trans) `trans_grad` (deform_pool.h:38-66; deform_pool_func.py:81-97)"""
    gd, go = deform_psroi_pool_backward(out_grad, input, bbox, trans, top_count, no_trans, spatial_scale, output_dim,
                                        group_size, pooled_size, part_size, sample_per_part, trans_std)
    _into(input_grad, gd, "deform_psroi_pooling_backward: input_grad")
    if go is not None:
        _into(trans_grad, go, "deform_psroi_pooling_backward: trans_grad")


def check_nonfinite():
    """Contraction mode 4's guard (include/dadet.h: dadet_nonfinite_poll): raises FloatingPointError naming the first GEMM
    launch whose sums were non-finite since the last call — an operand's largest-magnitude slot lay below its data (a stale
    slot, a `carry` across an operation that can raise the maximum), which overflows fp16 in the operand split.  A
    synchronising read: the trainer calls it at its logging period, next to the reference's NaN test."""
    buf = ctypes.create_string_buffer(512)
    n = _lib.load().dadet_nonfinite_poll(buf, 512)
    if n < 0:
        raise _lib.DadetError("dadet_nonfinite_poll failed")
    if n:
        raise FloatingPointError(buf.value.decode())


def set_gemm_mode(mode):
    """0 exact fp32 MFMA | 3 three-term bf16 split (fp32-class accuracy) | 2 two-term split; see include/dadet.h"""
    global _MODE
    _lib.call("dadet_set_gemm_mode", int(mode))
    _MODE = int(mode)
    _KNAME_CACHE.clear()


def get_gemm_mode():
    return _lib.load().dadet_get_gemm_mode()


def apply_env_gemm_mode():
    """DADET_GEMM_MODE = 0 | 2 | 3 selects the contraction mode at start-up (default: 3, the fp32-accurate
    three-term split; 0 = exact fp32 MFMA)."""
    import os

    set_gemm_mode(int(os.environ.get("DADET_GEMM_MODE", _lib.DEFAULT_GEMM_MODE)))


def device_info():
    cu, khz, hbm = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_size_t(0)
    name = ctypes.create_string_buffer(64)
    _lib.call("dadet_device_info", ctypes.byref(cu), ctypes.byref(khz), ctypes.byref(hbm), name, 64)
    return {"cu_count": cu.value, "clock_khz": khz.value, "hbm_bytes": hbm.value,
            "arch": name.value.decode()}
