#!/usr/bin/env python
"""EXPERIMENT: forward GEMM fed with bf16 term planes prepared once in HBM (csrc/conv_planes.hip) vs the production
kernel that splits fp32 operands while staging them.  Prints both times and the cost of preparing the planes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C, _lib  # noqa: E402
from da_detect_amd._lib import ConvDesc  # noqa: E402

CL = torch.channels_last
dev = torch.device("cuda:0")
lib = _lib.load()
P = ctypes.c_void_p


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def planes_of(t):
    flat = t.contiguous(memory_format=CL) if t.dim() == 4 else t.contiguous()
    out = torch.empty((3, flat.numel()), dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dadet_split_planes(P(flat.data_ptr()), P(out.data_ptr()), ctypes.c_int64(flat.numel()), st)
    assert rc == 0
    return flat, out


def case(name, N, Cin, H, W, Cout, k, pad):
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.05).contiguous(memory_format=CL)
    y_ref = _C.conv_forward(x, w, pad=pad)
    xf, xp = planes_of(x)
    wf, wp = planes_of(w)
    y = torch.empty_like(y_ref)
    d = _C._desc(N, H, W, Cin, Cout, k, k, 1, pad, H, W, H, W, 1, 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = lib.dadet_conv_forward_planes_probe(ctypes.byref(d), P(xp.data_ptr()), P(wp.data_ptr()), P(y.data_ptr()), st)
        assert rc == 0

    run()
    torch.cuda.synchronize()
    err = float((y - y_ref).abs().max()) / float(y_ref.abs().max())
    t_planes = timeit(run)
    t_prod = timeit(lambda: _C.conv_forward(x, w, pad=pad, out=y_ref))
    t_split = timeit(lambda: lib.dadet_split_planes(P(xf.data_ptr()), P(xp.data_ptr()), ctypes.c_int64(xf.numel()), st))
    print("%-24s production %.3f ms | planes kernel %.3f ms (%.1f%%) | split_planes(x) %.3f ms | rel err %.1e" % (
        name, t_prod, t_planes, 100.0 * (t_planes / t_prod - 1.0), t_split, err))


lib.dadet_split_planes.argtypes = [P, P, ctypes.c_int64, P]
lib.dadet_conv_forward_planes_probe.argtypes = [ctypes.POINTER(ConvDesc), P, P, P, P]
case("rpn 3x3 1024->1024", 2, 1024, 64, 128, 1024, 3, 1)
case("res5 3x3 512->512", 512, 512, 7, 7, 512, 3, 1)
case("res5 1x1 512->2048", 512, 512, 7, 7, 2048, 1, 0)
case("layer3 1x1 1024->256", 2, 1024, 64, 128, 256, 1, 0)
