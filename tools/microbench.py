"""Per-kernel micro-benchmarks at the BASELINE shapes (R-50-C4, 2 x 1024x2048 images, 512 ROIs).
Prints one line per kernel with achieved TFLOP/s (fp32 MFMA peak 157.3) or GB/s (HBM peak 8000)."""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C  # noqa: E402

CL = torch.channels_last
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def conv_case(name, N, Cin, H, W, Cout, k, stride, pad):
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.05).contiguous(memory_format=CL)
    sc = torch.rand(Cout, device=dev) + 0.5
    bi = torch.randn(Cout, device=dev)
    Ho, Wo = _C.conv_out_size(H, W, k, k, stride, pad)
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    y = _C.conv_forward(x, w, sc, bi, stride=stride, pad=pad, relu_mode=1)
    ms = timeit(lambda: _C.conv_forward(x, w, sc, bi, stride=stride, pad=pad, relu_mode=1, out=y))
    gy = torch.randn_like(y)
    msw = timeit(lambda: _C.conv_wgrad(x, gy, tuple(w.shape), stride, pad))
    print("%-28s M=%7d K=%5d N=%4d  fwd %7.3f ms %6.1f TF/s | wgrad %7.3f ms %6.1f TF/s" %
          (name, N * Ho * Wo, Cin * k * k, Cout, ms, flops / ms / 1e9, msw, flops / msw / 1e9), flush=True)


def main():
    print(_C.device_info())
    if len(sys.argv) > 1:
        _C.set_gemm_mode(int(sys.argv[1]))
        print("gemm mode", _C.get_gemm_mode())
    conv_case("layer1 1x1 64->256", 2, 64, 256, 512, 256, 1, 1, 0)
    conv_case("layer1 3x3 64->64", 2, 64, 256, 512, 64, 3, 1, 1)
    conv_case("layer2 1x1 256->128 s2", 2, 256, 256, 512, 128, 1, 2, 0)
    conv_case("layer2 3x3 128->128", 2, 128, 128, 256, 128, 3, 1, 1)
    conv_case("layer2 1x1 128->512", 2, 128, 128, 256, 512, 1, 1, 0)
    conv_case("layer3 3x3 256->256", 2, 256, 64, 128, 256, 3, 1, 1)
    conv_case("layer3 1x1 256->1024", 2, 256, 64, 128, 1024, 1, 1, 0)
    conv_case("layer3 1x1 1024->256", 2, 1024, 64, 128, 256, 1, 1, 0)
    conv_case("rpn 3x3 1024->1024", 2, 1024, 64, 128, 1024, 3, 1, 1)
    conv_case("res5 1x1 1024->512 s2", 512, 1024, 14, 14, 512, 1, 2, 0)
    conv_case("res5 3x3 512->512", 512, 512, 7, 7, 512, 3, 1, 1)
    conv_case("res5 1x1 512->2048", 512, 512, 7, 7, 2048, 1, 1, 0)
    conv_case("res5 1x1 2048->512", 512, 2048, 7, 7, 512, 1, 1, 0)
    conv_case("da img 1x1 1024->512", 2, 1024, 64, 128, 512, 1, 1, 0)
    conv_case("fc 2048->1024 (512 rows)", 512, 2048, 1, 1, 1024, 1, 1, 0)
    # stem
    x = torch.randn((2, 3, 1024, 2048), device=dev)
    x4 = _C.nchw3_to_nhwc4(x)
    w4 = torch.zeros((64, 4, 7, 8), device=dev)
    w4[:, :3, :, :7] = torch.randn((64, 3, 7, 7), device=dev) * 0.05
    w4 = w4.contiguous(memory_format=CL)
    ms = timeit(lambda: _C.conv_forward(x4, w4, stride=2, pad=3, relu_mode=1, out_size=(512, 1024)))
    print("stem 7x7 s2 3->64: %.3f ms (%.1f TF/s algorithmic 147-tap)" % (ms, 2.0 * 2 * 512 * 1024 * 64 * 147 / ms / 1e9))
    y = _C.conv_forward(x4, w4, stride=2, pad=3, relu_mode=1, out_size=(512, 1024))
    ms = timeit(lambda: _C.maxpool3x3s2(y))
    print("maxpool: %.3f ms  %.0f GB/s" % (ms, (y.numel() * 4 * 1.25) / ms / 1e6))
    # ROIAlign
    feat = torch.randn((2, 1024, 64, 128), device=dev).contiguous(memory_format=CL)
    g = torch.Generator(device="cpu").manual_seed(0)
    xy = torch.rand((512, 2), generator=g) * torch.tensor([1800.0, 900.0])
    wh = torch.rand((512, 2), generator=g) * 300 + 16
    rois = torch.cat([(torch.arange(512) % 2).float().view(-1, 1), xy, xy + wh], 1).to(dev)
    ms = timeit(lambda: _C.roi_align_forward(feat, rois, 1 / 16.0, 14, 14, 0))
    out_bytes = 512 * 1024 * 196 * 4
    print("roi_align fwd: %.3f ms  %.0f GB/s (algorithmic %.0f MB)" % (ms, (out_bytes + feat.numel() * 4) / ms / 1e6, (out_bytes + feat.numel() * 4) / 1e6))
    go = torch.randn((512, 1024, 14, 14), device=dev).contiguous(memory_format=CL)
    ms = timeit(lambda: _C.roi_align_backward(go, rois, 1 / 16.0, 14, 14, 2, 1024, 64, 128, 0))
    print("roi_align bwd (gather): %.3f ms  %.0f GB/s" % (ms, (out_bytes + feat.numel() * 4) / ms / 1e6))
    ms = timeit(lambda: _C.roi_align_backward(go, rois, 1 / 16.0, 14, 14, 2, 1024, 64, 128, 0, atomic=True))
    print("roi_align bwd (atomic): %.3f ms  %.0f GB/s" % (ms, (out_bytes + 2 * feat.numel() * 4) / ms / 1e6))
    # NMS
    n = 12000
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1900.0, 950.0])
    wh = torch.rand((n, 2), generator=g) * 200 + 8
    boxes = torch.cat([xy, xy + wh], 1).to(dev)
    scores = torch.rand(n, generator=g).to(dev)
    ms = timeit(lambda: _C.nms_with_count(boxes, scores, 0.7, max_keep=2000))
    k, c = _C.nms_with_count(boxes, scores, 0.7, max_keep=2000)
    print("nms n=12000 thr .7 max_keep 2000: %.3f ms (kept %d)" % (ms, int(c)))
    ms = timeit(lambda: _C.nms_with_count(boxes, scores, 0.7))
    k, c = _C.nms_with_count(boxes, scores, 0.7)
    print("nms n=12000 thr .7 unlimited: %.3f ms (kept %d)" % (ms, int(c)))
    # weight transpose
    w = torch.randn((1024, 1024, 3, 3), device=dev).contiguous(memory_format=CL)
    ms = timeit(lambda: _C.conv_weight_transpose(w))
    print("weight transpose 1024x1024x3x3: %.3f ms %.0f GB/s" % (ms, 2 * w.numel() * 4 / ms / 1e6))


if __name__ == "__main__":
    main()


def image_prep():
    """device-side Resize + flip + BGR-255 + normalise + pad of one 1024x2048 image to the DA yamls' 600x1200"""
    import numpy as np
    from da_detect_amd.config import cfg
    from da_detect_amd.data.device_prep import DeviceBatchPreparer

    c = cfg.clone()
    c.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (600,), "INPUT.MAX_SIZE_TRAIN", 1200, "DATALOADER.SIZE_DIVISIBILITY", 32])
    prep = DeviceBatchPreparer(c, is_train=True)
    img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1024, 2048, 3), dtype=np.uint8)).to(dev)
    ms = timeit(lambda: prep([img], None, [((600, 1200), True)]))
    print("image prep 1024x2048 -> 600x1200 (+pad 608x1216): %.3f ms  (%.0f MB/s of input pixels)" % (
        ms, 1024 * 2048 * 3 / ms / 1e3))
    try:
        import time
        from PIL import Image
        pil = Image.fromarray(img.cpu().numpy())
        t0 = time.perf_counter()
        for _ in range(5):
            r = np.asarray(pil.resize((1200, 600), Image.BILINEAR), dtype=np.float32)
            r = (r[:, ::-1, ::-1] / 255.0 * 255.0 - np.asarray(c.INPUT.PIXEL_MEAN, np.float32))
        print("  host chain (Pillow resize + numpy normalise, 1 core): %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    except ImportError:
        pass


if __name__ == "__main__" and os.environ.get("DADET_MICROBENCH_IMAGE", "1") == "1":
    image_prep()
