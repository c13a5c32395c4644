"""Per-output-channel accuracy of contraction mode 4 on the REAL operands of one training step.

Every forward / data-gradient / weight-gradient GEMM of one step of a recipe is captured with its operands (activations,
weights, output gradients as the step produced them: heavy-tailed, with dead and nearly dead channels), then replayed as
a bare contraction in mode 0 (exact fp32 MFMA) and mode 4 (two fp16 terms under a per-tensor power-of-two scale) and
compared with a float64 contraction PER OUTPUT CHANNEL:
    err_m[c] = || y_m[:, c] - y64[:, c] ||_2 / || y64[:, c] ||_2
The per-tensor scale keeps full relative accuracy for elements within 2^-16 of the tensor's largest; an output channel
fed only by smaller entries (a weight row, a gradient channel far below the maximum) is where mode 4 could be worse than
fp32.  Reported: for every GEMM the worst err_4 / err_0 over channels, how far below the operand's maximum the channel's
own operand entries lie (log2), and a histogram of that distance over all operand channels of the step.

usage (GPU box): python tools/probes/real_operand_error.py [--workload da] [--hw 1024x2048] [--out profiles/...txt]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd import _C  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import train_step  # noqa: E402

CL = torch.channels_last


def capture(workload, hw, steps):
    device = torch.device("cuda", 0)
    yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[workload]
    c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
    h, w = hw
    images, targets = make_batch(c, images_per_gpu, h, w, seed=100, device=device)
    for _ in range(steps):                      # a few optimizer steps first: not the init-time statistics
        train_step(model, opt, images, targets)
    fwd, wg = [], []
    orig_f, orig_w = _C.conv_forward, _C.conv_wgrad

    def conv_forward(x, w_, scale=None, bias=None, addend=None, mask_ref=None, stride=1, pad=0, relu_mode=0, out=None,
                     out_spatial_stride=1, out_hw=None, out_size=None):
        if out_spatial_stride == 1 and out_size is None and x.shape[1] % 4 == 0:
            fwd.append((x.detach().clone(), w_.detach().clone(), stride, pad))
        return orig_f(x, w_, scale, bias, addend, mask_ref, stride, pad, relu_mode, out, out_spatial_stride, out_hw, out_size)

    def conv_wgrad(x, gy, weight_shape, stride=1, pad=0, out_scale=None, dw=None, accumulate=False, pending=None):
        if gy.shape[1] == weight_shape[0]:
            wg.append((x.detach().clone(), gy.detach().clone(), tuple(weight_shape), stride, pad))
        return orig_w(x, gy, weight_shape, stride, pad, out_scale, dw, accumulate, pending)

    orig_g = _C.conv_wgrad_group

    def conv_wgrad_group(requests, pending):      # a block's weight gradients in one launch: the same operands, per layer
        for r in requests:
            if r["gy"].shape[1] == r["weight_shape"][0]:
                wg.append((r["x"].detach().clone(), r["gy"].detach().clone(), tuple(r["weight_shape"]), r.get("stride", 1),
                           r.get("pad", 0)))
        return orig_g(requests, pending)

    _C.conv_forward, _C.conv_wgrad, _C.conv_wgrad_group = conv_forward, conv_wgrad, conv_wgrad_group
    try:
        train_step(model, opt, images, targets)
        torch.cuda.synchronize()
    finally:
        _C.conv_forward, _C.conv_wgrad, _C.conv_wgrad_group = orig_f, orig_w, orig_g
    return fwd, wg


def per_channel(y, ref, cdim):
    dims = [d for d in range(ref.dim()) if d != cdim]
    num = (y.double() - ref).pow(2).sum(dim=dims).sqrt()
    den = ref.pow(2).sum(dim=dims).sqrt()
    return num, den


def below_max(t, cdim):
    """log2(max|t| / max|t[channel]|) per channel (inf for an all-zero channel)"""
    dims = [d for d in range(t.dim()) if d != cdim]
    cm = t.abs().amax(dim=dims).double()
    top = float(cm.max())
    return torch.where(cm > 0, torch.log2(top / cm), torch.full_like(cm, float("inf")))


def analyse(fwd, wg):
    """-> (rows, hist): one record per GEMM with its per-channel statistics; operand-channel depth histogram input"""
    hist = {}

    def add_hist(kind, d):
        hist.setdefault(kind, []).append(d[torch.isfinite(d)].cpu())

    rows = []
    for kind, items in (("fwd/dgrad", fwd), ("wgrad", wg)):
        for it in items:
            if kind == "fwd/dgrad":
                x, w_, stride, pad = it
                ref = torch.nn.functional.conv2d(x.double(), w_.double(), stride=stride, padding=pad)
                out = {}
                for mode in (0, 4):
                    _C.set_gemm_mode(mode)
                    out[mode] = _C.conv_forward(x, w_, stride=stride, pad=pad)
                cdim = 1
                dist = below_max(w_, 0)                      # the weight row that feeds output channel c
                add_hist("weight rows", dist)
                add_hist("activation channels", below_max(x, 1))
                shape = "M=%d N=%d K=%d k%d s%d" % (ref.shape[0] * ref.shape[2] * ref.shape[3], w_.shape[0],
                                                     w_.shape[1] * w_.shape[2] * w_.shape[3], w_.shape[2], stride)
            else:
                x, gy, wshape, stride, pad = it
                ref = torch.nn.grad.conv2d_weight(x.double(), wshape, gy.double(), stride=stride, padding=pad)
                out = {}
                for mode in (0, 4):
                    _C.set_gemm_mode(mode)
                    out[mode] = _C.conv_wgrad(x, gy, wshape, stride=stride, pad=pad)
                cdim = 0
                dist = below_max(gy, 1)                      # the gradient channel that feeds dW[c]
                add_hist("output-gradient channels", dist)
                shape = "M=%d N=%d K=%d k%d s%d" % (gy.shape[0] * gy.shape[2] * gy.shape[3], wshape[0],
                                                     wshape[1] * wshape[2] * wshape[3], wshape[2], stride)
            _C.set_gemm_mode(4)
            e0, den = per_channel(out[0], ref, cdim)
            e4, _ = per_channel(out[4], ref, cdim)
            ok = den > 0
            rows.append(dict(kind=kind, shape=shape, r0=(e0[ok] / den[ok]), r4=(e4[ok] / den[ok]), depth=dist[ok],
                             a4=e4[ok] / den.max(), a0=e0[ok] / den.max()))
            del out, ref
    return rows, hist


SHALLOW = 14.0      # binades below the operand's largest magnitude within which fp32's own relative accuracy is demanded


def verdicts(rec):
    """the two statements tests/test_real_operands_gpu.py asserts, per GEMM:
    rel   channels fed by operand entries within 2^-14 of the operand's largest: err_4 <= 2 x max(err_0, median err_0)
          (the median guards against a channel whose fp32 error happens to be far below its neighbours')
    abs   EVERY channel: || y_4[c] - y64[c] || <= 2^-22 x the largest channel norm of that output — what the per-tensor scale
          promises for entries far below the maximum (absolute, not relative, accuracy)"""
    r0, r4, depth = rec["r0"], rec["r4"], rec["depth"]
    floor = torch.clamp(torch.maximum(r0, r0.median()), min=2.0 ** -24)
    shallow = depth <= SHALLOW
    rel = float((r4 / floor)[shallow].max()) if bool(shallow.any()) else 0.0
    deep = ~shallow
    return dict(rel=rel, abs=float(rec["a4"].max()), abs0=float(rec["a0"].max()), n_deep=int(deep.sum()),
                deep_rel=float(r4[deep].max()) if bool(deep.any()) else 0.0,
                deep_rel0=float(r0[deep].max()) if bool(deep.any()) else 0.0,
                deepest=float(depth[torch.isfinite(depth)].max()) if bool(torch.isfinite(depth).any()) else 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="da")
    ap.add_argument("--hw", default="1024x2048")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    hw = [int(v) for v in args.hw.lower().split("x")]
    fwd, wg = capture(args.workload, hw, args.steps)
    rows, hist = analyse(fwd, wg)
    lines = ["# tools/probes/real_operand_error.py --workload %s --hw %s: %d forward / data-gradient and %d weight-gradient GEMMs "
             "of step %d, replayed as bare contractions in mode 0 (exact fp32 MFMA) and mode 4, against float64"
             % (args.workload, args.hw, len(fwd), len(wg), args.steps + 1),
             "# per output channel c: err_m[c] = ||y_m[:, c] - y64[:, c]|| / ||y64[:, c]||; depth = log2(operand max / max of the "
             "operand entries that feed channel c) — the weight row for forward / data gradient, the output-gradient channel for "
             "the weight gradient",
             "# rel = worst err_4 / max(err_0, median err_0) over channels of depth <= %g; abs = worst ||y_4[c] - y64[c]|| over ALL "
             "channels, in units of the largest channel norm of that output (fp32 resolution of it: 2^-24 = 6.0e-08)" % SHALLOW, ""]
    recs = []
    for r in rows:
        v = verdicts(r)
        recs.append((v["rel"], "%-10s %-34s ch %4d | err_0 med %.1e max %.1e | err_4 med %.1e max %.1e | rel %.2f | abs %.1e "
                               "(mode 0: %.1e) | %d channels deeper than %g binades (deepest %.1f): err_4 up to %.1e (mode 0: %.1e)"
                     % (r["kind"], r["shape"], r["r0"].numel(), float(r["r0"].median()), float(r["r0"].max()),
                        float(r["r4"].median()), float(r["r4"].max()), v["rel"], v["abs"], v["abs0"], v["n_deep"], SHALLOW,
                        v["deepest"], v["deep_rel"], v["deep_rel0"])))
    recs.sort(key=lambda t: -t[0])
    lines += [t[1] for t in recs]
    lines.append("")
    lines.append("## how far below their tensor's largest magnitude the operand channels of this step lie (log2, all GEMMs)")
    for kind, parts in hist.items():
        d = torch.cat(parts)
        edges = [0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 200]
        counts = [int(((d >= a) & (d < b)).sum()) for a, b in zip(edges[:-1], edges[1:])]
        lines.append("%-26s n=%-7d " % (kind, d.numel()) + "  ".join(
            "[%d,%s): %d" % (a, b if b < 200 else "inf", n) for (a, b), n in zip(zip(edges[:-1], edges[1:]), counts)))
    vs = [verdicts(r) for r in rows]
    lines.append("")
    lines.append("worst rel over all GEMMs: %.2f   worst abs: %.2e (mode 0: %.2e)   worst relative error of a deep channel: %.1e"
                 % (max(v["rel"] for v in vs), max(v["abs"] for v in vs), max(v["abs0"] for v in vs),
                    max(v["deep_rel"] for v in vs)))
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
