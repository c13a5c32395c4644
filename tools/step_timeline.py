#!/usr/bin/env python
"""What runs while no GEMM runs: the GEMM-free stretches of one steady-state training step, from a rocprofv3
kernel_trace.csv (tools/profile_gaps.sh), each with the kernels that executed inside it.

usage: step_timeline.py <kernel_trace.csv> [min_stretch_us=15] [step_from_end=1] [list=A:B]
       list=A:B  also print EVERY kernel (GEMMs included) that starts between A and B ms into the step, in start order

A "GEMM" is one of this repo's MFMA kernels (conv_fwd_* / conv_big* / conv1x1_ws_* / conv_wgrad_*); everything else — library launches, the
latency-bound kernels of proposal selection and sampling, ROIAlign, reductions, the optimizer — only costs wall time
where it is not hidden under a GEMM, i.e. inside these stretches.  The step window runs from the end of one sgd_kernel to
the end of the next."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
window = None
for a in sys.argv[4:]:
    if a.startswith("list="):
        window = tuple(float(v) for v in a[5:].split(":"))
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
rows.sort()
ends = [e for _, e, n, _ in rows if "sgd_kernel" in n]
lo, hi = ends[-back - 1], ends[-back]
step = [r for r in rows if r[1] > lo and r[0] < hi]
is_gemm = lambda n: ("conv_fwd_" in n or "conv_wgrad_" in n or "conv1x1_ws" in n or "conv_big" in n) and "reduce" not in n  # noqa: E731


def short(n):
    n = n.replace("void ", "").replace("dadet::", "")
    for cut in ("(", "<at::native", ", std::array"):
        i = n.find(cut, 12)
        if i > 0:
            n = n[:i]
    return n[:58]


# union of GEMM intervals
gem = sorted((max(s, lo), min(e, hi)) for s, e, n, _ in step if is_gemm(n))
merged = []
for s, e in gem:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
busy = sum(e - s for s, e in merged)
free = []
prev = lo
for s, e in merged:
    if s > prev:
        free.append((prev, s))
    prev = max(prev, e)
if hi > prev:
    free.append((prev, hi))
print("step %.3f ms: a GEMM running %.3f ms, none %.3f ms in %d stretches; kernels in the step: %d (%d GEMM launches)" % (
    (hi - lo) / 1e6, busy / 1e6, (hi - lo - busy) / 1e6, len(free), len(step), len(gem)))
tot_by_kernel = defaultdict(lambda: [0, 0.0])
print("GEMM-free stretches of at least %.0f us, longest first (offset in the step, length; kernels inside: start offset in "
      "the stretch, duration, queue):" % min_us)
for a, b in sorted(free, key=lambda ab: ab[0] - ab[1]):
    inside = [(max(s, a), min(e, b), n, q) for s, e, n, q in step if e > a and s < b and not is_gemm(n)]
    covered = 0
    cur = a
    for s, e, n, q in sorted(inside):
        if e > cur:
            covered += e - max(s, cur)
            cur = e
        tot_by_kernel[short(n)][0] += 1
        tot_by_kernel[short(n)][1] += (e - s) / 1e3
    if (b - a) / 1e3 < min_us:
        continue
    print("  +%7.3f ms  %6.1f us  (GPU idle inside: %5.1f us)" % ((a - lo) / 1e6, (b - a) / 1e3, (b - a - covered) / 1e3))
    for s, e, n, q in sorted(inside):
        print("        +%6.1f  %6.1f us  q%-3s %s" % ((s - a) / 1e3, (e - s) / 1e3, str(q)[-3:], short(n)))
print("time of non-GEMM kernels INSIDE GEMM-free stretches, by kernel (count, us):")
for n, (c, t) in sorted(tot_by_kernel.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %4d %8.1f  %s" % (c, t, n))
if window is not None:
    print("every kernel starting between +%.2f and +%.2f ms (start offset ms, duration us, queue):" % window)
    for s0, e0, n, q in step:
        off = (s0 - lo) / 1e6
        if window[0] <= off < window[1]:
            print("  +%7.3f  %7.1f us  q%-3s %s%s" % (off, (e0 - s0) / 1e3, str(q)[-3:], "GEMM " if is_gemm(n) else "", short(n)))
