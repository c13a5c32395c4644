#!/usr/bin/env python
"""Idle gaps between consecutive kernels in a rocprofv3 kernel_trace.csv: where does the GPU wait for the host?
usage: gap_analysis.py <kernel_trace.csv> [min_gap_us] [last_n_steps]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if steps:  # steady-state window: from the end of the (steps+1)-th last optimizer kernel to the end of the last one
    ends = [e for _, e, n in rows if "sgd_kernel" in n]
    lo, hi = ends[-steps - 1], ends[-1]
    lo_window, hi_window = lo, hi
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print("window: last %d steps, %.2f ms/step" % (steps, (hi - lo) / 1e6 / steps))
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
union, cur_s, cur_e = 0, rows[0][0], rows[0][1]
big_idle = []
for s_, e_, n_ in rows[1:]:
    if s_ > cur_e:
        union += cur_e - cur_s
        if s_ - cur_e > 50000:
            big_idle.append((s_ - cur_e, n_))
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += cur_e - cur_s
print("union of all queues: GPU executing something %.1f ms, nothing %.1f ms (%.1f%%); idle stretches > 50 us: %s" % (
    union / 1e6, (span - union) / 1e6, 100.0 * (span - union) / span,
    ", ".join("%.0fus before %s" % (g / 1e3, n[:40]) for g, n in sorted(big_idle, reverse=True)[:12])))
print("kernels %d  busy %.1f ms  span %.1f ms  idle %.1f ms (%.1f%%)" % (len(rows), busy / 1e6, span / 1e6,
                                                                       (span - busy) / 1e6, 100.0 * (span - busy) / span))
gaps = defaultdict(lambda: [0, 0.0])
hist = defaultdict(float)
prev_end, prev_name = rows[0][1], rows[0][2]
for s, e, name in rows[1:]:
    g = (s - prev_end) / 1e3
    if g > 0:
        b = "<5us" if g < 5 else "<20us" if g < 20 else "<100us" if g < 100 else "<1ms" if g < 1000 else ">=1ms"
        hist[b] += g
    if g >= min_gap:
        k = (prev_name, name)
        gaps[k][0] += 1
        gaps[k][1] += g
    if e > prev_end:
        prev_end, prev_name = e, name
print("idle time by gap size (ms):", {k: round(v / 1e3, 2) for k, v in hist.items()})
print("top gaps >= %.0f us (count, total ms, after -> before):" % min_gap)
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%5d %8.2f  %s  ->  %s" % (n, t / 1e3, a, b))

# ---- runs of small (library) kernels between this repo's kernels, last step only
if steps:
    ends = [e for _, e, n in rows if "sgd_kernel" in n]
    lo = ends[-2] if len(ends) > 1 else rows[0][0]
    last = [r for r in rows if r[0] >= lo]
    print("\nlast step: runs of library kernels between dadet:: kernels (n >= 8): n, wall us (incl. gaps), busy us, after-kernel")
    run_n, run_start, run_busy, prev = 0, None, 0.0, "step start"
    prev_end = last[0][0]
    tot_n = tot_wall = 0
    for s, e, name in last:
        if "dadet::" in name:
            if run_n >= 8:
                print("%5d %9.1f %9.1f   after %s  -> before %s" % (run_n, (s - run_start) / 1e3, run_busy / 1e3, prev[:48], name[:48]))
            if run_n:
                tot_n += run_n
                tot_wall += (s - run_start) / 1e3
            run_n, run_busy, prev, run_start = 0, 0.0, name, None
            prev_end = e
        else:
            if run_n == 0:
                run_start = prev_end
            run_n += 1
            run_busy += e - s
    print("library kernels in the step: %d, wall %.2f ms (incl. their gaps)" % (tot_n, tot_wall / 1e3))

# ---- per-queue view (streams map to HSA queues): busy time per queue and the compute queue's largest idle gaps
if steps:
    import collections
    qrows = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s_ >= lo_window and e_ <= hi_window:
                qrows[r.get("Queue_Id", "?")].append((s_, e_, r["Kernel_Name"][:60]))
    print("\nper queue (window of %d steps): kernels, busy ms/step" % steps)
    for q, rs in sorted(qrows.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        print("  queue %s: %5d kernels  %.2f ms/step" % (q, len(rs), sum(e - s for s, e, _ in rs) / 1e6 / steps))
    mainq = max(qrows, key=lambda q: sum(e - s for s, e, _ in qrows[q]))
    rs = sorted(qrows[mainq])
    gaps2 = []
    for (s0, e0, n0), (s1, e1, n1) in zip(rs, rs[1:]):
        if s1 - e0 > 30000:
            gaps2.append((s1 - e0, n0, n1))
    gaps2.sort(reverse=True)
    print("compute queue %s: idle gaps > 30 us: %d, total %.2f ms/step; largest:" % (
        mainq, len(gaps2), sum(g for g, _, _ in gaps2) / 1e6 / steps))
    for g, a, b in gaps2[:25]:
        print("  %8.1f us  after %s -> before %s" % (g / 1e3, a[:50], b[:50]))

# ---- stretches with NO GEMM kernel running: what does the GPU do there? (last step of the window)
if steps:
    ends = [e for _, e, n in rows if "sgd_kernel" in n]
    lo, hi = (ends[-2], ends[-1]) if len(ends) >= 2 else (rows[0][0], rows[-1][1])
    last = [r for r in rows if r[0] >= lo and r[1] <= hi]
    is_gemm = lambda n: ("conv_fwd" in n or "conv_big" in n or "conv1x1_ws" in n or "conv_wgrad_split" in n  # noqa: E731
                         or "conv_wgrad_big" in n or "conv_wgrad_kernel" in n)
    gemm = sorted((s, e) for s, e, n in last if is_gemm(n))
    merged = []
    for s, e in gemm:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    holes, prev = [], lo
    for s, e in merged:
        if s > prev:
            holes.append((prev, s))
        prev = max(prev, e)
    if hi > prev:
        holes.append((prev, hi))
    total = sum(b - a for a, b in holes)
    print("\nlast step %.2f ms: no GEMM kernel running for %.2f ms in %d stretches; the longest, with the kernels that "
          "run inside (us of the stretch they cover):" % ((hi - lo) / 1e6, total / 1e6, len(holes)))
    others = [(s, e, n) for s, e, n in last if not is_gemm(n)]
    for a, b in sorted(holes, key=lambda h: h[0] - h[1])[:14]:
        inside = defaultdict(float)
        covered = []
        for s, e, n in others:
            if e > a and s < b:
                inside[n[:48]] += (min(e, b) - max(s, a)) / 1e3
                covered.append((max(s, a), min(e, b)))
        covered.sort()
        cov, ce = 0.0, a
        for s, e in covered:
            if e > ce:
                cov += e - max(s, ce)
                ce = e
        after = [n for s, e, n in last if is_gemm(n) and s >= b][:1]
        print("  %7.1f us at +%.2f ms (GPU idle %5.1f us) before %s: %s" % (
            (b - a) / 1e3, (a - lo) / 1e6, (b - a - cov) / 1e3, (after[0][:40] if after else "end"),
            ", ".join("%s %.0f" % (k, v) for k, v in sorted(inside.items(), key=lambda kv: -kv[1])[:5])))

# ---- optional timeline of the last step: DADET_TIMELINE="from_ms:to_ms" prints every kernel (runs of one name merged)
import os  # noqa: E402
if steps and os.environ.get("DADET_TIMELINE"):
    t0, t1 = [float(v) for v in os.environ["DADET_TIMELINE"].split(":")]
    with open(path, newline="") as f:
        full = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"))
                for r in csv.DictReader(f)]
    full = sorted(r for r in full if r[0] >= lo + t0 * 1e6 and r[0] <= lo + t1 * 1e6)
    print("\ntimeline of the last step, +%.1f .. +%.1f ms (start ms, dur us, queue, kernel [x count])" % (t0, t1))
    i = 0
    while i < len(full):
        j = i
        while j + 1 < len(full) and full[j + 1][2] == full[i][2] and full[j + 1][3] == full[i][3]:
            j += 1
        dur = sum(e - s for s, e, _, _ in full[i:j + 1]) / 1e3
        print("  %8.3f %8.1f  q%-3s %s%s" % ((full[i][0] - lo) / 1e6, dur, full[i][3], full[i][2],
                                           " x%d" % (j - i + 1) if j > i else ""))
        i = j + 1
