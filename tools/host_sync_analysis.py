#!/usr/bin/env python
"""Host side of one training step from a rocprofv3 kernel trace + HIP runtime API trace: every API call in which the host
BLOCKS (synchronisations, device -> host copies) with its place in the step, its length, what the GPU executed meanwhile,
and how much of the step the host spends inside launch calls.
usage: host_sync_analysis.py <kernel_trace.csv> <hip_api_trace.csv> [min_us]"""
import csv
import sys
from collections import defaultdict

kpath, apath = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
kernels = []
with open(kpath, newline="") as f:
    for r in csv.DictReader(f):
        kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
kernels.sort()
ends = [e for _, e, n in kernels if "sgd_kernel" in n]
lo, hi = ends[-3], ends[-2]                     # one steady-state step: between two optimizer kernels
step = [k for k in kernels if k[0] >= lo and k[1] <= hi]
api = []
with open(apath, newline="") as f:
    for r in csv.DictReader(f):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e >= lo and s <= hi:
            api.append((s, e, r["Function"], r.get("Thread_Id", "")))
api.sort()
print("step %.3f ms, %d kernels; HIP API calls inside it: %d" % ((hi - lo) / 1e6, len(step), len(api)))
by_fn = defaultdict(lambda: [0, 0])
for s, e, fn, _ in api:
    by_fn[fn][0] += 1
    by_fn[fn][1] += e - s
print("\nAPI time by function (calls, total us, mean us):")
for fn, (n, t) in sorted(by_fn.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-44s %5d  %9.1f  %7.2f" % (fn, n, t / 1e3, t / 1e3 / n))
threads = defaultdict(int)
for s, e, fn, th in api:
    threads[th] += 1
print("\nthreads issuing calls: %s" % dict(threads))


def gpu_during(s, e):
    busy, names = 0, defaultdict(int)
    for ks, ke, n in step:
        a, b = max(ks, s), min(ke, e)
        if b > a:
            busy += b - a
            names[n.split("(")[0].replace("void ", "").replace("dadet::", "")[:34]] += b - a
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    return busy, ", ".join("%s %.0f" % (n, t / 1e3) for n, t in top)


BLOCKING = ("Synchronize", "hipMemcpy", "hipStreamQuery", "hipEventQuery", "hipMalloc", "hipFree", "hipHostMalloc")
print("\ncalls of at least %.0f us in which the host waits (offset in the step, length, GPU kernel-time inside [summed over queues]):" % min_us)
for s, e, fn, th in api:
    if e - s >= min_us * 1e3 and any(b in fn for b in BLOCKING):
        busy, top = gpu_during(s, e)
        print("  + %7.3f ms  %8.1f us  %-30s thread %s | GPU %.0f us: %s" % ((s - lo) / 1e6, (e - s) / 1e3, fn, th[-5:], busy / 1e3, top))
launch = [(s, e) for s, e, fn, _ in api if "Launch" in fn]
print("\nlaunch calls: %d, %.1f us inside them in total (%.2f us each)" % (len(launch), sum(e - s for s, e in launch) / 1e3,
                                                                       sum(e - s for s, e in launch) / 1e3 / max(1, len(launch))))
# host-idle view: gaps between consecutive API calls of the main thread (python between calls)
main_th = max(threads.items(), key=lambda kv: kv[1])[0]
calls = [(s, e, fn) for s, e, fn, th in api if th == main_th]
gaps = []
for (s0, e0, f0), (s1, e1, f1) in zip(calls[:-1], calls[1:]):
    if s1 - e0 > 30e3:
        gaps.append((s1 - e0, e0, f0, f1))
print("\nstretches of at least 30 us WITHOUT any API call on the main thread (interpreter / ATen dispatch time), longest first:")
for g, at, f0, f1 in sorted(gaps, reverse=True)[:25]:
    busy, top = gpu_during(at, at + g)
    print("  + %7.3f ms  %7.1f us  after %-28s before %-28s | GPU %.0f us: %s" % ((at - lo) / 1e6, g / 1e3, f0[:28], f1[:28], busy / 1e3, top))
print("total %.1f us in %d such stretches" % (sum(g[0] for g in gaps) / 1e3, len(gaps)))
