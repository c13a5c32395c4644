"""last training step of a rocprofv3 kernel_trace.csv as a compact timeline: start offset, duration, queue, kernel
usage: step_timeline.py <kernel_trace.csv> [min_dur_us]"""
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
min_dur = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ends = [e for _, e, _, n in rows if "sgd_kernel" in n]
lo, hi = ends[-2], ends[-1]
print("step %.3f ms" % ((hi - lo) / 1e6))
queues = {}
for s, e, q, n in rows:
    if s < lo or e > hi:
        continue
    qi = queues.setdefault(q, len(queues))
    if (e - s) / 1e3 < min_dur:
        continue
    short = n.replace("void dadet::", "").replace("(anonymous namespace)::", "")[:70]
    print("%9.1f %8.1f q%d %s" % ((s - lo) / 1e3, (e - s) / 1e3, qi, short))
