"""for the last step of a rocprofv3 run with --kernel-trace --hip-runtime-trace: when was each kernel ISSUED by the host
(hipLaunchKernel / hipModuleLaunchKernel call) and when did it START on the GPU?  usage: issue_delay.py <dir> [min_dur_us]"""
import csv
import glob
import sys

d = sys.argv[1]
min_dur = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
at = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
with open(at, newline="") as f:
    for r in csv.DictReader(f):
        if "Launch" in r["Function"]:
            api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), r["Thread_Id"])
rows = []
with open(kt, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Correlation_Id"], r.get("Queue_Id", "?"),
                     r["Kernel_Name"]))
rows.sort()
ends = [e for _, e, _, _, n in rows if "sgd_kernel" in n]
lo, hi = ends[-2], ends[-1]
print("step %.3f ms" % ((hi - lo) / 1e6))
threads, queues = {}, {}
for s, e, cid, q, n in rows:
    if s < lo or e > hi:
        continue
    qi = queues.setdefault(q, len(queues))
    issued = api.get(cid)
    if (e - s) / 1e3 < min_dur and not any(k in n for k in ("sample", "nms", "rpn_", "roi_", "box_match")):
        continue
    short = n.replace("void dadet::", "").replace("dadet::", "")[:48]
    if issued:
        ti = threads.setdefault(issued[1], len(threads))
        print("start %8.1f  dur %7.1f  q%d  issued %9.1f (t%d)  waited %8.1f  %s" % (
            (s - lo) / 1e3, (e - s) / 1e3, qi, (issued[0] - lo) / 1e3, ti, (s - issued[0]) / 1e3, short))
    else:
        print("start %8.1f  dur %7.1f  q%d  issued ?                      %s" % ((s - lo) / 1e3, (e - s) / 1e3, qi, short))
