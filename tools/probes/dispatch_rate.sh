# per kernel: workgroups per launch, threads per workgroup, time per workgroup — a kernel at a few ns per (small) workgroup is
# bound by the dispatch rate, not by its work.  usage: dispatch_rate.sh [bench args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dr; rocprofv3 --kernel-trace --output-format csv -d /tmp/dr -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-kernel-timing --others none --steps 6 --warmup 4 > /tmp/dr.log 2>&1
F=$(find /tmp/dr -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    d[(r["Kernel_Name"].split("(")[0][-60:], g // max(w, 1), w)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for (n, wgs, w), v in d.items():
    avg = sum(v) / len(v)
    rows.append((avg / max(wgs, 1), n, wgs, w, len(v), avg / 1e3))
print("ns/workgroup  workgroups  threads  launches  avg us  kernel   (workgroups >= 2000 only)")
for nspw, n, wgs, w, cnt, avg in sorted(rows):
    if wgs >= 2000 and avg > 8:
        print("%10.1f  %9d  %6d  %7d  %7.1f  %s" % (nspw, wgs, w, cnt, avg, n))
PY
