cd /tmp && export TMPDIR=/tmp
for v in 0; do
  rm -rf /tmp/dp_$v; DADET_DEFORM_ABLATE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_$v -- python $GRAFT_REPO_ROOT/tools/deform_bwd_bench.py 1.5 > /tmp/dp_$v.log 2>&1
  echo "== DADET_DEFORM_ABLATE=$v"; grep -v amdgpu /tmp/dp_$v.log | tail -3
  F=$(find /tmp/dp_$v -name "*kernel_trace.csv" | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "deform" in n:
        d[(n.split("(")[0].replace("dadet::", ""), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    print("  %-34s grid %-9s n %3d  median %7.1f us  min %7.1f" % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
PY
done
