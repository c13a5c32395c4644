# usage: ab.sh "<ENV=a ENV=b ...>" [workloads]
VARS=${1:-"X=0 X=1"}; WLS=${2:-img_only}
for wl in $WLS; do
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing --others none --workload $wl"
for i in 1 2; do
for v in $VARS; do
  echo "== $wl $v: $(env $v timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
done; done; done
