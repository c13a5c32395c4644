B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --others none"
for i in 1 2; do
for v in "DADET_WGRAD_REDUCE_STREAM=0" "DADET_WGRAD_REDUCE_STREAM=1" "DADET_WGRAD_REDUCE_STREAM_ITEMS=1" "DADET_WGRAD_REDUCE_STREAM_ITEMS=12"; do
  echo "== $v"; env $v timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
