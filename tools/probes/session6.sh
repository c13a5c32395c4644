set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_ops_gpu.py -x -q -k "topk or anchor or pyramid or fpn_selection" 2>&1 | tail -15 ) > gpurun_out/s6_tests.log 2>&1
tail -4 gpurun_out/s6_tests.log
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing --others none --resolutions none --workload fpn_dcn_da"
for i in 1 2; do for v in "DADET_PYRAMID_ROIALIGN=0 DADET_ANCHOR_SCAN=0" "DADET_PYRAMID_ROIALIGN=1 DADET_ANCHOR_SCAN=0" "DADET_PYRAMID_ROIALIGN=1 DADET_ANCHOR_SCAN=1" "DADET_PYRAMID_ROIALIGN=1 DADET_ANCHOR_SCAN=1 DADET_WS_1X1=0"; do
  echo "== $v: $(env $v timeout 300 $B 2>gpurun_out/s6_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" || tail -3 gpurun_out/s6_err.log)"
done; done > gpurun_out/s6_ab_fpn.log 2>&1
cat gpurun_out/s6_ab_fpn.log
( timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_default_path_gpu.py tests/test_full_size_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -40 ) > gpurun_out/s6_tests_model.log 2>&1
tail -6 gpurun_out/s6_tests_model.log
