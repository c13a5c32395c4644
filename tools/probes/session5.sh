set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pyramid or roi_align" 2>&1 | tail -15 ) > gpurun_out/s5_tests.log 2>&1
tail -4 gpurun_out/s5_tests.log
bash tools/probes/ab.sh "DADET_ANCHOR_SCAN=0 DADET_ANCHOR_SCAN=1" "fpn_dcn_da" > gpurun_out/s5_ab_anchor.log 2>&1; cat gpurun_out/s5_ab_anchor.log
bash tools/probes/ab.sh "DADET_PYRAMID_ROIALIGN=0 DADET_PYRAMID_ROIALIGN=1" "fpn_dcn_da" > gpurun_out/s5_ab_pyr.log 2>&1; cat gpurun_out/s5_ab_pyr.log
( timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_default_path_gpu.py tests/test_full_size_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -12 ) > gpurun_out/s5_tests_model.log 2>&1
tail -6 gpurun_out/s5_tests_model.log
