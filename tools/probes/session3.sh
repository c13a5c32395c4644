set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "weight_stationary or anchor or pyramid or roi_align" 2>&1 | tail -15 ) > gpurun_out/s3_tests.log 2>&1
tail -8 gpurun_out/s3_tests.log
timeout 300 python tools/probes/ws_probe.py > gpurun_out/s3_ws_probe.log 2>&1; cat gpurun_out/s3_ws_probe.log
bash tools/probes/ab.sh "DADET_WS_1X1=0 DADET_WS_1X1=1" "img_only" > gpurun_out/s3_ab_ws.log 2>&1; cat gpurun_out/s3_ab_ws.log
bash tools/probes/ab.sh "DADET_ANCHOR_SCAN=0 DADET_ANCHOR_SCAN=1" "img_only fpn_dcn_da" > gpurun_out/s3_ab_anchor.log 2>&1; cat gpurun_out/s3_ab_anchor.log
bash tools/probes/ab.sh "DADET_PYRAMID_ROIALIGN=0 DADET_PYRAMID_ROIALIGN=1" "fpn_dcn_da" > gpurun_out/s3_ab_pyr.log 2>&1; cat gpurun_out/s3_ab_pyr.log
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_default_path_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/s3_tests_model.log 2>&1
tail -5 gpurun_out/s3_tests_model.log
