set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "weight_stationary or vectorised or split_bf16 or conv" 2>&1 | tail -15 ) > gpurun_out/s2_tests.log 2>&1
tail -8 gpurun_out/s2_tests.log
timeout 300 python tools/probes/ws_probe.py > gpurun_out/s2_ws_probe.log 2>&1; cat gpurun_out/s2_ws_probe.log
bash tools/probes/ab.sh "DADET_WS_1X1=0 DADET_WS_1X1=1" "img_only da" > gpurun_out/s2_ab_ws.log 2>&1; cat gpurun_out/s2_ab_ws.log
