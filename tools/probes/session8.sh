set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_multirank_gpu.py -x -q -k "production" -s 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/s8_rccl.log 2>&1; tail -4 gpurun_out/s8_rccl.log
timeout 300 python tools/probes/second_model_probe.py first da > gpurun_out/s8_second_model_first.log 2>&1
timeout 400 python tools/probes/second_model_probe.py second da > gpurun_out/s8_second_model_second.log 2>&1
grep -h "ms/step\|tables\|allocator" gpurun_out/s8_second_model_first.log gpurun_out/s8_second_model_second.log
timeout 600 python tools/op_sites.py fpn_dcn_da > gpurun_out/s8_op_sites_fpn.log 2>&1; tail -5 gpurun_out/s8_op_sites_fpn.log
