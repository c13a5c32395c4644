set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_deform_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -k "not oracle" 2>&1 | tail -12 ) > gpurun_out/s10_tests.log 2>&1
tail -4 gpurun_out/s10_tests.log
T=da_detect_amd.modeling.rpn.inference._ROWS_TOPK_SINGLE
for i in 1 2; do for v in False True; do timeout 300 python tools/probes/variant_ab.py $T $v img_only 40 2>/dev/null | tail -1; done; done > gpurun_out/s10_ab_topk_single.log
cat gpurun_out/s10_ab_topk_single.log
for i in 1 2; do for v in False True; do timeout 300 python tools/probes/variant_ab.py $T $v da 30 2>/dev/null | tail -1; done; done > gpurun_out/s10_ab_topk_single_da.log
cat gpurun_out/s10_ab_topk_single_da.log
