set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_full_size_gpu.py tests/test_multirank_gpu.py -x -q -k "oracle or rccl or two_ranks" 2>&1 | tail -40 ) > gpurun_out/s7_tests.log 2>&1
tail -6 gpurun_out/s7_tests.log
