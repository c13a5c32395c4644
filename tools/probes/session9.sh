set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
tail -30 gpurun_out/r04_profile_round.log
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/s9_tests.log 2>&1
tail -4 gpurun_out/s9_tests.log
