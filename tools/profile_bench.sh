#!/bin/bash
# rocprofv3 kernel-trace of bench.py on the GPU box; keeps only the small CSV summaries under gpurun_out/<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/$TAG/bench.log 2>&1
tail -1 $R/gpurun_out/$TAG/bench.log | cut -c1-600
find $OUT -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_stats.csv \;
find $OUT -name "*domain_stats.csv" -exec cp {} $R/gpurun_out/$TAG/domain_stats.csv \;
ls -la $R/gpurun_out/$TAG
head -30 $R/gpurun_out/$TAG/kernel_stats.csv | cut -c1-200
