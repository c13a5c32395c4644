#!/bin/bash
# usage: timeline.sh <tag> <bench args...>   (env passes through)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/tl_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py "$@" --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing > /tmp/tl_$TAG.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $R/tools/scratch/step_timeline.py "$F" 0 > $R/gpurun_out/timeline_$TAG.txt
head -1 $R/gpurun_out/timeline_$TAG.txt
