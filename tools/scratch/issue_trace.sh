#!/bin/bash
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/it_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT -- python $R/bench.py "$@" --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing > /tmp/it_$TAG.log 2>&1
tail -2 /tmp/it_$TAG.log | cut -c1-200
python $R/tools/scratch/issue_delay.py $OUT 30 > $R/gpurun_out/issue_$TAG.txt 2>&1
head -3 $R/gpurun_out/issue_$TAG.txt
