#!/bin/bash
# kernel trace of bench.py + idle-gap analysis (only the text summary is kept)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/gaps_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/$TAG/bench.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_analysis.py "$F" 20 4 > $R/gpurun_out/$TAG/gaps.txt; tail -5 $R/gpurun_out/$TAG/gaps.txt | cut -c1-200
python $R/tools/step_timeline.py "$F" 15 2 ${TIMELINE_LIST:-} > $R/gpurun_out/$TAG/step_timeline.txt; head -3 $R/gpurun_out/$TAG/step_timeline.txt | cut -c1-200
