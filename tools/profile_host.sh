#!/bin/bash
# kernel trace + HIP runtime API trace of bench.py: where does the HOST block inside a step, and what is the GPU doing then?
# (no counters in this run: a trace domain next to --pmc is refused on this pool)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/host_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/$TAG/bench.log 2>&1
K=$(find $OUT -name "*kernel_trace.csv" | head -1)
A=$(find $OUT -name "*hip_api_trace.csv" | head -1)
python $R/tools/host_sync_analysis.py "$K" "$A" > $R/gpurun_out/$TAG/host_syncs.txt 2>&1; head -60 $R/gpurun_out/$TAG/host_syncs.txt | cut -c1-220
