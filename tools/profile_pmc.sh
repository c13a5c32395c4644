#!/bin/bash
# HBM-traffic counters of bench.py on the GPU box, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes:
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC has 4 slots: 3 + 2), so two separate --pmc passes, each with
# --kernel-trace only (never combined with sys/runtime traces).  Keeps only the per-kernel aggregate.
# usage: tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=/tmp/pmc_${TAG}_$C
  rm -rf $OUT; mkdir -p $OUT
  timeout 1200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- \
      python $R/bench.py "$@" --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/$TAG/bench_$C.log 2>&1
  tail -1 $R/gpurun_out/$TAG/bench_$C.log | cut -c1-300
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  echo "counter file: $F"; head -2 "$F" | cut -c1-400
  python $R/tools/pmc_aggregate.py "$F" $C > $R/gpurun_out/$TAG/$C.csv
  head -8 $R/gpurun_out/$TAG/$C.csv | cut -c1-200
done
