#!/bin/bash
# SQ counters of one GEMM-lab kernel (tools/native/gemm_lab c <P|B|D>) on the 16384 x 4096 x 4096 shape: separate --pmc
# passes, kernel-trace only.  usage: tools/pmc_lab.sh <tag> <P|B|D>
set -u
TAG=$1; WHICH=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
export LD_LIBRARY_PATH=$R/da_detect_amd
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  OUT=/tmp/pmclab_${TAG}_${WHICH}_$i
  rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- $R/tools/native/gemm_lab c $WHICH > $R/gpurun_out/$TAG/log_${WHICH}_$i.txt 2>&1
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "pass $i: no counter file"; tail -5 $R/gpurun_out/$TAG/log_${WHICH}_$i.txt; continue; fi
  K=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python - "$F" "$K" <<'PY' | tee -a $R/gpurun_out/$TAG/counters_$WHICH.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm_" in k or "conv_" in k:
        acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        print("%-42s %-32s %16.0f  (n=%d)" % (k, c, sum(v[1:]) / max(len(v) - 1, 1), len(v)))
dur = defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    k = r["Kernel_Name"]
    if "gemm_" in k or "conv_" in k:
        dur[k.split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    print("%-42s %-32s %16.1f  (n=%d)" % (k, "duration_us", sum(v[1:]) / max(len(v) - 1, 1), len(v)))
PY
done
