#!/bin/bash
# SQ counters of the GEMM kernels on one conv shape (separate --pmc passes, kernel-trace only)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  OUT=/tmp/pmcconv_${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- python $R/tools/pmc_conv_probe.py "$@" > $R/gpurun_out/$TAG/log_$i.txt 2>&1
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "pass $i: no counter file"; tail -5 $R/gpurun_out/$TAG/log_$i.txt; continue; fi
  python - "$F" <<'PY' | tee -a $R/gpurun_out/$TAG/counters.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv_" in k and "dadet" in k:
        acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        print("%-42s %-32s %16.0f  (n=%d)" % (k, c, sum(v[1:]) / max(len(v) - 1, 1), len(v)))
PY
done
