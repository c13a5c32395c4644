#!/bin/bash
# The profile set of a round, on the GPU box (one gpurun call): kernel stats of the four bench workloads, HBM counters
# of the three R-50-C4 ones, step timeline + idle gaps, per-shape GEMM table with the GEMM-free stretches, and the default
# bench line.  Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
# usage: tools/profile_round.sh <tag>
#        tools/profile_round.sh <tag> --ranks N     (an N-GPU node: kernel trace of the N-rank bench with the RCCL kernels)
set -u
TAG=${1:-r04}
if [ "${2:-}" = "--ranks" ]; then
  # One rocprofv3 per rank (launched under torch.distributed.run, one output directory per rank): the kernel trace holds
  # the ncclDevKernel_* launches of the bucket all-reduces on RCCL's stream next to the backward GEMMs of the compute stream;
  # tools/step_timeline.py on rank 0's trace shows where in backward each of them ran.  The bench line's `comm` block carries
  # the same overlap as numbers (buckets issued during backward, all-reduce time, exposed wait in finalize()).
  N=${3:?--ranks N}
  R=${GRAFT_REPO_ROOT:-/root/repo}
  cd $R
  export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
  OUT=gpurun_out/${TAG}_ranks${N}
  mkdir -p $OUT
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
      --no-python bash -c "cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/rank\$RANK -o trace -- \
      python $R/bench.py --gpus $N --steps 12 --warmup 6 --others none --resolutions none --no-cpu-baseline" \
      > $OUT/bench.log 2> $OUT/bench.err
  grep '^{"metric"' $OUT/bench.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('comm'))"
  for r in $(seq 0 $((N - 1))); do
    f=$(ls $OUT/rank$r/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && { echo "== rank $r: RCCL kernels"; grep -i "nccl" $f | cut -c1-160 | head -5; }
  done
  exit 0
fi
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for wl in img_only da triplet fpn_dcn_da; do
  echo "=== kernel stats: $wl"
  bash tools/profile_bench.sh ${TAG}_stats_$wl --workload $wl --others none --steps 12 --warmup 6 > gpurun_out/${TAG}_stats_$wl.log 2>&1
  grep '^{"metric"' gpurun_out/${TAG}_stats_$wl/bench.log | tail -1 | cut -c1-200
done
for wl in img_only da triplet; do
  echo "=== PMC: $wl"
  bash tools/profile_pmc.sh ${TAG}_pmc_$wl --workload $wl --others none --steps 6 --warmup 3 > gpurun_out/${TAG}_pmc_$wl.log 2>&1
  python tools/pmc_combine.py gpurun_out/${TAG}_pmc_$wl/FETCH_SIZE.csv gpurun_out/${TAG}_pmc_$wl/WRITE_SIZE.csv \
      gpurun_out/${TAG}_pmc_$wl/hbm_traffic.json "bench.py --workload $wl --others none --steps 6 --warmup 3 (two rocprofv3 --pmc passes)"
  ls -la gpurun_out/${TAG}_pmc_$wl | tail -4
done
for wl in img_only da fpn_dcn_da; do
  echo "=== timeline: $wl"
  bash tools/profile_gaps.sh ${TAG}_gaps_$wl --workload $wl --steps 12 --warmup 6 --others none 2>&1 | tail -3
done
echo "=== GEMM table"
python tools/gemm_table.py --workload img_only --steps 3 --top 60 --holes > gpurun_out/${TAG}_gemm_table_img_only.txt 2>&1; tail -3 gpurun_out/${TAG}_gemm_table_img_only.txt
python tools/gemm_table.py --workload img_only --steps 3 --top 60 > gpurun_out/${TAG}_gemm_table_per_shape.txt 2>&1
python tools/gemm_table.py --workload fpn_dcn_da --steps 3 --top 80 --holes > gpurun_out/${TAG}_gemm_table_fpn_dcn_da.txt 2>&1; tail -3 gpurun_out/${TAG}_gemm_table_fpn_dcn_da.txt
python tools/gemm_table.py --workload da --steps 3 --top 60 > gpurun_out/${TAG}_gemm_table_da.txt 2>&1
python tools/gemm_table.py --workload fpn_dcn_da --steps 3 --top 90 > gpurun_out/${TAG}_gemm_table_per_shape_fpn_dcn_da.txt 2>&1
echo "=== R-101-FPN-DCN bench line"
python bench.py --workload fpn_dcn_da --others none --no-cpu-baseline > gpurun_out/${TAG}_bench_fpn_dcn_da.json 2>/dev/null; tail -1 gpurun_out/${TAG}_bench_fpn_dcn_da.json | cut -c1-200
echo "=== two ranks on one GPU over gloo (functional rig: the N > 1 line with its comm block)"
DADET_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 6 --warmup 4 > gpurun_out/${TAG}_bench_2ranks_one_gpu_gloo.json 2> gpurun_out/${TAG}_bench_2ranks.err; tail -1 gpurun_out/${TAG}_bench_2ranks_one_gpu_gloo.json | cut -c1-200
echo "=== default bench"
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; tail -1 gpurun_out/${TAG}_bench_default.json | cut -c1-300
