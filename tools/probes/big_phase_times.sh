#!/bin/bash
# phase times of the large-tile forward kernels per workgroup (tools/native/build_timing_lib.sh first): box-head / RPN shapes on
# the 256 x 256 tile, res3 / res4 shapes on the 256 x 128 tile
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LD_LIBRARY_PATH=tools/native/timing_lib DADET_BIG_GEMM=2
if [ "${1:-256}" = "256" ]; then
for shape in "12544 512 4608" "12544 512 2048" "12544 2048 512" "8192 1024 9216" "16384 256 2304"; do
  for s in 1 2 4; do
    DADET_BIG_SPLITS=$s timeout 60 tools/native/gemm_lab t $shape
  done
done
else
export DADET_BIG_TILE_N=128
for shape in "16384 256 2304" "65536 128 1152" "16384 256 1024" "65536 128 512"; do
  for s in 1 2; do
    DADET_BIG_SPLITS=$s timeout 60 tools/native/gemm_lab t $shape
  done
done
fi
