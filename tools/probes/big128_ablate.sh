#!/bin/bash
# which stage of conv_big128_kernel's load segment costs what: phase times with stages compiled to run-time skips
# (tools/native/build_timing_lib.sh; DADET_ABLATE bit 1 no global loads, 2 no split + LDS store, 4 no fragment reads, 8 no MFMA)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LD_LIBRARY_PATH=tools/native/timing_lib DADET_BIG_GEMM=2 DADET_BIG_TILE_N=128 DADET_BIG_SPLITS=1
for abl in 0 1 2 4 8 3 6 7 9 10 12 14 15; do
  echo -n "ablate=$abl  "; DADET_ABLATE=$abl timeout 60 tools/native/gemm_lab t 16384 256 2304 | tail -1
done
