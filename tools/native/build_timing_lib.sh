#!/bin/bash
# Probe build of the operator library with the phase stamps of conv_big_kernel compiled in (-DDADET_BIG_TIMING=1), and of the
# GEMM laboratory: tools/native/timing_lib/libdadet_hip.so + tools/native/gemm_lab.  Cross-compiles without a GPU.
#   LD_LIBRARY_PATH=tools/native/timing_lib tools/native/gemm_lab t M N K      (on the GPU box)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/da_detect_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $R/tools/native/timing_lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DDADET_BIG_TIMING=1 -c $C/conv_big.hip -o $R/tools/native/timing_lib/conv_big.o
OBJS=$(ls $C/*.o | grep -v conv_big.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/tools/native/timing_lib/conv_big.o -o $R/tools/native/timing_lib/libdadet_hip.so
hipcc --offload-arch=gfx950 -O3 -w $R/tools/native/gemm_lab.hip -L$R/da_detect_amd -ldadet_hip -ldl -o $R/tools/native/gemm_lab
echo built
