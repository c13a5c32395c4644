#!/bin/bash
# usage: build_ab_lib.sh <git-rev> [name]  ->  tools/native/ab_lib/libdadet_<name>.so built from that revision's csrc
# (A/B of two library builds on ONE box: DADET_LIB=tools/native/ab_lib/libdadet_<name>.so python bench.py ...)
set -e
REV=${1:?revision}; NAME=${2:-$REV}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d)
mkdir -p $W/da_detect_amd/csrc $W/include $ROOT/tools/native/ab_lib
for f in $(git -C $ROOT ls-tree --name-only $REV da_detect_amd/csrc/); do git -C $ROOT show $REV:$f > $W/$f; done
git -C $ROOT show $REV:include/dadet.h > $W/include/dadet.h
make -C $W/da_detect_amd/csrc -j8 > /dev/null
cp $W/da_detect_amd/libdadet_hip.so $ROOT/tools/native/ab_lib/libdadet_$NAME.so
rm -rf $W
echo built tools/native/ab_lib/libdadet_$NAME.so
