set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=$PWD/da_detect_amd
( python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s1_tests.log 2>&1
echo "== tests done"; tail -3 gpurun_out/s1_tests.log
bash tools/probes/ab.sh "DADET_LIB=$P/libdadet_hip_prev.so DADET_LIB=$P/libdadet_hip.so" "img_only da" > gpurun_out/s1_ab_epilogue.log 2>&1
cat gpurun_out/s1_ab_epilogue.log
for lib in libdadet_hip_prev.so libdadet_hip.so; do echo "== $lib"; DADET_LIB=$P/$lib python tools/probes/short_k_probe.py; done > gpurun_out/s1_short_k.log 2>&1
cat gpurun_out/s1_short_k.log
python bench.py > gpurun_out/s1_bench_default.json 2> gpurun_out/s1_bench_default.err
tail -1 gpurun_out/s1_bench_default.json | cut -c1-600
