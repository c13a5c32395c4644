T=tests/test_default_path_gpu.py::test_three_step_trajectory_matches_oracle
for m in 3 4; do for f in 0 1; do
  echo "== mode $m fp64 oracle $f"; DADET_GEMM_MODE=$m DADET_TRAJECTORY_FP64=$f python -m pytest $T -q -m gpu -s 2>&1 | grep -E "three-step|passed|failed|Error" | cut -c1-300
done; done
python -m pytest tests -q -m gpu --deselect $T 2>&1 | tail -30
for wl in img_only fpn_dcn_da; do python tools/probes/amax_sites.py $wl 2>&1 | grep -v amdgpu.ids | cut -c1-330 | head -22; done
