python tools/probes/data_dependence.py 4,3,2 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_data_dependence_modes.txt
for wl in img_only fpn_dcn_da; do python tools/probes/amax_sites.py $wl 2>&1 | grep -v amdgpu.ids | cut -c1-300; done > gpurun_out/r04_amax_sites.txt
