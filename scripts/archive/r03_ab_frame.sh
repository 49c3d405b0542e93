#!/bin/bash
# A/B of two library builds on the frame: m-loam_amd/lib/libmloam_hip_base.so (the previous commit) against the current one, three alternations in one call
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_fullsize.py -q -m gpu -k "voxel or extract or downsample or sort or tie" 2>&1 | tail -2
SOAK_EXTRACT=60 SOAK_SORT=3000 SOAK_FRAMES=5 timeout 400 python scripts/r03_soak.py 2>&1 | sed -n 2p | cut -c1-150
for rep in 1 2 3; do
  for lib in m-loam_amd/lib/libmloam_hip_base.so m-loam_amd/lib/libmloam_hip.so; do
    echo "$lib: $(MLOAM_HIP_LIB=$lib FRAMEBENCH_DEV_ONLY=1 timeout 300 python scripts/framebench.py 2>&1 | grep 'one launch set, both kinds' | head -1 | cut -c90-220)"
  done
done
