#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_facade.py -q -m gpu -k "voxel or extract or downsample or fuse or device or facade or front_end" 2>&1 | tail -3
for rep in 1 2 3; do
  for lib in m-loam_amd/lib/libmloam_hip_base.so m-loam_amd/lib/libmloam_hip.so; do
    echo "$lib: $(MLOAM_HIP_LIB=$lib FRAMEBENCH_DEV_ONLY=1 timeout 300 python scripts/framebench.py 2>&1 | grep 'one launch set, both kinds' | cut -c90-220)"
  done
done
