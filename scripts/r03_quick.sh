#!/bin/bash
for rep in 1 2 3; do
  for lib in m-loam_amd/lib/libmloam_hip_twopass.so m-loam_amd/lib/libmloam_hip.so; do
    echo "$lib: $(MLOAM_HIP_LIB=$lib FRAMEBENCH_DEV_ONLY=1 timeout 300 python scripts/framebench.py 2>&1 | grep 'one launch set, both kinds' | cut -c90-220)"
  done
done
