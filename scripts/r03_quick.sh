#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_fullsize.py -q -m gpu -k "voxel or extract or downsample or sort or tie" 2>&1 | tail -3
SOAK_EXTRACT=150 SOAK_SORT=2500 SOAK_FRAMES=5 timeout 400 python scripts/r03_soak.py 2>&1 | tail -3 | cut -c1-150
MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so timeout 300 python scripts/stageclock_sort.py 2>&1 | tail -12
FRAMEBENCH_DEV_ONLY=1 timeout 300 python scripts/framebench.py 2>&1 | grep 'one launch set, both kinds' | cut -c90-220
FRAMEBENCH_DEV_ONLY=1 timeout 300 python scripts/framebench.py 2>&1 | grep 'one launch set, both kinds' | cut -c90-220
