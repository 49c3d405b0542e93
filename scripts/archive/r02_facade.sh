#!/bin/bash
mkdir -p gpurun_out/chk
timeout 300 python -m pytest tests/test_gpu_facade.py -q -x < /dev/null > gpurun_out/chk/facade.log 2>&1; echo "rc=$?"; grep -E "passed|failed|stdout|extract:|matchSurf|scan2map|voxel|cloudUCT|Lidar|track|device-resident|round-2|returncode" gpurun_out/chk/facade.log | head -30 | cut -c1-300
