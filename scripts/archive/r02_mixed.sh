#!/bin/bash
mkdir -p gpurun_out/mx
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "plain_parity or mixed_lidar or downsample or member_order or voxel" > gpurun_out/mx/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/mx/pytest.log | tail -1)"; grep -E "^E  |Error" gpurun_out/mx/pytest.log | head -12
timeout 600 python scripts/framebench.py > gpurun_out/mx/framebench.txt 2>&1; echo "[framebench] rc=$?"; tail -5 gpurun_out/mx/framebench.txt
