#!/bin/bash
mkdir -p gpurun_out/chk
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_fullsize.py tests/test_gpu_golden.py -q -x -k "voxel or extract or ring or golden or config1 or joint or 128" < /dev/null > gpurun_out/chk/a3.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/chk/a3.log | tail -1)"
grep -iE "^(FAILED|ERROR)|^E  " gpurun_out/chk/a3.log | head -20
FRAMEBENCH_DEV_ONLY=1 timeout 200 python scripts/framebench.py < /dev/null > gpurun_out/chk/frame_a3.txt 2>&1; grep "GPU path" gpurun_out/chk/frame_a3.txt | cut -c1-250
