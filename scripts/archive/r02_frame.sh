#!/bin/bash
mkdir -p gpurun_out/fr
timeout 240 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py -q -x -k "std_sort or voxel or downsample or member or mixed or local_map" --durations=3 < /dev/null > gpurun_out/fr/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/fr/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)|^E  |s call" gpurun_out/fr/pytest.log | head -30
FRAMEBENCH_DEV_ONLY=1 timeout 200 python scripts/framebench.py < /dev/null > gpurun_out/fr/framebench.txt 2>&1; tail -4 gpurun_out/fr/framebench.txt | cut -c1-330
