#!/bin/bash
# device std::sort: parity tests, then per-launch durations inside the frame pipeline
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/fr
cd $R
timeout 240 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py -q -x -k "std_sort or voxel or downsample or member or mixed or local_map" < /dev/null > gpurun_out/fr/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/fr/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)|^E  " gpurun_out/fr/pytest.log | head -20
bash scripts/r02_prof_sort.sh
grep "GPU path" $R/gpurun_out/ps/log.txt | cut -c1-260
