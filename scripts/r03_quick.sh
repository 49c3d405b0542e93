#!/bin/bash
for i in 1 2 3; do for lib in m-loam_amd/lib/libmloam_hip_base.so m-loam_amd/lib/libmloam_hip.so; do MLOAM_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['ms_per_step_synchronous_submission'])"; done; done
