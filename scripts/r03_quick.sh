#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_fullsize.py -q -m gpu -x 2>&1 | tail -4
for i in 1 2 3; do python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['ms_per_step_synchronous_submission'], d['kernel_us_per_launch'], d['final_pose'][:3])"; done
