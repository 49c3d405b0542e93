#!/bin/bash
# chained start pose: the new test, the contract test, then the bench in the driver's form and in the default form
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_bench_contract.py -q -m gpu -k "chained or submitted or contract" 2>&1 | tail -8
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-form', d['ms_per_step'], d['value'], d['final_pose'][:3], d['ms_per_step_synchronous_submission'])"; done
for i in 1 2 3; do python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['value'], d['final_pose'][:3])"; done
