#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_golden.py -q -m gpu -k "scan2map or golden or lm" 2>&1 | tail -4
for i in 1 2 3; do python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], 'scan2map', d['scan2map']['ms_per_frame'], d['scan2map']['lm_iterations'])"; done
timeout 300 python scripts/framebench.py 2>&1 | grep "one launch set, both kinds" 
