python -m pytest tests/test_gpu_edge_cases.py -q -m gpu -k "submitted_and_collected" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap', d['ms_per_step'], d.get('ms_per_step_synchronous_submission'))"; done
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-overlap-staging 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-overlap', d['ms_per_step'])"
FRACTION=0.3 bash scripts/r03_timeline.sh 2>&1 | tail -60
