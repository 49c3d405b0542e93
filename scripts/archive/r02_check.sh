#!/bin/bash
# full GPU suite (exit status kept) + bench lines
mkdir -p gpurun_out/chk
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/chk/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/chk/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/chk/pytest.log | head -20
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/chk/bench$i.json 2> gpurun_out/chk/bench$i.log
python - $i <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/chk/bench{sys.argv[1]}.json"))
print(d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["config"]["map_index"], d["final_pose"][:2], d["roofline"]["frac"])
PY
done
