#!/bin/bash
# full GPU suite, then config-5 selection bench, frame bench (member-order default flipped), one bench line
mkdir -p gpurun_out/sel
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/sel/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/sel/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)|^E  " gpurun_out/sel/pytest.log | head -20
timeout 300 python scripts/gfbench.py > gpurun_out/sel/gfbench.txt 2>&1; tail -5 gpurun_out/sel/gfbench.txt
timeout 400 python scripts/framebench.py < /dev/null > gpurun_out/sel/framebench.txt 2>&1; tail -12 gpurun_out/sel/framebench.txt | cut -c1-400
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/sel/bench.json 2> gpurun_out/sel/bench.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/sel/bench.json"))
print(d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["roofline"]["frac"])
PY
