#!/bin/bash
mkdir -p gpurun_out/chk
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline < /dev/null > gpurun_out/chk/bench_final.json 2> gpurun_out/chk/bench_final.log; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/chk/bench_final.json"))
print(d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["roofline"]["frac"], d.get("final_pose", [])[:3])
PY
