#!/bin/bash
mkdir -p gpurun_out/r03
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --durations=5 < /dev/null > gpurun_out/r03/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/r03/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)" gpurun_out/r03/pytest.log | head -30
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_form.json 2> gpurun_out/r03/bench_driver_form.err; echo "[bench] rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/bench_driver_form.json").read().strip().splitlines()[-1])
print("driver form: ms/step", d["ms_per_step"], "value", d["value"], "sync", d["ms_per_step_synchronous_submission"], "scan2map", d["scan2map"]["ms_per_frame"])
PY
