#!/bin/bash
# round 3: full GPU suite without -x (all failures at once), the multi-rank worker with one rank, then a bench line
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q --durations=8 < /dev/null > gpurun_out/r03/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/r03/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)" gpurun_out/r03/pytest.log | head -30
for mode in map features; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 tests/_multirank_worker.py $mode 2> gpurun_out/r03/worker_$mode.err | grep "^{" ; echo "[worker $mode] rc=$?"
done
timeout 300 python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; echo "[bench] rc=$?"; head -c 400 gpurun_out/r03/bench.json; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench.json"))
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "unavoidable_frac", "binding", "avg_kernel_us")})
print("fit roofline", {k: d["roofline_time_dominant_kernel"][k] for k in ("achieved", "frac", "avg_kernel_us")}, "outgrow", d["ms_per_step_map_outgrows_its_grid_box"], "spinup", d["gpu_clock_spinup_ms"])
print("kernels", d["kernel_us_per_launch"], "cpu", d["cpu_baseline"]["value"], "s2m", d.get("scan2map", {}).get("ms_per_frame"))
PY
