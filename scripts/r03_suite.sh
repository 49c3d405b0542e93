#!/bin/bash
# round 3: full GPU suite without -x (all failures at once), then a bench line
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q --durations=15 < /dev/null > gpurun_out/r03/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/r03/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)" gpurun_out/r03/pytest.log | head -30
timeout 300 python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; echo "[bench] rc=$?"; head -c 600 gpurun_out/r03/bench.json
