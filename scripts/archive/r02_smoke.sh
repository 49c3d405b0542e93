#!/bin/bash
mkdir -p gpurun_out/chk
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_facade.py tests/test_gpu_bench_contract.py -q -x < /dev/null > gpurun_out/chk/facade.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/chk/facade.log | tail -1)"
