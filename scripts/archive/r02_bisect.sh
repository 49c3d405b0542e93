#!/bin/bash
# exit status of test selections (a crash at interpreter exit shows as rc != 0 with every test passed)
mkdir -p gpurun_out/bis
i=0
for sel in "$@"; do
  i=$((i+1))
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -q -k "$sel" > gpurun_out/bis/$i.log 2>&1
  echo "[$sel] rc=$? $(grep -E 'passed|failed' gpurun_out/bis/$i.log | tail -1) double-free:$(grep -c 'double free' gpurun_out/bis/$i.log)"
done
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/bis/full.log 2>&1; echo "[full] rc=$? $(grep -E 'passed|failed' gpurun_out/bis/full.log | tail -1)"
