#!/bin/bash
# parity suite with the correspondence kernel's lanes pinned to 8 and to 16 (every search variant against the oracle), then the frame-level scripts
mkdir -p gpurun_out/lf
for L in 8 16; do
  MLH_KNN_LANES=$L timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/lf/pytest_lanes$L.log 2>&1
  echo "[lanes $L] rc=$? $(grep -E 'passed|failed' gpurun_out/lf/pytest_lanes$L.log | tail -1)"
done
timeout 600 python scripts/framebench.py > gpurun_out/lf/framebench.txt 2>&1; echo "[framebench] rc=$?"; tail -12 gpurun_out/lf/framebench.txt
timeout 600 python scripts/trackbench.py > gpurun_out/lf/trackbench.txt 2>&1; echo "[trackbench] rc=$?"; tail -4 gpurun_out/lf/trackbench.txt
