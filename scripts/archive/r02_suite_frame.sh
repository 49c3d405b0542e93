#!/bin/bash
# full GPU suite (exit status kept), then the frame bench
mkdir -p gpurun_out/chk
timeout 1200 python -m pytest tests -m gpu -q -x < /dev/null > gpurun_out/chk/pytest.log 2>&1; echo "[pytest] rc=$? $(grep -E 'passed|failed' gpurun_out/chk/pytest.log | tail -1)"
grep -iE "^(FAILED|ERROR)|^E  " gpurun_out/chk/pytest.log | head -20
timeout 300 python scripts/framebench.py < /dev/null > gpurun_out/chk/framebench.txt 2>&1; grep "GPU path\|the same" gpurun_out/chk/framebench.txt | cut -c1-260
