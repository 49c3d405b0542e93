#!/bin/bash
# config-4 (calibbench) and tracker (trackbench) numbers at HEAD
mkdir -p gpurun_out/ref
timeout 400 python scripts/calibbench.py < /dev/null > gpurun_out/ref/calibbench.txt 2>&1; echo "calib rc=$?"; tail -12 gpurun_out/ref/calibbench.txt | cut -c1-240
timeout 200 python scripts/trackbench.py < /dev/null > gpurun_out/ref/trackbench.txt 2>&1; echo "track rc=$?"; tail -4 gpurun_out/ref/trackbench.txt | cut -c1-240
