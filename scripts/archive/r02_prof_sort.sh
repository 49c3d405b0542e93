#!/bin/bash
# per-launch durations of the device std::sort inside the frame pipeline (rocprofv3 kernel trace -> rocpd database, read with sqlite3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ps; mkdir -p $R/gpurun_out/ps
FRAMEBENCH_DEV_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/ps/out -o ps -- python $R/scripts/framebench.py < /dev/null > $R/gpurun_out/ps/log.txt 2>&1
ls $R/gpurun_out/ps/out | head
