#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ps
FRAMEBENCH_DEV_ONLY=1 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ps/out -o ps -- python $R/scripts/framebench.py > $R/gpurun_out/ps/log.txt 2>&1
f=$(find $R/gpurun_out/ps/out -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-160
t=$(find $R/gpurun_out/ps/out -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# durations of stdsort_level_kernel in launch order for the last sort call
lv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "stdsort" in r["Kernel_Name"]]
lv.sort()
# find last init
idx = [i for i, r in enumerate(lv) if "init" in r[2]]
s = idx[-1]
seq = lv[s:]
print("last sort call: %d launches, span %.1f us" % (len(seq), (seq[-1][1] - seq[0][0]) / 1e3))
print("durations us:", [round((e - b) / 1e3, 1) for b, e, _ in seq])
print("gaps us:", [round((seq[i + 1][0] - seq[i][1]) / 1e3, 1) for i in range(len(seq) - 1)])
PY
