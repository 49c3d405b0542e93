#!/bin/bash
# one bench step's GPU timeline (rocprofv3 kernel trace)
REPO=$PWD
OUT=$REPO/gpurun_out/r03/tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-events 0 > $OUT/bench.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
python $REPO/scripts/timeline.py $DB 70 pack_check_kernel ${FRACTION:-0.3} > $OUT/timeline.txt
rm -rf $OUT/trace
head -70 $OUT/timeline.txt
