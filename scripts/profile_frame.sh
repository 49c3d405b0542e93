#!/bin/bash
# One device-resident mapper frame (scripts/framebench.py, FRAMEBENCH_DEV_ONLY): per-kernel statistics and the GPU timeline of one frame.
# Run through gpurun from the repo root; outputs in gpurun_out/prof_frame_$TAG/ -> copy into profiles/.
TAG=${1:-r01h}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_frame_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FRAMEBENCH_DEV_ONLY=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/trace -o trace -- python $REPO/scripts/framebench.py < /dev/null > $OUT/framebench_under_trace.log 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
  python $REPO/scripts/timeline.py $DB 140 curvature_kernel > $OUT/timeline.txt
fi
rm -rf $OUT/trace
cd $REPO
timeout 300 python scripts/framebench.py < /dev/null > $OUT/framebench.log 2>&1
ls -la $OUT
