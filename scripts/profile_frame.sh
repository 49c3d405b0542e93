#!/bin/bash
# One device-resident mapper frame (scripts/framebench.py, FRAMEBENCH_DEV_ONLY): per-kernel statistics and the GPU timeline of one frame.
# Run through gpurun from the repo root; outputs in gpurun_out/prof_frame_$TAG/ -> copy into profiles/.
TAG=${1:-r01h}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_frame_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FRAMEBENCH_DEV_ONLY=1 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/trace -o trace -- python $REPO/scripts/framebench.py > $OUT/framebench_under_trace.log 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
python $REPO/scripts/timeline.py $DB 100 curvature_kernel > $OUT/timeline.txt
rm -rf $OUT/trace
cd $REPO
python scripts/framebench.py > $OUT/framebench.log 2>&1
ls -la $OUT
