#!/bin/bash
# GPU timeline of the C++ frame bench (m-loam_amd/host/framebench: both loops -- index rebuilt on the critical path, then the map staged beside the front end).
# Run through gpurun from the repo root; output in gpurun_out/prof_frame_cpp_$TAG/ -> copy into profiles/.
TAG=${1:-r03}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_frame_cpp_$TAG
mkdir -p $OUT
D=$(mktemp -d)
FRAMEBENCH_DEV_ONLY=1 FRAMEBENCH_KEEP_DIR=$D python scripts/framebench.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/trace -o trace -- $REPO/m-loam_amd/host/framebench $D 40 > $OUT/framebench_under_trace.log 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $REPO/scripts/timeline.py $DB 95 curvature_kernel 0.30 > $OUT/timeline_index_on_the_critical_path.txt
  python $REPO/scripts/timeline.py $DB 95 curvature_kernel 0.80 > $OUT/timeline_map_staged_beside_the_front_end.txt
fi
rm -rf $OUT/trace
cd $REPO
ls -la $OUT
