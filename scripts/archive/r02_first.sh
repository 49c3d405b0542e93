#!/bin/bash
# round-2 baseline on the new (keyframe corner map) workload: gpu tests, bench line, kernel trace
REPO=$PWD
OUT=$REPO/gpurun_out/r02a
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python bench.py --steps 100 --warmup 10 > $OUT/bench_line.json 2> $OUT/bench.log; tail -3 $OUT/bench.log; cat $OUT/bench_line.json
timeout 300 python bench.py --steps 100 --warmup 10 --map-rebuild-only --no-cpu-baseline > $OUT/bench_rebuild_only.json 2>> $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
rm -rf $OUT/trace
head -40 $OUT/kernel_stats.txt
