#!/bin/bash
# Round-5 evidence (run through gpurun from the repo root): bench line (default = pipelined + overlapped staging), the same under rocprofv3
# --kernel-trace --stats, a second trace with synchronous submission (kernels one after the other: the per-kernel durations without the staging
# stream's contention), three PMC passes, the step timeline, then the frame / calibration / tracker / segmenter benches.
TAG=${1:-r05}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ -z "$ONLY_TRACES" ]; then
timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench_line.json 2> $OUT/bench.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --synchronous > $OUT/bench_synchronous.json 2>> $OUT/bench.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overlap-staging > $OUT/bench_no_overlap_staging.json 2>> $OUT/bench.log
fi
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-supplementary"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_line_under_trace.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
  python $REPO/scripts/timeline.py $DB 70 pack_check_kernel 0.3 > $OUT/step_timeline.txt
fi
rm -rf $OUT/trace
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH --synchronous --profile-events 0 > $OUT/bench_synchronous_under_trace.json 2> $OUT/trace_sync.log
DB=$(find $OUT/trace -name '*.db' | head -1)
[ -n "$DB" ] && python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats_synchronous.txt
rm -rf $OUT/trace
PMCB="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-supplementary --profile-events 0 --synchronous"
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o pmc -- $PMCB > /dev/null 2> $OUT/pmc$i.log
  DBP=$(find $OUT/pmc$i -name '*.db' | head -1)
  [ -n "$DBP" ] && python $REPO/profiles/summarize_pmc.py $DBP _kernel > $OUT/pmc$i.txt
  rm -rf $OUT/pmc$i
done
cd $REPO
[ -n "$ONLY_TRACES" ] && { head -12 $OUT/kernel_stats.txt; head -10 $OUT/kernel_stats_synchronous.txt; exit 0; }
timeout 300 python scripts/framebench.py < /dev/null > $OUT/framebench.txt 2>&1
timeout 200 python scripts/calibbench.py < /dev/null > $OUT/calibbench.txt 2>&1
timeout 200 python scripts/trackbench.py < /dev/null > $OUT/trackbench.txt 2>&1
timeout 200 python scripts/segbench.py < /dev/null > $OUT/segbench.txt 2>&1
timeout 200 python scripts/gfbench.py < /dev/null > $OUT/gfbench.txt 2>&1
REPS=150 timeout 200 python scripts/thinbench.py 2>/dev/null | tail -1 > $OUT/thinbench.txt
MLH_SS_WIDE_OFF=1 REPS=150 timeout 200 python scripts/thinbench.py 2>/dev/null | tail -1 >> $OUT/thinbench.txt
MLH_THIN_SLOTS_FIRST=1 MLH_SS_WIDE_OFF=1 REPS=150 timeout 200 python scripts/thinbench.py 2>/dev/null | tail -1 >> $OUT/thinbench.txt
# the frame from C++ under the tracer: per-kernel stats and the timeline of one frame
D=$(mktemp -d)
FRAMEBENCH_DEV_ONLY=1 FRAMEBENCH_KEEP_DIR=$D python scripts/framebench.py > /dev/null 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/ftrace -o trace -- $REPO/m-loam_amd/host/framebench $D 40 > $OUT/framebench_under_trace.log 2> $OUT/ftrace.log)
DBF=$(find $OUT/ftrace -name '*.db' | head -1)
if [ -n "$DBF" ]; then
  python profiles/summarize_rocprof.py $DBF > $OUT/frame_kernel_stats.txt
  python scripts/timeline.py $DBF 95 curvature_kernel 0.80 > $OUT/frame_timeline.txt
fi
rm -rf $OUT/ftrace
# the scan2map Levenberg-Marquardt schedules side by side (classic launches | consumer-side launches | the loop as one launch), and the loop kernel's stage clocks
REPS=2 timeout 600 scripts/ab_lm.sh > $OUT/lm_schedule_ab.txt 2>&1
[ -f m-loam_amd/lib/libmloam_hip_dbg.so ] && MLOAM_HIP_LIB=$PWD/m-loam_amd/lib/libmloam_hip_dbg.so timeout 200 python scripts/stageclock_loop.py 2>/dev/null | tail -10 > $OUT/stageclock_loop.txt
{ timeout 300 python scripts/soak_schedule.py 40 21 2>&1 | tail -1; timeout 300 python scripts/soak_stdsort.py 300 21; timeout 300 python scripts/soak_parity_frontend.py 100 21 segment,rough,voxel,uct; timeout 300 python scripts/soak_parity.py 100 21; } > $OUT/soak.txt 2>&1
python - <<PY
import json
for n in ("bench_line", "bench_synchronous", "bench_no_overlap_staging"):
    d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"])
PY
head -24 $OUT/kernel_stats.txt
head -16 $OUT/kernel_stats_synchronous.txt
grep -h knn_features $OUT/pmc*.txt
tail -n 5 $OUT/framebench.txt; tail -n 5 $OUT/segbench.txt
