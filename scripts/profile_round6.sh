#!/bin/bash
# Round-6 evidence (run through gpurun from the repo root): the bench line as the driver runs it (--steps 20 --warmup 5) and at 200 steps, the same under
# rocprofv3 --kernel-trace --stats (pipelined and synchronous submission), three PMC passes, the step timeline; then the frame from C++ threads (one pipeline by
# stage, the estimator / mapper pair, K pipelines), its kernel statistics and timeline, the stage benches, the residency soak and the bounded soaks.
TAG=${1:-r06}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ -z "$ONLY_TRACES" ]; then
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_form.json 2> $OUT/bench.log
timeout -k 10 400 python bench.py --steps 200 --warmup 20 > $OUT/bench_line.json 2>> $OUT/bench.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --synchronous > $OUT/bench_synchronous.json 2>> $OUT/bench.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overlap-staging > $OUT/bench_no_overlap_staging.json 2>> $OUT/bench.log
fi
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-supplementary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_line_under_trace.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
  python $REPO/scripts/timeline.py $DB 70 pack_count_kernel 0.3 > $OUT/step_timeline.txt
fi
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH --synchronous --profile-events 0 > $OUT/bench_synchronous_under_trace.json 2> $OUT/trace_sync.log
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats_synchronous.txt
  python $REPO/scripts/timeline.py $DB 40 pack_count_kernel 0.3 > $OUT/step_timeline_synchronous.txt
fi
rm -rf $OUT/trace
PMCB="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-supplementary --profile-events 0 --synchronous"
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o pmc -- $PMCB > /dev/null 2> $OUT/pmc$i.log
  DBP=$(find $OUT/pmc$i -name '*.db' | head -1)
  [ -n "$DBP" ] && python $REPO/profiles/summarize_pmc.py $DBP _kernel > $OUT/pmc$i.txt
  rm -rf $OUT/pmc$i
done
cd $REPO
[ -n "$ONLY_TRACES" ] && { head -12 $OUT/kernel_stats.txt; head -10 $OUT/kernel_stats_synchronous.txt; exit 0; }
# the frame from C++ threads
D=/tmp/fb_in; python scripts/framebench_inputs.py $D > /dev/null 2>&1
{ m-loam_amd/host/framebench $D 100 single; m-loam_amd/host/framebench $D 100 all; m-loam_amd/host/framebench $D 100 all; MLH_HOST_WAIT=yield m-loam_amd/host/framebench $D 100 all; } > $OUT/framebench.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/ftrace -o trace -- $REPO/m-loam_amd/host/framebench $D 40 single > $OUT/framebench_under_trace.log 2> $OUT/ftrace.log)
DBF=$(find $OUT/ftrace -name '*.db' | head -1)
if [ -n "$DBF" ]; then
  python profiles/summarize_rocprof.py $DBF > $OUT/frame_kernel_stats.txt
  python scripts/timeline.py $DBF 95 curvature_kernel 0.80 > $OUT/frame_timeline.txt
fi
rm -rf $OUT/ftrace
timeout -k 10 200 python scripts/calibbench.py < /dev/null > $OUT/calibbench.txt 2>&1
timeout -k 10 200 python scripts/trackbench.py < /dev/null > $OUT/trackbench.txt 2>&1
{ for o in firing firing_rev shuffled; do echo "--- SEGBENCH_ORDER=$o"; SEGBENCH_ORDER=$o timeout -k 10 200 python scripts/segbench.py < /dev/null 2>&1 | grep -v amdgpu.ids; done; } > $OUT/segbench.txt 2>&1
{ for i in 1 2 3; do m-loam_amd/host/framebench $D 60 raw; done; } > $OUT/framebench_raw.txt 2>&1
timeout -k 10 200 python scripts/gfbench.py < /dev/null > $OUT/gfbench.txt 2>&1
timeout -k 10 300 python scripts/frontbench.py 40 > $OUT/frontbench.txt 2>/dev/null
REPS=150 timeout -k 10 200 python scripts/thinbench.py 2>/dev/null | tail -1 > $OUT/thinbench.txt
MLH_SS_MID_OFF=1 REPS=150 timeout -k 10 200 python scripts/thinbench.py 2>/dev/null | tail -1 >> $OUT/thinbench.txt
REPS=2 timeout -k 10 600 scripts/ab_lm.sh > $OUT/lm_schedule_ab.txt 2>&1
# residency: four contexts in four threads, 2 000 whole frames each; the masked-stream and fallback tests
{ MLOAM_RESIDENCY_FRAMES=2000 timeout -k 10 900 python -m pytest tests/test_gpu_residency.py -q -k four_contexts 2>&1 | tail -3; timeout -k 10 600 python -m pytest tests/test_gpu_residency.py -q 2>&1 | tail -3; } > $OUT/residency.txt 2>&1
{ timeout -k 10 300 python scripts/soak_schedule.py 40 21 2>&1 | tail -1; timeout -k 10 300 python scripts/soak_stdsort.py 300 21; timeout -k 10 300 python scripts/soak_parity_frontend.py 100 21 segment,rough,voxel,uct; timeout -k 10 300 python scripts/soak_parity.py 100 21; MLOAM_SCENE_FAMILY=hard timeout -k 10 300 python scripts/soak_parity.py 100 22 | tail -1; timeout -k 10 600 python scripts/soak_api.py 20 23 4 | tail -2; } > $OUT/soak.txt 2>&1
python - <<PY
import json
for n in ("bench_line_driver_form", "bench_line", "bench_synchronous", "bench_no_overlap_staging"):
    d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], (d.get("frame") or {}).get("ms_per_frame"), (d.get("frame") or {}).get("period_ms_two_contexts"), (d.get("frame") or {}).get("frames_per_s_at_K"))
PY
head -24 $OUT/kernel_stats.txt
grep -h knn_features $OUT/pmc*.txt
tail -n 4 $OUT/framebench.txt | cut -c1-600; tail -n 5 $OUT/segbench.txt; cat $OUT/residency.txt
