#!/bin/bash
# Collects the evidence behind bench.py's roofline object on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> per-kernel durations
#   2. three separate --pmc passes (kernel-trace only, no other trace domains) -> HBM bytes, L2 hit rate, wave occupancy
# Outputs land in gpurun_out/prof_$TAG/ ; copy the summaries into profiles/.
TAG=${1:-r01c}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_line_under_trace.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
PMCB="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-events 0"
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o pmc -- $PMCB > /dev/null 2> $OUT/pmc$i.log
  DBP=$(find $OUT/pmc$i -name '*.db' | head -1)
  python $REPO/profiles/summarize_pmc.py $DBP _kernel > $OUT/pmc$i.txt
done
cd $REPO
python bench.py --steps 200 --warmup 20 > $OUT/bench_line.json 2> $OUT/bench.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dense-features > $OUT/bench_dense.json 2>> $OUT/bench.log
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
ls -la $OUT
