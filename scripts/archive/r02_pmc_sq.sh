#!/bin/bash
# SQ counters of the match kernels (fused and two-kernel path): is the launch issue-bound or latency-bound?
REPO=$PWD
OUT=$REPO/gpurun_out/r02pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCB="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-events 0 --map-rebuild-only"
for mode in 1 0; do
  MLH_FUSED=$mode rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/p$mode -o pmc -- $PMCB > /dev/null 2> $OUT/p$mode.log
  DBP=$(find $OUT/p$mode -name '*.db' | head -1)
  python $REPO/profiles/summarize_pmc.py $DBP _kernel > $OUT/sq_fused$mode.txt
  MLH_FUSED=$mode rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/q$mode -o pmc -- $PMCB > /dev/null 2> $OUT/q$mode.log
  DBP=$(find $OUT/q$mode -name '*.db' | head -1)
  python $REPO/profiles/summarize_pmc.py $DBP _kernel > $OUT/sq2_fused$mode.txt
  rm -rf $OUT/p$mode $OUT/q$mode
done
grep -E "match_fused|knn_features|fit_linearize" $OUT/*.txt
