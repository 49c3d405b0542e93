#!/bin/bash
# Round-2 evidence (run through gpurun from the repo root): gpu tests, bench line, rocprofv3 kernel trace, three PMC passes.
TAG=${1:-r02a}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench_line.json 2> $OUT/bench.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --map-rebuild-only > $OUT/bench_rebuild_only.json 2>> $OUT/bench.log
MLH_FUSED=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_fused_kernel.json 2>> $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_line_under_trace.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name '*.db' | head -1)
[ -n "$DB" ] && python $REPO/profiles/summarize_rocprof.py $DB > $OUT/kernel_stats.txt
PMCB="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-events 0"
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o pmc -- $PMCB > /dev/null 2> $OUT/pmc$i.log
  DBP=$(find $OUT/pmc$i -name '*.db' | head -1)
  [ -n "$DBP" ] && python $REPO/profiles/summarize_pmc.py $DBP _kernel > $OUT/pmc$i.txt
done
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
cd $REPO
python - <<PY
import json
for n in ("bench_line","bench_rebuild_only","bench_fused_kernel"):
    d=json.load(open("$OUT/%s.json"%n))
    print(n, d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["roofline"]["frac"], d["config"]["n_valid_per_iter_surf_corner"][0], d["config"]["input_sha1"])
PY
[ -f $OUT/kernel_stats.txt ] && head -30 $OUT/kernel_stats.txt
