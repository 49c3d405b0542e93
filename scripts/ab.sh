#!/bin/bash
# A/B inside ONE gpurun call (same box, same clocks): every argument is one variant's environment ("VAR=val VAR2=val2", or "-" for none), the variants
# are alternated REPS times (default 3). BENCH_ARGS adds bench.py flags. Prints one line per run; JSON lines land in gpurun_out/ab/.
OUT=gpurun_out/ab
mkdir -p $OUT
REPS=${REPS:-3}
i=0
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    i=$((i+1))
    vv="$v"; [ "$v" = "-" ] && vv=""
    env $vv timeout 300 python bench.py --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline $BENCH_ARGS > $OUT/ab_$i.json 2> $OUT/ab_$i.err
    python - "$v" $OUT/ab_$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["kernel_us_per_launch"]
    print(f"{sys.argv[1]:36s} ms/step {d['ms_per_step']:.4f} sync {d.get('ms_per_step_synchronous_submission', 0):.4f} bracketed {d['ms_per_step_all_kernels_bracketed']:.4f}  knn {k['knn_features (surf+corner)']}  fit {k['fit_linearize+gn_finish (surf+corner)']}  build {k['map_index_build (both maps, 4 launches)']}  s2m {d.get('scan2map', {}).get('ms_per_frame')}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace('.json', '.err')).read()[-600:])
PY
  done
done
