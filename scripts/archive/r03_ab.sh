#!/bin/bash
# A/B inside one gpurun call: env settings given as arguments ("VAR=val VAR2=val2" per variant), alternated 3 times
mkdir -p gpurun_out/r03
i=0
for rep in 1 2 3; do
  for v in "$@"; do
    i=$((i+1))
    env $v timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/r03/ab_$i.json 2> gpurun_out/r03/ab_$i.err
    python - "$v" gpurun_out/r03/ab_$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["kernel_us_per_launch"]
    print(f"{sys.argv[1]:40s} ms/step {d['ms_per_step']:.4f}  all-bracketed {d['ms_per_step_all_kernels_bracketed']:.4f}  knn {k['knn_features (surf+corner)']}  fit {k['fit_linearize+gn_finish (surf+corner)']}  build {k['map_index_build (both maps, 4 launches)']}  roofline knn us {d['roofline']['avg_kernel_us']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace('.json', '.err')).read()[-600:])
PY
  done
done
