#!/bin/bash
# A/B of library variants (scripts/build_variant.py) on the thinning stage, inside ONE gpurun call: arguments = variant names ("-" = the shipped library), alternated REPS times.
OUT=gpurun_out/ab_thin; mkdir -p $OUT
REPS_OUTER=${REPS_OUTER:-2}
for rep in $(seq 1 $REPS_OUTER); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then env -u MLOAM_HIP_LIB timeout 300 python scripts/thinbench.py 2>$OUT/err_$v.txt | tail -1
    else MLOAM_HIP_LIB=$PWD/m-loam_amd/lib_ab/$v/libmloam_hip.so timeout 300 python scripts/thinbench.py 2>$OUT/err_$v.txt | tail -1; fi
  done
done | tee $OUT/result.txt
