#!/bin/bash
# scan2map (the reference's per-frame call) per library variant: bench's supplementary scan2map object + gfbench wo_gf (with_ua)
for v in "$@"; do
  lib=m-loam_amd/lib/libmloam_hip$v.so
  MLOAM_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v' or 'base', 'scan2map ms', d['scan2map']['ms_per_frame'], 'step', d['ms_per_step'])"
  MLOAM_HIP_LIB=$PWD/$lib timeout 300 python scripts/gfbench.py 2>&1 | grep -E "^wo_gf" | cut -c1-40
done
