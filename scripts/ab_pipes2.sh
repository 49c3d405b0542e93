#!/bin/bash
# where K pipelines lose their scaling: threads of one process vs separate processes (the HIP runtime's locks are per process), per-stage host times
out=${1:-gpurun_out/pipes}; mkdir -p $out; d=/tmp/fb_in
[ -f $d/fb_pose.f64 ] || python scripts/framebench_inputs.py $d > /dev/null 2>&1
exe=m-loam_amd/host/framebench
{
for rep in 1 2; do
  echo "== threads of one process"; $exe $d 200 pipes 1,2,4,8
  for n in 2 4 8; do
    echo "== $n processes, one pipeline each (each line: that process's frames per second)"
    for i in $(seq $n); do $exe $d 400 pipes 1 > /tmp/p_$i.txt 2>&1 & done; wait
    for i in $(seq $n); do python - /tmp/p_$i.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["frames_per_s_at_K"]["1"]["frames_per_s"], d["frames_per_s_at_K"]["1"]["host_ms_per_frame"])
PY
    done
  done
done
} > $out/ab_pipes2.txt 2>&1
