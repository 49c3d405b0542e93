#!/bin/bash
# K independent frame pipelines / the estimator-mapper pair from C++ threads, under the runtime settings that could serialise them (one gpurun call)
# usage: scripts/ab_pipes.sh <out dir>
out=${1:-gpurun_out/pipes}; mkdir -p $out; d=/tmp/fb_in
python scripts/framebench_inputs.py $d > /dev/null 2>&1
exe=m-loam_amd/host/framebench
for rep in 1 2; do
  echo "== default (GPU_MAX_HW_QUEUES unset)"; $exe $d 100 all
  for q in 2 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q $exe $d 100 pipes; done
  echo "== MLH_STAGE_CU_MASK=ffffffff,ffffffff (no mask on the staging streams)"; MLH_STAGE_CU_MASK=ffffffff,ffffffff $exe $d 100 pipes
  echo "== MLH_HOST_WAIT=yield"; MLH_HOST_WAIT=yield $exe $d 100 all
done > $out/ab_pipes.txt 2>&1
nproc >> $out/ab_pipes.txt
