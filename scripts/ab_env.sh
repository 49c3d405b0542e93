#!/bin/bash
# A/B of an environment switch on the bench line, inside ONE gpurun call, alternated: scripts/ab_env.sh VAR valueA valueB [reps] [bench args...]
# prints ms_per_step (pipelined), synchronous, frame, scan2map, and the index-build stage per run
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4
for rep in $(seq 1 $REPS); do
  for v in "$A" "$B"; do
    if [ "$v" = "-" ]; then r=$(env -u $VAR python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1); else r=$(env $VAR=$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1); fi
    echo "$r" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d.get('frame') or {}; s=d.get('scan2map') or {}
print('$VAR=$v', 'step', d['ms_per_step'], 'sync', d.get('ms_per_step_synchronous_submission'), 'frame', f.get('ms_per_frame'), 'stages', f.get('stages_ms_each_followed_by_a_wait'), 's2m', s.get('ms_per_frame'), s.get('ms_per_frame_pipelined'))"
  done
done
