#!/bin/bash
# A/B of environment switches inside one gpurun call: bench.py's scan2map / frame legs for every "VAR=value" argument ("-" = nothing set), alternated REPS times
REPS=${REPS:-2}
for i in $(seq $REPS); do for e in "$@"; do
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  env $envs python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scan2map']
print('%-28s' % '$e', 'sync', s['ms_per_frame'], 'sync_maps_staged', s['ms_per_frame_synchronous_maps_staged'], 'pipelined', s['ms_per_frame_pipelined'], 'frame', d['frame']['ms_per_frame'], 'step', d['ms_per_step'])"
done; done
