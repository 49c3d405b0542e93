#!/bin/bash
# A/B of the scan2map Levenberg-Marquardt schedule inside one gpurun call: bench.py's scan2map legs with MLH_LM_CONSUMER=0 / 1, alternated REPS times
REPS=${REPS:-2}
for i in $(seq $REPS); do for e in MLH_LM_CONSUMER=0 MLH_LM_CONSUMER=1; do
  env $e $EXTRA_ENV python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scan2map']
print('$e', 'sync', s['ms_per_frame'], 'sync_maps_staged', s['ms_per_frame_synchronous_maps_staged'], 'pipelined', s['ms_per_frame_pipelined'], 'frame', d['frame']['ms_per_frame'], 'step', d['ms_per_step'], {k: s[k] for k in s if 'lm' in k or 'iter' in k})"
done; done
