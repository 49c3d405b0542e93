#!/bin/bash
# A/B of the scan2map Levenberg-Marquardt schedule inside one gpurun call, alternated REPS times: the classic launches (the step in the last workgroup of the launch
# that evaluated), the consumer-side launches (the step in every workgroup of the next launch), the loop as one launch behind a grid barrier, and the loop as one
# launch with tagged records summed by polling (the default). Prints scan2map (synchronous / pipelined) and the frame per run.
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
  for cfg in "MLH_LM_CONSUMER=0" "MLH_LM_LOOP=0" "MLH_LOOP_TAGGED=0" "DEFAULT=1"; do
    env $cfg python bench.py --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d.get('frame') or {}; s=d.get('scan2map') or {}
print('%-20s' % '$cfg', 'scan2map ms per frame', s.get('ms_per_frame'), 'pipelined', s.get('ms_per_frame_pipelined'), '| frame', f.get('ms_per_frame'), 'its scan2map stage', (f.get('stages_ms_each_followed_by_a_wait') or {}).get('scan2map'))"
  done
done
