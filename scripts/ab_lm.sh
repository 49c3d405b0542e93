#!/bin/bash
# A/B of the scan2map Levenberg-Marquardt schedule inside one gpurun call: classic launches, consumer-side launches, the loop as one launch; alternated REPS times
REPS=${REPS:-2} exec scripts/ab_env.sh "MLH_LM_CONSUMER=0" "MLH_LM_LOOP=0" "-"
