#!/bin/bash
# where the overlapped map staging call sits in the frame's front end (m-loam_amd/host/framebench.cpp, FB_STAGE_POS), and the staging stream's CU mask
cd "$(dirname "$0")/../.."
D=$(mktemp -d)
FRAMEBENCH_DEV_ONLY=1 FRAMEBENCH_KEEP_DIR=$D python scripts/framebench.py > /dev/null 2>&1
for rep in 1 2; do
for pos in 0 1 2 3; do
  echo -n "pos $pos half-CU staging: "; FB_STAGE_POS=$pos m-loam_amd/host/framebench $D 50 | tail -1 | cut -c1-200
  echo -n "pos $pos all-CU staging:  "; MLH_STAGE_CU_MASK=ffffffff,ffffffff FB_STAGE_POS=$pos m-loam_amd/host/framebench $D 50 | tail -1 | cut -c1-200
done
done
