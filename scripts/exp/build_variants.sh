#!/bin/bash
# experiment libraries: match.hip compiled with -DMLH_EXP=<n> (timing-only variants, results are wrong), the other objects from the product build
set -e
cd "$(dirname "$0")/../../m-loam_amd"
mkdir -p lib/exp
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMLH_EXP=$n -c csrc/match.hip -o lib/exp/match_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls build/*.o | grep -v match.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/exp/libmloam_hip_x$n.so $objs lib/exp/match_$n.o -ldl
done
ls -la lib/exp/*.so
