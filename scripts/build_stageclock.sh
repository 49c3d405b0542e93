#!/bin/bash
# Debug build with per-stage timestamps in the fit kernel -> m-loam_amd/lib/libmloam_hip_dbg.so (not the product library)
set -e
cd "$(dirname "$0")/../m-loam_amd/csrc"
mkdir -p ../lib/dbg
for f in capi grid match solver extract comm select voxel voxelgrid odom track frontend segment stdsort; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMLH_STAGE_CLOCK -I../../include -c $f.hip -o ../lib/dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmloam_hip_dbg.so ../lib/dbg/*.o -ldl
