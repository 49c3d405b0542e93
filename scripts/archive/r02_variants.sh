#!/bin/bash
# bench line per library variant (MLOAM_HIP_LIB): kernel durations + ms/step
mkdir -p gpurun_out/var
for v in "$@"; do
  lib=m-loam_amd/lib/libmloam_hip$v.so
  MLOAM_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/var/bench$v.json 2> gpurun_out/var/bench$v.log
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/var/bench{v}.json"))
print(v or "base", d["ms_per_step"], d["kernel_us_per_launch"], d["final_pose"][:2])
PY
done
