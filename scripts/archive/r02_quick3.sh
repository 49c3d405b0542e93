#!/bin/bash
OUT=$PWD/gpurun_out/r02q
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --map-rebuild-only > $OUT/bench_$name.json 2>> $OUT/bench.log; }
run fused_strided MLH_FUSED=1
run fused_consec MLH_FUSED=1 MLH_FUSED_STRIDED=0
run unfused MLH_FUSED=0
python - <<'PY'
import json
for n in ("fused_strided","fused_consec","unfused"):
    d=json.load(open(f"gpurun_out/r02q/bench_{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["config"]["n_valid_per_iter_surf_corner"][0], d["final_pose"][:3], d.get("scan2map",{}).get("ms_per_frame"))
PY
