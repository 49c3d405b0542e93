#!/bin/bash
# quick check: gpu tests + one bench line, fused path vs MLH_FUSED=0
OUT=$PWD/gpurun_out/r02q
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --map-rebuild-only > $OUT/bench_fused.json 2> $OUT/bench.log
MLH_FUSED=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --map-rebuild-only > $OUT/bench_unfused.json 2>> $OUT/bench.log
python - <<'PY'
import json
for n in ("fused","unfused"):
    d=json.load(open(f"gpurun_out/r02q/bench_{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["kernel_us_per_launch"], d["config"]["n_valid_per_iter_surf_corner"], d["final_pose"][:3])
PY
