#!/bin/bash
# Frame size vs the two kernels of a Gauss-Newton iteration (Next 6): bench.py on 1..4 LiDARs with the mapper's thinning (BASELINE's kind of
# frame) and without it (every less-flat / less-sharp point a query), one row each; PMC bytes (FETCH_SIZE, WRITE_SIZE: separate passes) for the 2 x 64 frame and
# the densest point. Run through gpurun from the repo root; writes gpurun_out/sweep_<tag>/feature_sweep.txt.
TAG=${1:-r04}
REPO=$PWD
OUT=$REPO/gpurun_out/sweep_$TAG
mkdir -p $OUT
row() {  # $1 = label, rest = bench flags
  local label=$1; shift
  timeout 400 python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python - "$label" $OUT/$label.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r, k, c = d["roofline"], d["kernel_us_per_launch"], d["config"]
    so = r.get("search_only_launch") or {}
    pre = next((v for kk, v in k.items() if "behind" in kk and v), 0.0)
    first = next((v for kk, v in k.items() if "chained frame" in kk and v), 0.0)
    print(f"{sys.argv[1]:14s} features {c['features_surf'] + c['features_corner']:7d} (surf {c['features_surf']:6d} corner {c['features_corner']:6d})  ms/step {d['ms_per_step']:.4f}  "
          f"knn cold {so.get('avg_kernel_us') or k['knn_features (surf+corner)']:8.2f} us  knn+finish {pre:8.2f} us  knn first-of-frame {first:8.2f} us  fit {k['fit_linearize+gn_finish (surf+corner)']:8.2f} us  "
          f"frac {r['frac']:.3f}  unavoidable {r['unavoidable_frac']:.3f}  search-only frac {so.get('frac', float('nan')):.3f}  bytes/launch {r['algorithmic_bytes_per_launch'] / 1e6:7.1f} MB  C-bar {r['mean_candidates_per_feature']}")
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e), open(sys.argv[2].replace('.json', '.err')).read()[-400:])
PY
}
{
echo "# scripts/feature_sweep.sh (one MI355X, 500k map): python bench.py --steps 60 --warmup 5 --no-cpu-baseline --lidars N [--dense-features]"
echo "# frac = SURVEY 8(d) 27-cell bytes / duration of the dominant correspondence launch (iterations >= 1: search behind the previous iteration's finish) / 8 TB/s;"
echo "# unavoidable = the cells the 5th-neighbour ball touches; search-only = iteration 0's launch (no prologue, no bound)"
for n in 1 2 3 4; do row thin_${n}x64 --lidars $n; done
for n in 1 2 4; do row dense_${n}x64 --lidars $n --dense-features; done
} | tee $OUT/feature_sweep.txt
cd /tmp && export TMPDIR=/tmp
for label in "thin_2x64 --lidars 2" "dense_4x64 --lidars 4 --dense-features"; do
  set -- $label; name=$1; shift
  i=0
  for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${name}_$i -o pmc -- python $REPO/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-supplementary --profile-events 0 --synchronous "$@" > /dev/null 2> $OUT/pmc_${name}_$i.log
    DBP=$(find $OUT/pmc_${name}_$i -name '*.db' | head -1)
    [ -n "$DBP" ] && python $REPO/profiles/summarize_pmc.py $DBP knn_features_kernel > $OUT/pmc_${name}_$i.txt
    rm -rf $OUT/pmc_${name}_$i
  done
  echo "# PMC, $label (mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, gfx950 correction: FETCH_SIZE x 2):" | tee -a $OUT/feature_sweep.txt
  cat $OUT/pmc_${name}_1.txt $OUT/pmc_${name}_2.txt 2>/dev/null | grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS" | tee -a $OUT/feature_sweep.txt
done
