#!/bin/bash
# per-launch durations of the device std::sort inside one frame (rocprofv3 kernel trace of scripts/framebench.py, device-resident legs only)
REPO=$PWD; OUT=$REPO/gpurun_out/sortlv; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf $OUT/trace
  MLOAM_HIP_LIB=$REPO/$lib FRAMEBENCH_DEV_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $REPO/scripts/framebench.py > /dev/null 2> $OUT/trace.log
  DB=$(find $OUT/trace -name '*.db' | head -1)
  python - "$DB" "$lib" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select k.name, d.start, d.end from kernels k join (select * from rocpd_kernel_dispatch) d on 1=0").fetchall() if False else None
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
cols = [d[1] for d in db.execute(f"pragma table_info({view})")]
q = db.execute(f"select name, start, end from {view} order by start").fetchall()
seq = [(n, (e - s) / 1000.0) for n, s, e in q]
# the last complete thinning pipeline: from the last stdsort_init_kernel to the leaf after it
idx = [i for i, (n, _) in enumerate(seq) if "stdsort_init_kernel" in n]
i0 = idx[-3]
out = []
for n, d in seq[i0:]:
    if "stdsort" in n: out.append(round(d, 1))
    if "stdsort_leaf" in n: break
print(sys.argv[2], "init, big levels, leaf (us):", out, "sum %.1f" % sum(out))
PY
done
