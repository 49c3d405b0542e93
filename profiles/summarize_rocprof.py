#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (default output of `rocprofv3 --kernel-trace --stats`) into the per-kernel summary
committed under profiles/. Usage: python profiles/summarize_rocprof.py <results.db> [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), max(vgpr_count), max(lds_size), "
                  "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
print(f"{'kernel':86s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'total_ms':>9s} {'pct':>6s} {'vgpr':>5s} {'lds':>7s} {'scr':>5s} {'grid':>8s} {'wg':>4s}")
for r in rows:
    print(f"{r[0][:86]:86s} {r[1]:6d} {r[2] / 1e3:8.2f} {r[3] / 1e3:8.2f} {r[4] / 1e3:8.2f} {r[5] / 1e6:9.3f} {100 * r[5] / tot:6.1f} "
          f"{r[6]:5d} {r[7]:7d} {r[8]:5d} {r[9]:8d} {r[10]:4d}")
