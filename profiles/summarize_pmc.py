#!/usr/bin/env python
"""Per-kernel means of the PMC counters in a rocprofv3 rocpd database (one --pmc pass).
Usage: python profiles/summarize_pmc.py <results.db> [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in db.execute("pragma table_info(pmc_events)")]
# pmc_events view: one row per (dispatch, counter)
name_col = "counter_name" if "counter_name" in cols else ("name" if "name" in cols else None)
val_col = "counter_value" if "counter_value" in cols else ("value" if "value" in cols else None)
kcol = "kernel_name" if "kernel_name" in cols else None
if not (name_col and val_col):
    print("unexpected pmc_events schema:", cols)
    sys.exit(1)
if kcol is None:
    q = (f"select k.name, p.{name_col}, count(*), avg(p.{val_col}), sum(p.{val_col}) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
         f"group by k.name, p.{name_col} order by k.name")
else:
    q = f"select {kcol}, {name_col}, count(*), avg({val_col}), sum({val_col}) from pmc_events group by {kcol}, {name_col} order by {kcol}"
print(f"{'kernel':86s} {'counter':24s} {'dispatches':>10s} {'mean/dispatch':>16s}")
for k, c, n, a, s in db.execute(q):
    if flt in k:
        print(f"{k[:86]:86s} {c:24s} {n:10d} {a:16.1f}")
