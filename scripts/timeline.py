"""Prints the GPU timeline of a few bench steps from a rocprofv3 rocpd database (kernel + memory-copy trace):
start offset, duration and the idle gap before every kernel / copy. Usage: python scripts/timeline.py <results.db> [n_rows] [first_kernel] [fraction of the run at which the window starts]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = [(s, e, name) for name, s, e in db.execute("select name, start, end from kernels")]
try:
    cols = [d[1] for d in db.execute("pragma table_info(memory_copies)")]
    if cols:
        rows += [(s, e, "memcpy " + str(nm)) for nm, s, e in db.execute("select name, start, end from memory_copies")]
except Exception as ex:
    print("no memory_copies view:", ex)
rows.sort()
# take a window from the middle of the run
mid = len(rows) // 2
# align to the start of a step: find the next cell_count kernel (or the one named on the command line)
first = sys.argv[3] if len(sys.argv) > 3 else "cell_count"
mid = int(len(rows) * (float(sys.argv[4]) if len(sys.argv) > 4 else 0.8)) if len(sys.argv) > 3 else mid
while mid < len(rows) and first not in rows[mid][2]:
    mid += 1
win = rows[mid: mid + n]
t0 = win[0][0]
prev_end = None
gaps = []
for s, e, name in win:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    gaps.append(gap)
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:7.2f}  {name[:70]}")
    prev_end = max(prev_end or e, e)
print("sum of gaps in window: %.1f us over %d events" % (sum(gaps), len(win)))
