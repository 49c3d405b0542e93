import sqlite3, glob, sys
f = glob.glob('/root/repo/gpurun_out/ps/out/*.db')[0]
cur = sqlite3.connect(f).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
ss = [(s, e, n) for n, s, e in rows if 'stdsort' in n]
idx = [i for i, r in enumerate(ss) if 'init' in r[2]]
for k in (-2, -6):
    seq = ss[idx[k]:idx[k + 1]]
    print("launches", len(seq), "span us", (seq[-1][1] - seq[0][0]) / 1e3, [round((e - b) / 1e3, 1) for b, e, n in seq])
