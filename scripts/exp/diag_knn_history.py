import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest, oracle as O, torch
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
caseA = conftest._make_case(synth, "50k", 16, 1)
featA = conftest.features_from_extraction(synth, caseA["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
A = (caseA["surf_map"], caseA["corner_map"])
Ad = tuple(torch.from_numpy(x).cuda() for x in A)
q = featA[1][:2000]
from scipy.spatial import cKDTree
d, i = cKDTree(A[1][:, :3].astype(np.float64)).query(q[:, :3].astype(np.float64), k=5)
print("brute force rows 839/840:", i[839], d[839] ** 2, i[840], d[840] ** 2)
print("query 839", q[839], "map bounds", A[1][:, :3].min(0), A[1][:, :3].max(0), "surf bounds", A[0][:, :3].min(0), A[0][:, :3].max(0))
def show(tag, r):
    n_missing = 0
    for row in range(len(q)):
        want = [(ii, dd) for ii, dd in zip(i[row], d[row] ** 2) if dd < 0.999]
        got = [(ii, dd) for ii, dd in zip(r[0][row], r[1][row]) if dd < 0.999]
        if [w[0] for w in want] != [g[0] for g in got]: n_missing += 1
    print(tag, "rows 839/840:", r[0][839], r[1][839], r[0][840], r[1][840], "| rows whose in-radius neighbours differ from brute force:", n_missing)
for how in ("host", "two", "dev", "two_rev"):
    c = mla.Context(0)
    if how == "host": c.map_set_pair(*A)
    elif how == "two": c.map_set(mla.SURF, A[0]); c.map_set(mla.CORNER, A[1])
    elif how == "two_rev": c.map_set(mla.CORNER, A[1]); c.map_set(mla.SURF, A[0])
    elif how == "dev": c.map_set_pair(*Ad)
    show("fresh " + how, c.knn(mla.CORNER, q)); print("   ", c.map_info(mla.CORNER))
    c.close()
