"""BASELINE config 1 (one 16-ring scan vs the 50 k map): ms per frame of map staging + 5 GN iterations, for the lane rule in force
(MLH_KNN_LANES pins it: 16, 8, or 816 = surf 8 / corner 16)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O, conftest
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
case = conftest._make_case(synth, "50k", 16, 1)
fs, fc = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
c = mla.Context(0)
c.map_set_pair(case["surf_map"], case["corner_map"])
c.features_set(mla.SURF, fs); c.features_set(mla.CORNER, fc)
opts = mla.default_opts()
def frame():
    c.map_rebuild(mla.ALL_KINDS)
    return c.gn_solve(case["p0"], 5, opts, want_stats=False)[0]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: frame()
c.synchronize(); t0 = time.perf_counter(); n = 500
for _ in range(n): pose = frame()
c.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / n
print(f"config 1: {len(fs)} + {len(fc)} features, lanes surf/corner {c.map_info(mla.SURF)['knn_lanes']}/{c.map_info(mla.CORNER)['knn_lanes']}: {ms:.4f} ms per frame (index rebuild + 5 GN iterations), pose {pose[:3]}")
