"""mlh_scan2map time for a library variant: MLOAM_HIP_LIB=... python scripts/exp/s2m_time.py"""
import importlib, os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
p0 = synth.perturbed_pose(gt, seed=43)
ctx = mla.Context(0)
ex = []
for s in scans:
    ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run(); ex.append(ctx.extract_fetch())
surf, corner = bench.fuse_features(synth, scans, ex)
ctx.map_set_pair(surf_map, corner_map)
ctx.features_set(mla.SURF, surf); ctx.features_set(mla.CORNER, corner)
opts = mla.default_opts(max_outer=2, max_lm_iterations=6)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    ctx.scan2map(p0, opts, want_stats=False)
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        ctx.scan2map(p0, opts, want_stats=False)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 300
    print(f"{os.environ.get('MLOAM_HIP_LIB', 'product'):50s} scan2map(2 outer x 6 LM) {1e3 * dt:.4f} ms")
