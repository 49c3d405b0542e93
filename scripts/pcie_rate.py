"""PCIe-inclusive frame rate: the boundary hands over HOST feature buffers every frame (map staged once, as on a non-keyframe)."""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
ctx.map_set(mla.SURF, surf_map); ctx.map_set(mla.CORNER, corner_map)
opts = mla.default_opts()
def frame(upload):
    if upload:
        ctx.features_set(mla.SURF, surf); ctx.features_set(mla.CORNER, corner)
    ctx.map_rebuild(mla.ALL_KINDS)
    return ctx.gn_solve(p0, 5, opts, want_stats=False)[0]
for up in (False, True):
    for _ in range(20): frame(True)
    ctx.synchronize(); t = time.perf_counter(); n = 200
    for _ in range(n): frame(up)
    ctx.synchronize(); ms = 1e3 * (time.perf_counter() - t) / n
    print(f"features {'uploaded from host every frame' if up else 'resident'}: {ms:.4f} ms/frame, {(len(surf)+len(corner))*5/ms*1e3:.3e} features/s")
