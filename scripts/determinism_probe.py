"""Hashes of every input bench.py builds (scene, maps, scans, GPU-extracted lists, fused features): run twice, diff the output."""
import hashlib, importlib, os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mla = importlib.import_module("m-loam_amd")
synth = importlib.import_module("m-loam_amd.synth")
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
print("cpus", os.cpu_count(), "boxes", h(sc.boxes), "surf_map", h(surf_map), "corner_map", h(corner_map), corner_map.shape)
for i, s in enumerate(scans):
    print("scan", i, h(s.points), h(s.scan_start), s.points.shape)
ctx = mla.Context(0)
ex = []
for s in scans:
    ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run(); e = ctx.extract_fetch(); ex.append(e)
    print("extract", {k: h(e[k]) for k in ("label", "sharp", "less_sharp", "flat", "less_flat_raw")})
surf, corner = bench.fuse_features(synth, scans, ex)
print("features", h(surf), h(corner), surf.shape, corner.shape)
