#!/bin/bash
# A/B of scan2map behind a good-feature selection, without statistics: the selected rows' LM loop as one launch (default) against the classic launches
python - <<'PY'
import importlib, os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.getcwd())
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
ctx.features_set(mla.SURF, surf); ctx.features_set(mla.CORNER, corner)
for method in ("rnd", "gd_fix"):
    opts = mla.default_opts(gf_method=mla.GF_METHODS[method], gf_ratio=0.2, gf_seed=3)
    for rep in range(2):
        for mode in ("0", "1"):
            os.environ["MLH_LM_CONSUMER"] = mode; os.environ["MLH_LM_LOOP"] = mode
            for _ in range(3): pose = ctx.scan2map(p0, opts, want_stats=False)[0]
            t = time.perf_counter(); n = 30
            for _ in range(n): pose = ctx.scan2map(p0, opts, want_stats=False)[0]
            print(method, "loop" if mode == "1" else "classic", round(1e3 * (time.perf_counter() - t) / n, 4), "ms", pose[:3])
PY
