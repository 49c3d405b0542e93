"""per-kernel event times of mlh_gn_solve (5 iterations) for a library variant: MLOAM_HIP_LIB=... python scripts/exp/fit_time.py  (timing only)"""
import importlib, os, sys, warnings, time
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
opts = mla.default_opts()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    ctx.gn_solve(p0, 5, opts, want_stats=False)
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(400):
    ctx.gn_solve(p0, 5, opts, want_stats=False)
ctx.synchronize(); dt = (time.perf_counter() - t0) / 400
print(f"{os.environ.get('MLOAM_HIP_LIB', 'product')[-28:]:28s} gn_solve(5) {1e3 * dt:.4f} ms (synchronous call, no events)")
for mode in ("gn_solve", "begin+end"):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(400):
        if mode == "gn_solve":
            ctx.gn_solve(p0, 5, opts, want_stats=False)
        else:
            ctx.gn_solve_begin(p0, 5, opts); ctx.gn_solve_end()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 400
    print(f"  {mode:10s} {1e3 * dt:.4f} ms per solve (MLH_GN_FINAL_DEFER={os.environ.get('MLH_GN_FINAL_DEFER', '1')})")
