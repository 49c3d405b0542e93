"""What ONE rank's two kernels take when it owns 1/N of the frame's features (bench workload, round-robin ownership, no communicator: the sums are partial and the
solve meaningless -- timing only). Answers what sharding can buy per iteration if the exchange were free."""
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
opts = mla.default_opts()
for n in (1, 2, 4, 8):
    ctx.shard_set_features(n, 0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        ctx.gn_solve(p0, 1, opts, want_stats=False)
    ctx.synchronize()
    ctx.profile_enable((1 << mla.K_KNN) | (1 << mla.K_FIT)); ctx.profile_reset()
    for _ in range(300):
        ctx.gn_solve(p0, 1, opts, want_stats=False)          # one iteration from p0: every launch does the first iteration's work
    ctx.synchronize()
    a = ctx.profile_get(mla.K_KNN); b = ctx.profile_get(mla.K_FIT)
    ctx.profile_enable(0)
    print(f"1/{n} of the features on this rank: correspondence kernel {1e3 * a[0] / a[1]:6.2f} us, fit + finish {1e3 * b[0] / b[1]:6.2f} us")
