"""Per-stage timestamps of match_fused_kernel on the bench workload (debug build, scripts/build_stageclock.sh).
Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock_fused.py"""
import ctypes as C, importlib, os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mla = importlib.import_module("m-loam_amd")
synth = importlib.import_module("m-loam_amd.synth")
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
opts = mla.default_opts()
for _ in range(5):
    ctx.gn_solve(p0, 5, opts, want_stats=False)
ctx.synchronize()
ctx.gn_solve(p0, 1, opts, want_stats=False)
ctx.synchronize()
lib = mla.load_library()
n_s, n_c = (len(surf) + 31) // 32, (len(corner) + 31) // 32
total = n_s + n_c
grid = ((total + 7) // 8) * 8
buf = (C.c_ulonglong * (grid * 8))()
lib.mlh_debug_stage_clock_knn.argtypes = [C.c_void_p, C.c_int]
assert lib.mlh_debug_stage_clock_knn(buf, grid * 8) == 0
t = np.frombuffer(buf, np.uint64).reshape(grid, 8).astype(np.int64)
per = (total + 7) >> 3
blk = np.arange(grid)
gtile = (blk & 7) * per + (blk >> 3)
live = gtile < total
kind = (gtile >= n_s).astype(int)
t0 = t[live, 0].min()
rel = (t - t0) * 0.01
names = ["start", "transform", "cell words", "walk", "tournament", "phase A + barrier", "phase B (fit+lin+reduce)", "finish"]
for k, nm in ((0, "surf"), (1, "corner")):
    sel = live & (kind == k)
    full = sel & (t[:, 2] > 0) & (t[:, 3] > 0) & (t[:, 4] > 0)
    print(f"--- {nm}: {int(sel.sum())} workgroups ({int(full.sum())} whose thread 0 searched)")
    for i, n_ in enumerate(names):
        col = rel[full if i in (2, 3, 4) else sel, i]
        print(f"  t[{i}] {n_:26s} min {col.min():7.2f} med {np.median(col):7.2f} p90 {np.percentile(col, 90):7.2f} max {col.max():7.2f} us")
    d56 = rel[sel, 6] - rel[sel, 5]
    print(f"  phase B duration med {np.median(d56):6.2f} p90 {np.percentile(d56, 90):6.2f} max {d56.max():6.2f} us")
buf2 = (C.c_ulonglong * (4096 * 8))()
lib.mlh_debug_stage_clock.argtypes = [C.c_void_p, C.c_int]
assert lib.mlh_debug_stage_clock(buf2, 4096 * 8) == 0
fin = (np.frombuffer(buf2, np.uint64).reshape(4096, 8).astype(np.int64)[4095, :3] - t0) * 0.01
print("finish (last workgroup): ticket won %.2f, partials summed %.2f, solved %.2f us" % tuple(fin))
print("kernel end (last stage 7):", rel[live, 7].max())
