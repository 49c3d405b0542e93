"""Per-stage timestamps of fit_linearize_kernel on the bench workload (debug build, scripts/build_stageclock.sh).
Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock.py"""
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
n_tiles = (len(surf) + 255) // 256 + (len(corner) + 255) // 256
buf = (C.c_ulonglong * (n_tiles * 8))()
lib.mlh_debug_stage_clock.argtypes = [C.c_void_p, C.c_int]
assert lib.mlh_debug_stage_clock(buf, n_tiles * 8) == 0
tall = np.frombuffer(buf, np.uint64).reshape(n_tiles, 8).astype(np.int64)
t = tall[:, :5]
t0 = t[:, 0].min()
rel = (t - t0) * 0.01     # us
print("tiles", n_tiles, "surf", len(surf), "corner", len(corner))
names = ["start", "fit done", "eval done", "reduce done", "finish done"]
for i, nm in enumerate(names):
    print(f"{nm:12s} min {rel[:, i].min():7.2f} med {np.median(rel[:, i]):7.2f} max {rel[:, i].max():7.2f} us")
d = np.diff(rel, axis=1)
sub = (tall[:, 5:7] - t0) * 0.01
n_surf_tiles = (len(surf) + 255) // 256
for nm_, sl in (("surf tiles", slice(0, n_surf_tiles)), ("corner tiles", slice(n_surf_tiles, n_tiles))):
    a = rel[sl, 0]; b = sub[sl, 0]; c = sub[sl, 1]; d_ = rel[sl, 1]
    print(f"{nm_}: start->loads arrived med {np.median(b - a):.2f} max {(b - a).max():.2f}; loads->fit+gates done med {np.median(c - b):.2f} max {(c - b).max():.2f}; ->corr stored med {np.median(d_ - c):.2f} us")
for i, nm in enumerate(["fit", "eval", "reduce", "finish"]):
    print(f"stage {nm:7s} med {np.median(d[:, i]):6.2f} max {d[:, i].max():6.2f} us")
buf2 = (C.c_ulonglong * (4096 * 8))()
assert lib.mlh_debug_stage_clock(buf2, 4096 * 8) == 0
fin = (np.frombuffer(buf2, np.uint64).reshape(4096, 8).astype(np.int64)[4095, :3] - t0) * 0.01
print("finish (last workgroup): ticket won %.2f, partials summed %.2f, solved %.2f us" % tuple(fin))
last = np.argmax(rel[:, 4])
print("last tile", last, "stages", rel[last])

# ---- correspondence kernel
n_wg = min(4096, (len(surf) + 31) // 32 + (len(corner) + 31) // 32)
buf = (C.c_ulonglong * (n_wg * 8))()
lib.mlh_debug_stage_clock_knn.argtypes = [C.c_void_p, C.c_int]
assert lib.mlh_debug_stage_clock_knn(buf, n_wg * 8) == 0
t = np.frombuffer(buf, np.uint64).reshape(n_wg, 8).astype(np.int64)[:, :6]
ok = (t > 0).all(axis=1)        # workgroups whose thread 0 went through every stage (its feature had >= K candidates ...)
print("knn workgroups", n_wg, "with all stages", int(ok.sum()))
t0 = t[:, 0][t[:, 0] > 0].min()
rel = (t - t0) * 0.01
names = ["start", "transform", "run table", "candidates", "tournament", "gather+store"]
for i, nm in enumerate(names):
    col = rel[ok, i]
    print(f"{nm:12s} min {col.min():7.2f} med {np.median(col):7.2f} max {col.max():7.2f} us")
d = np.diff(rel[ok], axis=1)
for i, nm in enumerate(names[1:]):
    print(f"stage {nm:12s} med {np.median(d[:, i]):6.2f} max {d[:, i].max():6.2f} us")
print("start spread over all workgroups: max", rel[:, 0][t[:, 0] > 0].max())
order = np.argsort(-rel[:, 5])[:6]
print("slowest knn workgroups (stage end times, us):")
for w in order:
    print(" wg", int(w), np.round(rel[w], 2))
fast = np.argsort(rel[:, 5])[:3]
for w in fast:
    print(" fast wg", int(w), np.round(rel[w], 2))
print("finish-time percentiles 50/90/99/100:", np.round(np.percentile(rel[ok, 5], [50, 90, 99, 100]), 2))
