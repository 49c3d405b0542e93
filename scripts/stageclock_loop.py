"""Per-stage timestamps of the THIRD iteration of lm_loop_kernel (the whole Levenberg-Marquardt loop of an outer iteration in one launch) on the bench workload
(debug build, scripts/build_stageclock.sh). Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock_loop.py"""
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
lib = mla.load_library()
lib.mlh_debug_stage_clock.argtypes = [C.c_void_p, C.c_int]
n_tiles = (len(surf) + 255) // 256 + (len(corner) + 255) // 256
opts = mla.default_opts(max_outer=1)
for _ in range(5):
    ctx.scan2map(p0, opts, want_stats=False)
ctx.synchronize()
buf = (C.c_ulonglong * (4096 * 8))()
assert lib.mlh_debug_stage_clock(buf, 4096 * 8) == 0
a = np.frombuffer(buf, np.uint64).reshape(4096, 8).astype(np.int64)
t = a[:n_tiles, :6]
t0 = t[:, 0].min()
rel = (t - t0) * 0.01
print(f"--- third iteration of the loop kernel: tiles {n_tiles}; us from the first workgroup's loop top")
for i, nm in enumerate(["loop top", "eval done", "record stored", "barrier passed", "records summed", "LM step done"]):
    print(f"{nm:16s} min {rel[:, i].min():7.2f} med {np.median(rel[:, i]):7.2f} max {rel[:, i].max():7.2f} us")
d = np.diff(rel, axis=1)
print("per stage (median over the workgroups):", " | ".join(f"{x:.2f}" for x in np.median(d, axis=0)))
lib.mlh_debug_stage_clock_step.argtypes = [C.c_void_p, C.c_int]
b2 = (C.c_ulonglong * 16)()
assert lib.mlh_debug_stage_clock_step(b2, 16) == 0
st = np.frombuffer(b2, np.uint64).astype(np.int64)[:8]
print("inside the LM step (workgroup 0, last step of the run; us): " + " | ".join(f"{nm} {(st[i + 1] - st[i]) * 0.01:.2f}" for i, nm in enumerate(
    ["state -> registers, stop tests", "accept: radius, gradient max-norm (pose_plus)", "proposal: rows, diagonal", "Cholesky", "substitutions, model cost change", "pose_plus", "state stored"])))
