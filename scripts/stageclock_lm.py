"""Per-stage timestamps of one Levenberg-Marquardt step launch (linearize_kernel<true> + fused lm_step_body) on the bench workload
(debug build, scripts/build_stageclock.sh). Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock_lm.py"""
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
for its in (2, 3):
    opts = mla.default_opts(max_outer=1, max_lm_iterations=its)
    for _ in range(5):
        ctx.scan2map(p0, opts, want_stats=False)
    ctx.synchronize()
    buf = (C.c_ulonglong * (4096 * 8))()
    assert lib.mlh_debug_stage_clock(buf, 4096 * 8) == 0
    a = np.frombuffer(buf, np.uint64).reshape(4096, 8).astype(np.int64)
    t = a[:n_tiles, :5]
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01
    print(f"--- last launch of a {its}-iteration LM run: tiles {n_tiles}")
    for i, nm in enumerate(["start", "(fit done)", "eval done", "reduce done", "finish done"]):
        print(f"{nm:12s} min {rel[:, i].min():7.2f} med {np.median(rel[:, i]):7.2f} max {rel[:, i].max():7.2f} us")
    fin = (a[4095, :3] - t0) * 0.01
    print("last workgroup: ticket won %.2f, partials summed %.2f, LM body done %.2f us" % tuple(fin))
