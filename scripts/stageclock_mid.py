"""What the mid launch of the device std::sort (stdsort.hip: stdsort_mid_kernel) does on the frame's thinning: per workgroup the range it took, its partitions and
their time (debug build, scripts/build_stageclock.sh). Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock_mid.py"""
import ctypes as C, importlib, os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
for e in ext: e[3:] /= np.linalg.norm(e[3:])
covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)]); meas = np.diag([0.0025] * 3)
ctx = mla.Context(0)
offs = np.cumsum([0] + [len(s.points) for s in scans])
pts = np.concatenate([s.points for s in scans])
st = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
en = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
ring_ofs = np.cumsum([0] + [s.n_rings for s in scans])
ctx.fuse_reset(); ctx.scan_upload(pts, st, en); ctx.extract_run(); ctx.extract_voxel_run(0.2)
for i in range(len(scans)): ctx.fuse_add_rings(ring_ofs[i], ring_ofs[i + 1], i, ext[i])
thin = lambda: ctx.downsample_current_scan_pair(ctx.fused_cloud(mla.SURF), ctx.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, True, 0.6)
lib = mla.load_library()
lib.mlh_debug_stage_clock_mid.argtypes = [C.c_void_p, C.c_int, C.c_int]
for _ in range(3): thin()
ctx.synchronize()
buf2 = (C.c_ulonglong * (1024 * 16))()
lib.mlh_debug_stage_clock_mid(buf2, 1024 * 16, 1)
thin(); ctx.synchronize()
lib.mlh_debug_stage_clock_mid(buf2, 1024 * 16, 0)
acc = np.frombuffer(buf2, np.uint64).reshape(1024, 16).astype(np.float64)
busy = np.nonzero(acc[:, 7] > 0)[0]
print("mid workgroups with a range:", len(busy))
for i in busy:
    a = acc[i]
    print(f"wg {i:3d}: range {int(a[7]):6d}  partitions {int(a[0]):3d} ({0.01 * a[1] / max(a[0], 1):5.2f} us each, mean size {a[2] / max(a[0], 1):7.0f})  heap sorts {int(a[3])} "
          f"({0.01 * a[4]:6.1f} us, {int(a[5])} elements)  total {0.01 * a[6]:6.1f} us")
