"""Per-stage timestamps of stdsort_leaf_kernel for the per-ring sort of one 2 x 64-ring extraction (debug build, scripts/build_stageclock.sh).
Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock_sort.py"""
import ctypes as C, importlib, os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
ctx = mla.Context(0)
offs = np.cumsum([0] + [len(s.points) for s in scans])
pts = np.concatenate([s.points for s in scans])
st = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
en = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
ctx.scan_upload(pts, st, en); ctx.extract_run()
lib = mla.load_library()
lib.mlh_debug_stage_clock_sort2.argtypes = [C.c_void_p, C.c_int, C.c_int]
for _ in range(2):
    ctx.extract_voxel_run(0.2)
ctx.synchronize()
buf2 = (C.c_ulonglong * (1024 * 16))()
lib.mlh_debug_stage_clock_sort2(buf2, 1024 * 16, 1)          # clear the accumulators
ctx.extract_voxel_run(0.2)
ctx.synchronize()
lib.mlh_debug_stage_clock_sort2(buf2, 1024 * 16, 0)
acc = np.frombuffer(buf2, np.uint64).reshape(1024, 16).astype(np.float64)[:128]
for cls, nm in enumerate(["<= 64", "<= 256", "longer"]):
    n = acc[:, cls].sum(); tk = acc[:, 3 + cls].sum()
    print(f"partitions of {nm:7s}: {n / 128:6.1f} per ring, {0.01 * tk / max(n, 1):5.2f} us each")
print(f"bookkeeping after a partition: {0.01 * acc[:, 6].sum() / acc[:, :3].sum():5.2f} us each; acquiring / waiting: {0.01 * acc[:, 7].sum() / 128:6.1f} us per ring summed over its 16 wavefronts")
buf = (C.c_ulonglong * (1024 * 8))()
lib.mlh_debug_stage_clock_sort.argtypes = [C.c_void_p, C.c_int]
assert lib.mlh_debug_stage_clock_sort(buf, 1024 * 8) == 0
t = np.frombuffer(buf, np.uint64).reshape(1024, 8).astype(np.int64)
busy = t[:, 7] > 0
print("leaf workgroups with a range:", int(busy.sum()), "range lengths min/med/max", int(t[busy, 7].min()), int(np.median(t[busy, 7])), int(t[busy, 7].max()))
t0 = t[:, 0][t[:, 0] > 0].min()
rel = (t[busy, :7] - t0) * 0.01
names = ["kernel start", "loaded + queue ready", "recursion done (this wave)", "... all waves", "insertion pass done (thread 0)", "... all threads", "stored"]
for i, nm in enumerate(names):
    print(f"{nm:34s} min {rel[:, i].min():7.2f} med {np.median(rel[:, i]):7.2f} max {rel[:, i].max():7.2f} us")
order = np.argsort(-rel[:, 2])
print("slowest rings (recursion done, us):", [(int(i), round(float(rel[i, 2]), 1)) for i in order[:12]])
print("fastest rings:", [(int(i), round(float(rel[i, 2]), 1)) for i in order[-6:]])
per_ring_parts = acc[:, :3].sum(axis=1)
print("partitions per ring for the slowest:", [(int(i), int(per_ring_parts[i]), round(0.01 * float(acc[i, 7]), 1)) for i in order[:12]])

print("heap sorts (depth budget used up): <= 64 elements %d in all rings, %.2f us each; longer: %d, %.2f us each; elements heap-sorted per ring: mean %.1f max %d"
      % (acc[:, 8].sum(), 0.01 * acc[:, 9].sum() / max(acc[:, 8].sum(), 1), acc[:, 10].sum(), 0.01 * acc[:, 11].sum() / max(acc[:, 10].sum(), 1), acc[:, 12].mean(), acc[:, 12].max()))
print("  slowest rings: (ring, long heap sorts, us in them, elements)", [(int(i), int(acc[i, 10]), round(0.01 * float(acc[i, 11]), 1), int(acc[i, 12])) for i in order[:8]])
