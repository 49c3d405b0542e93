"""Per-stage timestamps of label_kernel (debug build). Usage: MLOAM_HIP_LIB=m-loam_amd/lib/libmloam_hip_dbg.so python scripts/stageclock_label.py"""
import ctypes as C, importlib, os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mla = importlib.import_module("m-loam_amd")
synth = importlib.import_module("m-loam_amd.synth")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
ctx = mla.Context(0)
s = scans[0]
ctx.scan_upload(s.points, s.scan_start, s.scan_end)
for _ in range(3):
    ctx.extract_run()
ctx.synchronize()
lib = mla.load_library()
R = len(s.scan_start)
buf = (C.c_ulonglong * (R * 8))()
lib.mlh_debug_stage_clock_label.argtypes = [C.c_void_p, C.c_int]
assert lib.mlh_debug_stage_clock_label(buf, R * 8) == 0
t = np.frombuffer(buf, np.uint64).reshape(R, 8).astype(np.int64)[:, :6]
ok = (t > 0).all(axis=1)
rel = (t - t[ok, 0].min()) * 0.01
names = ["start", "ring loaded", "gap bits + keys", "sorted", "walks done", "written back"]
for i, nm in enumerate(names):
    print(f"{nm:16s} min {rel[ok, i].min():7.2f} med {np.median(rel[ok, i]):7.2f} max {rel[ok, i].max():7.2f} us")
d = np.diff(rel[ok], axis=1)
for i, nm in enumerate(names[1:]):
    print(f"stage {nm:16s} med {np.median(d[:, i]):6.2f} max {d[:, i].max():6.2f} us")
