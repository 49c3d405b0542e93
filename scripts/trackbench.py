"""Times LidarTracker::trackCloud (mlh_track_cloud) on two consecutive 64-ring synthetic scans, GPU vs the CPU oracle."""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc = synth.make_scene(seed=42, **synth.SCENE_PRESETS["500k"])
gt0 = synth.gt_body_pose()
motion = np.array([0.35, -0.12, 0.02, 0.0, 0.0, np.sin(np.deg2rad(0.75)), np.cos(np.deg2rad(0.75))])
from scipy.spatial.transform import Rotation as Rot
T1 = synth.pose_to_mat(gt0) @ synth.pose_to_mat(motion)
gt1 = np.concatenate([T1[:3, 3], Rot.from_matrix(T1[:3, :3]).as_quat()])
ctx = mla.Context(0)
feats = {}
for name, pose, seed in (("prev", gt0, 7), ("cur", gt1, 11)):
    s = synth.simulate_scan(sc, pose, synth.HERCULES_BODY_T_LASER[0], 64, seed=seed)
    begins = s.scan_start - 5
    for r in range(s.n_rings):
        e = begins[r + 1] if r + 1 < s.n_rings else len(s.points)
        s.points[begins[r]:e, 3] = r
    ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run(); ex = ctx.extract_fetch()
    lf = ctx.extract_voxel(0.2)
    feats[name] = dict(pts=s.points, ex=ex, lf=lf)
cl = np.ascontiguousarray(feats["prev"]["pts"][feats["prev"]["ex"]["less_sharp"]]); sl = np.ascontiguousarray(feats["prev"]["lf"][:, :4])
cs = np.ascontiguousarray(feats["cur"]["pts"][feats["cur"]["ex"]["sharp"]]); sf = np.ascontiguousarray(feats["cur"]["pts"][feats["cur"]["ex"]["flat"]])
print("prev corner/surf", len(cl), len(sl), "cur sharp/flat", len(cs), len(sf))
p0 = np.array([0, 0, 0, 0, 0, 0, 1.0])
def frame():
    ctx.track_set_prev(mla.CORNER, cl); ctx.track_set_prev(mla.SURF, sl); ctx.track_set_cur(mla.CORNER, cs); ctx.track_set_cur(mla.SURF, sf)
    return ctx.track_cloud(p0, want_stats=False)[0]
for _ in range(3): pose = frame()
t = time.perf_counter(); n = 20
for _ in range(n): pose = frame()
gpu_ms = 1e3 * (time.perf_counter() - t) / n
t = time.perf_counter(); ref = O.track_cloud(cl, sl, cs, sf, p0); cpu_ms = 1e3 * (time.perf_counter() - t)
print(f"trackCloud incl. staging + index build: GPU {gpu_ms:.3f} ms, CPU oracle {cpu_ms:.1f} ms; |dt| {np.linalg.norm(pose[:3]-ref['pose'][:3]):.2e}; motion error {np.linalg.norm(pose[:3]-motion[:3]):.3f} m")
# component timing
def tm(fn, n=20):
    fn(); ctx.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    ctx.synchronize(); return 1e3 * (time.perf_counter() - t) / n
print("set_prev corner %.3f ms, set_prev surf %.3f ms, set_cur corner %.3f, set_cur surf %.3f, track_cloud %.3f ms" % (
    tm(lambda: ctx.track_set_prev(mla.CORNER, cl)), tm(lambda: ctx.track_set_prev(mla.SURF, sl)),
    tm(lambda: ctx.track_set_cur(mla.CORNER, cs)), tm(lambda: ctx.track_set_cur(mla.SURF, sf)),
    tm(lambda: ctx.track_cloud(p0, want_stats=False))))

# the odometry front end of one LiDAR with device hand-overs: scan in -> extractCloud -> trackCloud against the previous frame -> this
# frame becomes the previous one; nothing but the raw scan and the pose crosses PCIe
sp, sc_ = feats["prev"], feats["cur"]
scan_prev = (sp["pts"], ) ; scan_cur = (sc_["pts"], )
def front_end(scan_pts, start, end, first=False):
    ctx.scan_upload(scan_pts, start, end); ctx.extract_run(); ctx.extract_voxel_run(0.2)
    pose = None
    if not first:
        ctx.track_set_from_scan(0)
        pose = ctx.track_cloud(p0, want_stats=False)[0]
    ctx.track_set_from_scan(1)
    return pose
ss = {}
for name, pose_, seed in (("prev", gt0, 7), ("cur", gt1, 11)):
    s_ = synth.simulate_scan(sc, pose_, synth.HERCULES_BODY_T_LASER[0], 64, seed=seed)
    ss[name] = (feats[name]["pts"], s_.scan_start, s_.scan_end)
front_end(*ss["prev"], first=True)
for _ in range(3): pose_dev = front_end(*ss["cur"]); front_end(*ss["prev"])
ctx.synchronize(); t = time.perf_counter()
for _ in range(10): pose_dev = front_end(*ss["cur"]); front_end(*ss["prev"])
ctx.synchronize()
print("odometry front end, device hand-overs (upload + extractCloud + trackCloud + hand-over): %.3f ms per LiDAR frame; pose vs host-staged path |dt| %.2e" % (
    1e3 * (time.perf_counter() - t) / 20, np.linalg.norm(pose_dev[:3] - pose[:3])))
if os.environ.get("TRACKBENCH_FRONT_END_ONLY"): sys.exit(0)

if os.environ.get("MLOAM_HIP_LIB"):
    import ctypes as C
    lib = mla.load_library()
    ctx.track_cloud(p0, want_stats=False); ctx.synchronize()
    nwg = (len(cs) + 3) // 4 + (len(sf) + 3) // 4
    buf = (C.c_ulonglong * (nwg * 4))()
    lib.mlh_debug_stage_clock_track.argtypes = [C.c_void_p, C.c_int]
    assert lib.mlh_debug_stage_clock_track(buf, nwg * 4) == 0
    t = np.frombuffer(buf, np.uint64).reshape(nwg, 4).astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01
    ok = (t > 0).all(axis=1)
    print("track_match workgroups", nwg, "complete", int(ok.sum()))
    for i, nm in enumerate(["start", "nn done", "walks done", "end"]):
        print(f"{nm:12s} med {np.median(rel[ok, i]):8.2f} max {rel[ok, i].max():8.2f} us")
    w = np.argsort(-rel[:, 3])[:5]
    print("slowest:", [(int(i), [round(float(x), 1) for x in rel[i]]) for i in w])
