"""One whole mapper frame, stage by stage: extractCloud for both LiDARs -> fusion (host glue) -> downsampleCurrentScan -> index build ->
scan2MapOptimization, GPU path vs CPU oracle, with HOST buffers at every hand-over (the reference's ROS nodes exchange host clouds)."""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench, oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
p0 = synth.perturbed_pose(gt, seed=43)
ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
for e in ext: e[3:] /= np.linalg.norm(e[3:])
covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
meas = np.diag([0.0025] * 3)
Ts = []
for i in range(2):
    T = np.eye(4); T[:3, :3] = synth.quat_to_rot(synth.HERCULES_BODY_T_LASER[i][:4]); T[:3, 3] = synth.HERCULES_BODY_T_LASER[i][4:7]; Ts.append(T)
import torch
torch.cuda.init()          # (before the library's first HIP call, as in bench.py: the other order leaves torch without a device)
ctx = mla.Context(0)
ctx.map_set(mla.SURF, surf_map); ctx.map_set(mla.CORNER, corner_map)
opts = mla.default_opts(flags=mla.FLAG_WITH_UA)

def fuse(lists):
    surf, corner = [], []
    for i, (pts, less_sharp, less_flat_ds) in enumerate(lists):
        for dst, xyz in ((corner, pts[less_sharp][:, :3]), (surf, less_flat_ds[:, :3])):
            a = np.empty((len(xyz), 4), np.float32)
            a[:, :3] = synth.transform_points(xyz, Ts[i]); a[:, 3] = i
            dst.append(a)
    return np.concatenate(surf), np.concatenate(corner)

def gpu_frame(t):
    lists = []
    t0 = time.perf_counter()
    for s in scans:
        ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run()
        ex = ctx.extract_fetch(); lf = ctx.extract_voxel(0.2)
        lists.append((s.points, ex["less_sharp"], lf))
    t1 = time.perf_counter()
    surf, corner = fuse(lists)
    t2 = time.perf_counter()
    ctx.downsample_current_scan(mla.SURF, surf, 0.4, ext, covs, meas, True, 0.6)
    ctx.downsample_current_scan(mla.CORNER, corner, 0.2, ext, covs, meas, True, 0.6)
    t3 = time.perf_counter()
    ctx.map_rebuild(mla.ALL_KINDS)
    pose, _ = ctx.scan2map(p0, opts, want_stats=False)
    t4 = time.perf_counter()
    for k, v in zip(("extract", "fuse(host)", "downsample", "scan2map"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): t[k] = t.get(k, 0.0) + v
    return pose, surf, corner

def gpu_frame_dev(t):
    """The same frame with every hand-over on the device: only the raw scans go in and the pose comes out."""
    t0 = time.perf_counter()
    ctx.fuse_reset()
    for i, s in enumerate(scans):
        ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run(); ctx.extract_voxel_run(0.2)
        ctx.fuse_add_scan(i, ext[i])
    t1 = time.perf_counter()
    if PAIR:
        ctx.downsample_current_scan_pair(ctx.fused_cloud(mla.SURF), ctx.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, True, 0.6)
    else:
        ctx.downsample_current_scan(mla.SURF, ctx.fused_cloud(mla.SURF), 0.4, ext, covs, meas, True, 0.6, fetch=False)
        ctx.downsample_current_scan(mla.CORNER, ctx.fused_cloud(mla.CORNER), 0.2, ext, covs, meas, True, 0.6, fetch=False)
    t3 = time.perf_counter()
    ctx.map_rebuild(mla.ALL_KINDS)
    pose, _ = ctx.scan2map(p0, opts, want_stats=False)
    t4 = time.perf_counter()
    for k, v in zip(("upload+extract+fuse", "downsample", "scan2map"), (t1 - t0, t3 - t1, t4 - t3)): t[k] = t.get(k, 0.0) + v
    return pose

# both LiDARs as ONE scan (rings back to back): one launch set extracts them together
both_pts = np.concatenate([s.points for s in scans])
offs = np.cumsum([0] + [len(s.points) for s in scans])
both_start = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
both_end = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
ring_ofs = np.cumsum([0] + [s.n_rings for s in scans])

def gpu_frame_dev1(t):
    t0 = time.perf_counter()
    ctx.fuse_reset()
    ctx.scan_upload(both_pts, both_start, both_end); ctx.extract_run(); ctx.extract_voxel_run(0.2)
    for i in range(len(scans)): ctx.fuse_add_rings(ring_ofs[i], ring_ofs[i + 1], i, ext[i])
    t1 = time.perf_counter()
    if PAIR:
        ctx.downsample_current_scan_pair(ctx.fused_cloud(mla.SURF), ctx.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, True, 0.6)
    else:
        ctx.downsample_current_scan(mla.SURF, ctx.fused_cloud(mla.SURF), 0.4, ext, covs, meas, True, 0.6, fetch=False)
        ctx.downsample_current_scan(mla.CORNER, ctx.fused_cloud(mla.CORNER), 0.2, ext, covs, meas, True, 0.6, fetch=False)
    t3 = time.perf_counter()
    ctx.map_rebuild(mla.ALL_KINDS)
    pose, _ = ctx.scan2map(p0, opts, want_stats=False)
    t4 = time.perf_counter()
    for k, v in zip(("upload+extract+fuse", "downsample", "scan2map"), (t1 - t0, t3 - t1, t4 - t3)): t[k] = t.get(k, 0.0) + v
    return pose

PAIR = False
for _ in range(3): gpu_frame_dev({})
td = {}
for _ in range(20): pose_dev = gpu_frame_dev(td)
print("GPU path, device-resident hand-overs, ms per frame:", {k: round(1e3 * v / 20, 3) for k, v in td.items()}, "total %.3f" % (1e3 * sum(td.values()) / 20))
for _ in range(3): gpu_frame_dev1({})
td1 = {}
for _ in range(20): pose_dev1 = gpu_frame_dev1(td1)
PAIR = True       # mlh_downsample_current_scan_pair: both kinds through one thinning pipeline
for _ in range(3): gpu_frame_dev1({})
td2 = {}
for _ in range(20): pose_dev2 = gpu_frame_dev1(td2)
# The same frame with the local map staged and indexed BESIDE the front end: the map is made of earlier keyframes, so its index does not wait for the scan
# (mlh_map_set_pair_overlapped with device-resident clouds: second stream, the other map set; enqueued after the front end's launches, before the first call
# that waits for them). The line above re-indexes between the thinning and the solve, on the frame's critical path.
surf_map_dev = torch.from_numpy(np.ascontiguousarray(surf_map, np.float32)).cuda(); corner_map_dev = torch.from_numpy(np.ascontiguousarray(corner_map, np.float32)).cuda()
def gpu_frame_dev2(t):
    t0 = time.perf_counter()
    ctx.fuse_reset()
    ctx.scan_upload(both_pts, both_start, both_end); ctx.extract_run(); ctx.extract_voxel_run(0.2)
    for i in range(len(scans)): ctx.fuse_add_rings(ring_ofs[i], ring_ofs[i + 1], i, ext[i])
    ctx.map_set_pair_overlapped(surf_map_dev, corner_map_dev)
    t1 = time.perf_counter()
    ctx.downsample_current_scan_pair(ctx.fused_cloud(mla.SURF), ctx.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, True, 0.6)
    t3 = time.perf_counter()
    pose, _ = ctx.scan2map(p0, opts, want_stats=False)
    t4 = time.perf_counter()
    for k, v in zip(("upload+extract+fuse+map staging", "downsample", "scan2map"), (t1 - t0, t3 - t1, t4 - t3)): t[k] = t.get(k, 0.0) + v
    return pose
for _ in range(3): gpu_frame_dev2({})
td3 = {}
for _ in range(20): pose_dev3 = gpu_frame_dev2(td3)
PAIR = False
print("GPU path, device-resident, one launch set, both kinds thinned in one pipeline, ms per frame:", {k: round(1e3 * v / 20, 3) for k, v in td2.items()}, "total %.3f" % (1e3 * sum(td2.values()) / 20),
      "same pose:", bool(np.array_equal(pose_dev2, pose_dev1)))
print("  the same with the local map staged and indexed beside the front end (second stream, other map set), ms per frame:", {k: round(1e3 * v / 20, 3) for k, v in td3.items()},
      "total %.3f" % (1e3 * sum(td3.values()) / 20), "same pose:", bool(np.array_equal(pose_dev3, pose_dev1)))
print("GPU path, device-resident, both LiDARs one launch set, ms per frame:", {k: round(1e3 * v / 20, 3) for k, v in td1.items()}, "total %.3f" % (1e3 * sum(td1.values()) / 20),
      "same pose as per-LiDAR launches:", bool(np.array_equal(pose_dev1, pose_dev)))
# Two pipelines on the one GPU, as the reference's estimator and mapper NODES are two processes: an estimator-side context extracts, fuses and thins frame k + 1
# while a mapper-side context indexes and solves frame k; the hand-over is mlh_features_copy (device to device; the reference's is a ROS message). Frame PERIOD,
# not latency: a frame still takes what the line above says.
import threading
ctxB = mla.Context(0)
ctxB.map_set(mla.SURF, surf_map); ctxB.map_set(mla.CORNER, corner_map)
N_PIPE = 60
ready = [threading.Semaphore(0) for _ in range(N_PIPE)]
copied = [threading.Semaphore(0) for _ in range(N_PIPE)]
poses_pipe = [None] * N_PIPE
def estimator_side():
    for k in range(N_PIPE):
        ctx.fuse_reset()
        ctx.scan_upload(both_pts, both_start, both_end); ctx.extract_run(); ctx.extract_voxel_run(0.2)
        for i in range(len(scans)): ctx.fuse_add_rings(ring_ofs[i], ring_ofs[i + 1], i, ext[i])
        fs, fc = ctx.fused_cloud(mla.SURF), ctx.fused_cloud(mla.CORNER)
        if k > 0: copied[k - 1].acquire()              # the mapper side has taken frame k - 1's features: this context's sets may be overwritten
        ctx.downsample_current_scan_pair(fs, fc, 0.4, 0.2, ext, covs, meas, True, 0.6)
        ready[k].release()
def mapper_side():
    for k in range(N_PIPE):
        ready[k].acquire()
        ctxB.features_copy_from(ctx, mla.SURF); ctxB.features_copy_from(ctx, mla.CORNER)
        copied[k].release()
        ctxB.map_rebuild(mla.ALL_KINDS)
        poses_pipe[k], _ = ctxB.scan2map(p0, opts, want_stats=False)
ta, tb = threading.Thread(target=estimator_side), threading.Thread(target=mapper_side)
t_pipe = time.perf_counter()
ta.start(); tb.start(); ta.join(); tb.join()
ctxB.synchronize()
t_pipe = time.perf_counter() - t_pipe
print("two contexts on the GPU (estimator side: upload + extract + fuse + thin; mapper side: index + scan2map; device-to-device hand-over), frame PERIOD: %.3f ms over %d frames"
      % (1e3 * t_pipe / N_PIPE, N_PIPE), " same pose as the single pipeline:", bool(all(np.array_equal(p, pose_dev2) for p in poses_pipe)))
ctxB.close()

# the same frame driven from C++ through the C-ABI (m-loam_amd/host/framebench.cpp): no interpreter between the dozen calls of a frame
import subprocess, tempfile
exe = os.path.join(ROOT, "m-loam_amd", "host", "framebench")
if os.path.exists(exe):
    import contextlib
    keep = os.environ.get('FRAMEBENCH_KEEP_DIR')      # (scripts/exp/fb_stage_pos.sh re-runs the executable on the same inputs)
    with (contextlib.nullcontext(keep) if keep else tempfile.TemporaryDirectory()) as d:
        both_pts.astype(np.float32).tofile(os.path.join(d, "fb_points.f32"))
        np.concatenate([both_start, both_end]).astype(np.int32).tofile(os.path.join(d, "fb_rings.i32"))
        np.asarray(ring_ofs, np.int32).tofile(os.path.join(d, "fb_ring_ofs.i32"))
        np.ascontiguousarray(ext, np.float64).tofile(os.path.join(d, "fb_ext.f64"))
        np.ascontiguousarray(covs, np.float64).tofile(os.path.join(d, "fb_covs.f64"))
        np.ascontiguousarray(meas, np.float64).tofile(os.path.join(d, "fb_meas.f64"))
        sm = np.ascontiguousarray(surf_map, np.float32); cmap = np.ascontiguousarray(corner_map, np.float32)
        assert sm.shape[1] == cmap.shape[1]
        sm.tofile(os.path.join(d, "fb_surf_map.f32")); cmap.tofile(os.path.join(d, "fb_corner_map.f32"))
        np.array([sm.shape[1] * 4, 1], np.int32).tofile(os.path.join(d, "fb_meta.i32"))
        np.ascontiguousarray(p0, np.float64).tofile(os.path.join(d, "fb_pose.f64"))
        r = subprocess.run([exe, d, "50"], capture_output=True, text=True, timeout=120)
        for line in (r.stdout.strip().splitlines() or [r.stderr.strip()]):
            print(line.split("  pose ")[0], " same pose as through ctypes (printed to 1e-9):",
                  bool(np.allclose([float(x) for x in line.split("  pose ")[1].split()], pose_dev2, rtol=0, atol=2e-9)) if "  pose " in line else line)
if os.environ.get('FRAMEBENCH_DEV_ONLY'): sys.exit(0)
for _ in range(3): gpu_frame({})
tg = {}; n = 20
for _ in range(n): pose, surf, corner = gpu_frame(tg)
print("GPU path, ms per frame:", {k: round(1e3 * v / n, 3) for k, v in tg.items()}, "total %.3f" % (1e3 * sum(tg.values()) / n))

tc = {}
t0 = time.perf_counter()
lists = []
for s in scans:
    ex = O.extract(s.points, s.scan_start, s.scan_end)
    lists.append((s.points, ex["less_sharp"], ex["less_flat_ds"]))
t1 = time.perf_counter()
surf_c, corner_c = fuse(lists)
t2 = time.perf_counter()
def cpu_downsample(member_order):
    """downsampleCurrentScan (lidar_mapper_keyframe.cpp:356-398): VoxelGridCovarianceMLOAM<PointI> (xyz mean, the LAST member's LiDAR id) ->
    evalPointUncertainty through that LiDAR's extrinsic -> trace gate. member_order 1 = point-index order inside a voxel (the HIP path's
    rule), 0 = the order libstdc++'s unstable std::sort leaves (the reference)."""
    res = []
    for cloud, leaf in ((surf_c, 0.4), (corner_c, 0.2)):
        ds = O.voxel_grid_mloam_plain(cloud, leaf, member_order)
        out = np.zeros((len(ds), 11), np.float32); out[:, :4] = ds
        for lid in range(2):
            m = ds[:, 3] == lid
            R = synth.quat_to_rot(ext[lid][3:])
            sel = ((ds[m, :3].astype(np.float64) - ext[lid][:3]) @ R).astype(np.float32)
            c = O.eval_point_uncertainty(sel, ext[lid], covs[lid], meas)
            out[m, 4:10] = np.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], axis=1)
        out[:, 10] = out[:, 4] + out[:, 7] + out[:, 9]
        res.append(out[out[:, 10] <= 0.6])
    return res
feats = cpu_downsample(0)
t3 = time.perf_counter()
ms_, mc_ = O.Map(surf_map), O.Map(corner_map)
tk = ms_.rebuild_seconds() + mc_.rebuild_seconds()
ref = O.scan2map(ms_, mc_, feats[0], feats[1], p0, O.mapper_params(with_ua=True))
t4 = time.perf_counter()
print("CPU oracle, ms per frame:", {"extract": round(1e3 * (t1 - t0), 1), "fuse(host)": round(1e3 * (t2 - t1), 1), "downsample": round(1e3 * (t3 - t2), 1),
      "kd-tree build": round(1e3 * tk, 1), "scan2map": round(1e3 * (t4 - t3) - 1e3 * tk, 1)}, "total %.1f" % (1e3 * (t4 - t0)))
print("pose agreement |dt| %.2e m (host hand-over) %.2e m (device hand-over)  [voxel members in the reference's std::sort order on both sides: the default]" % (np.linalg.norm(pose[:3] - ref["pose"][:3]), np.linalg.norm(pose_dev[:3] - ref["pose"][:3])))
f1 = cpu_downsample(1)
ref1 = O.scan2map(O.Map(surf_map), O.Map(corner_map), f1[0], f1[1], p0, O.mapper_params(with_ua=True))
ctx.set_voxel_member_order(False)
for _ in range(3): gpu_frame({})
tr = {}
for _ in range(n): pose_r, _, _ = gpu_frame(tr)
PAIR = True
for _ in range(3): gpu_frame_dev1({})
tr2 = {}
for _ in range(20): pose_r2 = gpu_frame_dev1(tr2)
PAIR = False
ctx.set_voxel_member_order(True)
print("GPU path with mlh_set_voxel_member_order(0) (point-index member order, no host pass), ms per frame:", {k: round(1e3 * v / n, 3) for k, v in tr.items()}, "total %.3f" % (1e3 * sum(tr.values()) / n),
      "pose vs CPU leg in that order |dt| %.2e m" % np.linalg.norm(pose_r[:3] - ref1["pose"][:3]))
print("  the same, device-resident, one launch set, both kinds thinned in one pipeline:", {k: round(1e3 * v / 20, 3) for k, v in tr2.items()}, "total %.3f" % (1e3 * sum(tr2.values()) / 20),
      "pose vs CPU leg in that order |dt| %.2e m" % np.linalg.norm(pose_r2[:3] - ref1["pose"][:3]))
print("effect of the member order on this frame: features surf/corner %d/%d (reference order) vs %d/%d (point-index order), pose moves %.2e m" %
      (len(feats[0]), len(feats[1]), len(f1[0]), len(f1[1]), np.linalg.norm(ref1["pose"][:3] - ref["pose"][:3])))
