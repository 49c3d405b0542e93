"""The thinning stage of a mapper frame by itself (mlh_downsample_current_scan_pair on the device-resident fused clouds of the 2 x 64-ring frame), for A/B runs:
ms per call over REPS calls (host clock around the call; the call returns when the thinned counts have arrived), the frame's other stages once, and a digest of the
result (counts + the pose scan2map reaches from those features: equal digests = the same features). MLOAM_HIP_LIB selects the library."""
import hashlib, importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
p0 = synth.perturbed_pose(gt, seed=43)
ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
for e in ext: e[3:] /= np.linalg.norm(e[3:])
covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
meas = np.diag([0.0025] * 3)
import torch
torch.cuda.init()
ctx = mla.Context(0)
ctx.map_set(mla.SURF, surf_map); ctx.map_set(mla.CORNER, corner_map)
opts = mla.default_opts(flags=mla.FLAG_WITH_UA)
both_pts = np.concatenate([s.points for s in scans])
offs = np.cumsum([0] + [len(s.points) for s in scans])
both_start = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
both_end = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
ring_ofs = np.cumsum([0] + [s.n_rings for s in scans])
REPS = int(os.environ.get("REPS", "200"))

def front():
    ctx.fuse_reset()
    ctx.scan_upload(both_pts, both_start, both_end); ctx.extract_run(); ctx.extract_voxel_run(0.2)
    for i in range(len(scans)): ctx.fuse_add_rings(ring_ofs[i], ring_ofs[i + 1], i, ext[i])

def thin():
    return ctx.downsample_current_scan_pair(ctx.fused_cloud(mla.SURF), ctx.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, True, 0.6)

front()
for _ in range(10): counts = thin()
ctx.synchronize()
ts = []
for _ in range(REPS):
    t0 = time.perf_counter(); thin(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
pose, _ = ctx.scan2map(p0, opts, want_stats=False)
# the whole frame, as framebench's device path
tt = {"front": 0.0, "thin": 0.0, "s2m": 0.0}
for it in range(30):
    t0 = time.perf_counter(); front(); t1 = time.perf_counter(); thin(); t2 = time.perf_counter(); ctx.map_rebuild(mla.ALL_KINDS); ctx.scan2map(p0, opts, want_stats=False); t3 = time.perf_counter()
    if it >= 5:
        tt["front"] += t1 - t0; tt["thin"] += t2 - t1; tt["s2m"] += t3 - t2
dig = hashlib.sha1(np.asarray(counts, np.int64).tobytes() + pose.tobytes()).hexdigest()[:12]
print(f"thin ms/call: median {np.median(ts):.4f} mean {ts.mean():.4f} min {ts.min():.4f}  |  frame: front {tt['front'] / 25 * 1e3:.3f} thin {tt['thin'] / 25 * 1e3:.3f} s2m {tt['s2m'] / 25 * 1e3:.3f}"
      f"  |  counts {counts} digest {dig}  lib {os.environ.get('MLOAM_HIP_LIB', 'default')}")
