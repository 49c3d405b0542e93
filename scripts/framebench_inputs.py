"""Writes the bench workload (BASELINE config 2) as the files m-loam_amd/host/framebench reads: python scripts/framebench_inputs.py <dir>"""
import importlib, os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
synth = importlib.import_module("m-loam_amd.synth")
d = sys.argv[1]
os.makedirs(d, exist_ok=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
p0 = synth.perturbed_pose(gt, seed=43)
ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:len(scans)]
for e in ext:
    e[3:] /= np.linalg.norm(e[3:])
covs = np.stack([np.zeros((6, 6))] + [np.diag([0.0025] * 3 + [0.00030461] * 3)] * (len(scans) - 1))
meas = np.diag([0.0025] * 3)
offs = np.cumsum([0] + [len(s.points) for s in scans])
all_pts = np.concatenate([s.points for s in scans])
all_start = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
all_end = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
ring_ofs = np.cumsum([0] + [s.n_rings for s in scans])
np.ascontiguousarray(all_pts, np.float32).tofile(os.path.join(d, "fb_points.f32"))
np.concatenate([all_start, all_end]).astype(np.int32).tofile(os.path.join(d, "fb_rings.i32"))
np.asarray(ring_ofs, np.int32).tofile(os.path.join(d, "fb_ring_ofs.i32"))
np.ascontiguousarray(ext, np.float64).tofile(os.path.join(d, "fb_ext.f64"))
np.ascontiguousarray(covs, np.float64).tofile(os.path.join(d, "fb_covs.f64"))
np.ascontiguousarray(meas, np.float64).tofile(os.path.join(d, "fb_meas.f64"))
sm, cm = np.ascontiguousarray(surf_map, np.float32), np.ascontiguousarray(corner_map, np.float32)
sm.tofile(os.path.join(d, "fb_surf_map.f32")); cm.tofile(os.path.join(d, "fb_corner_map.f32"))
np.array([sm.shape[1] * 4, 1], np.int32).tofile(os.path.join(d, "fb_meta.i32"))
np.ascontiguousarray(p0, np.float64).tofile(os.path.join(d, "fb_pose.f64"))
print("wrote", d)
