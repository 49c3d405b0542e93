"""BASELINE config 5: greedy good-feature selection (gd_fix, ratio 0.2) + uncertainty-weighted residuals, 2x64 rings, 1M map."""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench, oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "1M")
p0 = synth.perturbed_pose(gt, seed=43)
ctx = mla.Context(0)
ex = []
for s in scans:
    ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run(); ex.append(ctx.extract_fetch())
surf, corner = bench.fuse_features(synth, scans, ex)
ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
for e in ext: e[3:] /= np.linalg.norm(e[3:])
covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
meas = np.diag([0.0025] * 3)
ctx.map_set(mla.SURF, surf_map); ctx.map_set(mla.CORNER, corner_map)
fs = ctx.downsample_current_scan(mla.SURF, surf, 0.4, ext, covs, meas, True, 0.6)
fc = ctx.downsample_current_scan(mla.CORNER, corner, 0.2, ext, covs, meas, True, 0.6)
print("features surf/corner", len(fs), len(fc))
for method in (os.environ.get("GF_ONLY", "wo_gf,gd_fix,rnd,fps").split(",")):
    opts = mla.default_opts(flags=mla.FLAG_WITH_UA if hasattr(mla, "FLAG_WITH_UA") else 2, gf_method=mla.GF_METHODS[method], gf_ratio=0.2, gf_seed=7)
    for _ in range(2): ctx.map_rebuild(mla.ALL_KINDS); pose, st = ctx.scan2map(p0, opts)
    t = time.perf_counter(); n = 10
    for _ in range(n): ctx.map_rebuild(mla.ALL_KINDS); pose, st = ctx.scan2map(p0, opts)
    gpu_ms = 1e3 * (time.perf_counter() - t) / n
    ms_, mc_ = O.Map(surf_map), O.Map(corner_map)
    prm = O.mapper_params(with_ua=True, gf_method=method, gf_ratio=0.2, seed=7)
    t = time.perf_counter(); ref = O.scan2map(ms_, mc_, fs, fc, p0, prm); cpu_ms = 1e3 * (time.perf_counter() - t)
    print(f"{method:7s} scan2map GPU {gpu_ms:8.3f} ms  CPU oracle {cpu_ms:8.1f} ms (excl. kd-tree build)  sel {[ (s['n_surf'], s['n_corner']) for s in st]}  |dt| {np.linalg.norm(pose[:3]-ref['pose'][:3]):.2e}")
