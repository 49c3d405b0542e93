"""Timings of the §8 rows that bench.py / framebench.py / trackbench.py / gfbench.py / voxbench.py do not cover, GPU next to the CPU oracle:
  a19  LidarPureOdom{PlaneNorm,Edge}Factor::Evaluate for a whole optimisation window (one launch vs one call per factor)
  a18  evalPointUncertainty over a feature cloud
  f1   keyframe clouds -> cloudUCTAssociateToMap -> VoxelGridCovarianceMLOAM -> map index (the mapper's local map on a keyframe), host buffers"""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
ctx = mla.Context(0)
rng = np.random.default_rng(31)

def tm(fn, n=20):
    fn(); ctx.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    ctx.synchronize(); return 1e3 * (time.perf_counter() - t) / n

def rand_pose(scale):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    return np.concatenate([rng.uniform(-scale, scale, 3), q])

# ---- a19: window of 4 frames x 2 LiDARs, 60 k factors (estimator.cpp:733-780 adds one factor per selected feature per frame)
n_frames, n_ext, n = 4, 2, 60000
pivot = rand_pose(20.0); frames = np.stack([rand_pose(20.0) for _ in range(n_frames)]); exts = np.stack([rand_pose(1.0) for _ in range(n_ext)])
types = rng.integers(0, 2, n).astype(np.int32); points = rng.uniform(-40, 40, (n, 3)); coeffs = np.zeros((n, 6))
v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
c = rng.uniform(-40, 40, (n, 3))
pl = types == 0
coeffs[pl, :3] = v[pl]; coeffs[pl, 3] = rng.uniform(-5, 5, pl.sum())
coeffs[~pl, :3] = c[~pl] + 0.1 * v[~pl]; coeffs[~pl, 3:] = c[~pl] - 0.1 * v[~pl]
fi = rng.integers(0, n_frames, n).astype(np.int32); ei = rng.integers(0, n_ext, n).astype(np.int32); sq = rng.uniform(0.3, 1.0, n)
ctx.pure_odom_set(types, points, coeffs, fi, ei, sq)
g_all = tm(lambda: ctx.pure_odom_evaluate(pivot, frames, exts))
g_res = tm(lambda: ctx.pure_odom_evaluate(pivot, frames, exts, want_jacobians=False))
t = time.perf_counter(); rr, JJ = O.pure_odom_eval_batch(types, points, coeffs, sq, fi, ei, pivot, frames, exts); cpu = 1e3 * (time.perf_counter() - t)
r, J = ctx.pure_odom_evaluate(pivot, frames, exts)
print(f"a19 pure-odom window, {n} factors: GPU {g_all:.3f} ms with the 3x(1x7) Jacobians copied back ({n * 21 * 8 / 1e6:.1f} MB), {g_res:.3f} ms residuals only; "
      f"CPU oracle {cpu:.1f} ms; max |dr| {np.abs(r - rr).max():.1e} max |dJ| {np.abs(J - JJ).max():.1e}")

# ---- a18: evalPointUncertainty over 40 k features of 2 LiDARs
ext = np.array([np.concatenate([r_[4:7], r_[:4]]) for r_ in synth.HERCULES_BODY_T_LASER])[:2]
for e in ext: e[3:] /= np.linalg.norm(e[3:])
covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)]); meas = np.diag([0.0025] * 3)
m = 40000
feat = np.zeros((m, 4), np.float32); feat[:, :3] = rng.uniform(-50, 50, (m, 3)); feat[:, 3] = rng.integers(0, 2, m)
g_u = tm(lambda: ctx.point_uncertainty(feat, ext, covs, meas, 0.6))
t = time.perf_counter()
for lid in range(2):
    mm = feat[:, 3] == lid
    O.eval_point_uncertainty(np.ascontiguousarray(feat[mm, :3]), ext[lid], covs[lid], meas)
cpu_u = 1e3 * (time.perf_counter() - t)
print(f"a18 evalPointUncertainty, {m} points: GPU {g_u:.3f} ms (host buffers in and out), CPU oracle {cpu_u:.1f} ms")

# ---- f1: the local map on a keyframe: 20 keyframes x 25 k points -> associate -> thin at 0.4 m -> index
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc = synth.make_scene(seed=42, **synth.SCENE_PRESETS["500k"])
    surf_map, corner_map = synth.sample_maps(sc, seed=42)
K = 20
base = surf_map[:, :3]
kfs = []
for k in range(K):
    sel = base[k::K][:25000]
    pts = np.zeros((len(sel), 11), np.float32); pts[:, :3] = sel + rng.normal(0, 0.03, sel.shape).astype(np.float32); pts[:, 3] = rng.integers(0, 2, len(sel))
    kfs.append(pts)
A = rng.normal(size=(6, 6)); cov_global = A @ A.T * 2e-6
poses = [np.array([0.1 * k, -0.05 * k, 0.0, 0, 0, 0, 1.0]) for k in range(K)]
def gpu_map():
    acc = np.concatenate([ctx.cloud_uct_associate_to_map(kf, p, cov_global, ext, covs, meas, True, 0.6) for kf, p in zip(kfs, poses)])
    ds = ctx.voxel_filter(acc, 0.4, 0.6)
    ctx.map_set(mla.SURF, ds)
    return ds
g_m = tm(gpu_map, 5)
ds = gpu_map()
t = time.perf_counter()
acc = np.concatenate([O.cloud_uct_associate_to_map(kf, p, cov_global, ext, covs, meas, True, 0.6) for kf, p in zip(kfs, poses)])
t1 = time.perf_counter(); dsr = O.voxel_grid_cov(acc, 0.4, 0.6); t2 = time.perf_counter(); mp = O.Map(np.ascontiguousarray(dsr[:, :3])); tk = mp.rebuild_seconds(); t3 = time.perf_counter()
print(f"f1 local map from {K} keyframes ({sum(len(k) for k in kfs)} points -> {len(ds)} map points), host buffers at every step: GPU {g_m:.2f} ms; "
      f"CPU oracle associate {1e3 * (t1 - t):.0f} + voxel filter {1e3 * (t2 - t1):.0f} + kd-tree {1e3 * tk:.0f} ms; same map size: {len(ds) == len(dsr)}")
