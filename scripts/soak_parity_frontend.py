"""Randomised parity soak of the rows around the solver, HIP path against the oracle (not part of the test suite):
  track     LidarTracker::trackCloud on random motions (0-0.8 m, 0-4 deg) of random scenes: correspondences counts, LM iterations, terminations, pose 1e-9
  segment   ImageSegmenter::segmentCloud on random raw clouds (16 / 32 / 64 rings, clutter 0-30 %, shuffled / ring-major / firing order): every output bit
  voxel     VoxelGridCovarianceMLOAM plain + covariance branches and pcl::VoxelGrid on random clouds (leaf 0.1-1.0, a fraction of the points snapped onto voxel
            faces / repeated): every output bit
  select    goodFeatureMatching rnd / fps / gd_fix / gd_float (random ratio up to 0.9, seed) on random scenes: identical selections, H 1e-9
  rough     extractCloud + segmentCloud on scans with a sensor's artefacts (synth.roughen_scan: azimuth sectors missing, near-range returns < 1 m, rings with
            < 12 points, an empty ring): every output bit
usage: python scripts/soak_parity_frontend.py [trials] [seed] [families, comma separated]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest, oracle as O
from scipy.spatial.transform import Rotation as Rot
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
families = (sys.argv[3] if len(sys.argv) > 3 else "track,segment,voxel,select,odom_select,uct,rough").split(",")
rng = np.random.default_rng(seed)
O.build()
ctx = None if os.environ.get("SOAK_DRY") else mla.Context(0)
ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
t_all = time.time()


def pose_err(a, b):
    # (rotation: 2 |q_a -+ q_b| -- the arccos of a dot product one ulp below 1 already reads 3e-8)
    return max(float(np.linalg.norm(a[:3] - b[:3])), 2 * min(float(np.linalg.norm(a[3:] - b[3:])), float(np.linalg.norm(a[3:] + b[3:]))))


def ring_tagged(scn):
    ring = np.zeros(len(scn.points), np.float32)
    begins = scn.scan_start - 5
    for r in range(scn.n_rings):
        e = begins[r + 1] if r + 1 < scn.n_rings else len(scn.points)
        ring[begins[r]:e] = r
    scn.points[:, 3] = ring
    return scn


if "track" in families:
    t0 = time.time(); n_lm = 0
    for trial in range(trials):
        sseed = int(rng.integers(1, 10 ** 6))
        sc = synth.make_scene(seed=sseed, **synth.SCENE_PRESETS["50k"])
        gt0 = synth.gt_body_pose()
        dt, dy = float(rng.uniform(0, 0.8)), float(rng.uniform(0, 4.0))
        d = rng.normal(size=3); d[2] *= 0.1; d *= dt / np.linalg.norm(d)
        motion = np.concatenate([d, Rot.from_euler("z", dy, degrees=True).as_quat()])
        T1 = synth.pose_to_mat(gt0) @ synth.pose_to_mat(motion)
        gt1 = np.concatenate([T1[:3, 3], Rot.from_matrix(T1[:3, :3]).as_quat()])
        rings = int(rng.choice([16, 32]))
        s0 = ring_tagged(synth.simulate_scan(sc, gt0, synth.HERCULES_BODY_T_LASER[0], rings, seed=sseed + 1))
        s1 = ring_tagged(synth.simulate_scan(sc, gt1, synth.HERCULES_BODY_T_LASER[0], rings, seed=sseed + 2))
        e0, e1 = O.extract(s0.points, s0.scan_start, s0.scan_end), O.extract(s1.points, s1.scan_start, s1.scan_end)
        cl, sl = np.ascontiguousarray(s0.points[e0["less_sharp"]]), np.ascontiguousarray(e0["less_flat_ds"][:, :4])
        cs, sf = np.ascontiguousarray(s1.points[e1["sharp"]]), np.ascontiguousarray(s1.points[e1["flat"]])
        ctx.track_set_prev(mla.CORNER, cl); ctx.track_set_prev(mla.SURF, sl)
        ctx.track_set_cur(mla.CORNER, cs); ctx.track_set_cur(mla.SURF, sf)
        pose, stats = ctx.track_cloud(ident)
        ref = O.track_cloud(cl, sl, cs, sf, ident)
        what = f"track trial {trial}: scene {sseed}, {rings} rings, motion {dt:.2f} m / {dy:.1f} deg"
        for s, o in zip(stats, ref["outer"]):
            if (s["n_corner"], s["n_surf"], s["lm_iterations"], s["termination"]) != (o["n_corner"], o["n_surf"], o["lm_iterations"], o["termination"]):
                raise SystemExit(f"TRACK COUNTS {what}: {(s['n_corner'], s['n_surf'], s['lm_iterations'], s['termination'])} vs {(o['n_corner'], o['n_surf'], o['lm_iterations'], o['termination'])}")
            n_lm += s["lm_iterations"]
        if pose_err(pose, ref["pose"]) > 1e-9:
            raise SystemExit(f"TRACK POSE {what}: {pose_err(pose, ref['pose']):.2e}")
        if not np.array_equal(ctx.track_cloud(ident, want_stats=False)[0], pose):
            raise SystemExit(f"TRACK LEAN PATH {what}")
    print(f"track: {trials} random motions: correspondence counts, {n_lm} LM iterations and terminations equal, poses within 1e-9  [{time.time() - t0:.0f} s]", flush=True)

if "segment" in families:
    t0 = time.time(); n_pts = 0
    scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    for trial in range(trials):
        rings = int(rng.choice([16, 32, 64]))
        clutter = float(rng.choice([0.0, 0.05, 0.15, 0.3]))
        order = str(rng.choice(["shuffled", "ring_major", "firing"]))
        sseed = int(rng.integers(1, 10 ** 6))
        body = synth.gt_body_pose().copy(); body[:2] += rng.uniform(-3, 3, 2)
        s = synth.simulate_scan(scn, body, synth.HERCULES_BODY_T_LASER[int(rng.integers(2))], rings, seed=sseed)
        r2 = np.random.default_rng(sseed)
        pts = s.points.copy()
        pts[:, 3] = r2.uniform(0.0, 0.9, len(pts)).astype(np.float32)
        m = r2.random(len(pts)) < clutter
        pts[m, :3] *= r2.uniform(0.5, 1.3, (int(m.sum()), 1)).astype(np.float32)
        if order == "shuffled":
            pts = pts[r2.permutation(len(pts))]
        elif order == "firing":
            az = np.arctan2(pts[:, 1], pts[:, 0]); az = np.mod(az - az[len(pts) // 3], 2 * np.pi)
            pts = pts[np.argsort(az, kind="stable")]
        pts = np.ascontiguousarray(pts)
        flag = bool(rng.integers(2))
        prm = O.seg_params(vertical_scans=rings, segment_flag=flag)
        ref = O.segment_cloud(pts, prm)
        got = ctx.segment_cloud(pts, vertical_scans=rings, segment_flag=int(flag))
        what = f"segment trial {trial}: {rings} rings, clutter {clutter}, {order}, seed {sseed}, flag {flag}"
        for k in ("cloud", "outlier"):
            if got[k].shape != ref[k].shape or not np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)):
                raise SystemExit(f"SEGMENT {k} {what}: shapes {got[k].shape} / {ref[k].shape}")
        if not (np.array_equal(got["scan_start"], ref["scan_start"]) and np.array_equal(got["scan_end"], ref["scan_end"])):
            raise SystemExit(f"SEGMENT scan info {what}")
        n_pts += len(pts)
    print(f"segment: {trials} random raw clouds ({n_pts} points): ring-major cloud, ScanInfo and outlier cloud equal bit for bit  [{time.time() - t0:.0f} s]", flush=True)

if "rough" in families:
    # scans with a sensor's artefacts (synth.roughen_scan: azimuth sectors missing, near-range returns below a metre, rings with < 12 points, an empty ring):
    # extractCloud bit for bit on the ring-major form, and ImageSegmenter on the same points as an unordered raw cloud
    t0 = time.time(); n_pts = 0
    scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    for trial in range(trials):
        rings = int(rng.choice([16, 32, 64]))
        sseed = int(rng.integers(1, 10 ** 6))
        body = synth.gt_body_pose().copy(); body[:2] += rng.uniform(-3, 3, 2)
        clean = synth.simulate_scan(scn, body, synth.HERCULES_BODY_T_LASER[int(rng.integers(2))], rings, seed=sseed)
        s = synth.roughen_scan(clean, seed=sseed, n_sectors=int(rng.integers(0, 5)), near_fraction=float(rng.choice([0.0, 0.005, 0.03])), short_rings=int(rng.integers(0, 4)))
        what = f"rough trial {trial}: {rings} rings, seed {sseed}, {len(s.points)} of {len(clean.points)} points"
        ref = O.extract(s.points, s.scan_start, s.scan_end)
        got = ctx.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
        for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw"):
            if not np.array_equal(got[k], ref[k]):
                raise SystemExit(f"ROUGH extract {k} {what}")
        if got["less_flat_ds"].shape != ref["less_flat_ds"].shape or not np.array_equal(got["less_flat_ds"].view(np.uint32), ref["less_flat_ds"].view(np.uint32)):
            raise SystemExit(f"ROUGH per-ring voxel centroids {what}")
        r2 = np.random.default_rng(sseed)
        pts = np.ascontiguousarray(s.points[r2.permutation(len(s.points))])
        pts[:, 3] = r2.uniform(0.0, 0.9, len(pts)).astype(np.float32)
        flag = bool(rng.integers(2))
        refs = O.segment_cloud(pts, O.seg_params(vertical_scans=rings, segment_flag=flag))
        gots = ctx.segment_cloud(pts, vertical_scans=rings, segment_flag=int(flag))
        for k in ("cloud", "outlier"):
            if gots[k].shape != refs[k].shape or not np.array_equal(gots[k].view(np.uint32), refs[k].view(np.uint32)):
                raise SystemExit(f"ROUGH segment {k} {what}")
        if not (np.array_equal(gots["scan_start"], refs["scan_start"]) and np.array_equal(gots["scan_end"], refs["scan_end"])):
            raise SystemExit(f"ROUGH segment scan info {what}")
        n_pts += len(pts)
    print(f"rough: {trials} roughened scans ({n_pts} points): extractCloud labels / lists / per-ring centroids and segmentCloud clouds / ScanInfo equal bit for bit  [{time.time() - t0:.0f} s]", flush=True)

if "voxel" in families:
    t0 = time.time(); n_pts = 0
    for trial in range(trials):
        n = int(rng.integers(1, 60000))
        ext = float(rng.choice([5.0, 20.0, 60.0]))
        leaf = float(rng.choice([0.1, 0.2, 0.4, 1.0]))
        xyz = rng.uniform(-ext, ext, (n, 3)).astype(np.float32)
        xyz[:, 2] *= 0.1
        snap = rng.random(n) < 0.2                                  # a fifth of the points on voxel faces
        xyz[snap] = (np.round(xyz[snap] / leaf) * leaf).astype(np.float32)
        if n > 10:
            rep = rng.integers(0, n, n // 10)                      # a tenth repeated verbatim
            xyz[rng.integers(0, n, len(rep))] = xyz[rep]
        pts4 = np.concatenate([xyz, rng.integers(0, 3, (n, 1)).astype(np.float32)], axis=1).astype(np.float32)
        what = f"voxel trial {trial}: n {n}, extent {ext}, leaf {leaf}"
        got, ref = ctx.voxel_filter(pts4, leaf), O.voxel_grid_mloam_plain(pts4, leaf, member_order=0)
        if got.shape != ref.shape or not np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
            raise SystemExit(f"VOXEL plain {what}")
        got, ref = ctx.voxel_grid(pts4, leaf), O.voxel_grid(pts4, leaf)
        if got.shape != ref.shape or not np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
            raise SystemExit(f"VOXEL pcl {what}")
        cov = np.abs(rng.normal(0.01, 0.01, (n, 6))).astype(np.float32)
        pts11 = np.concatenate([pts4, cov, (cov[:, 0] + cov[:, 3] + cov[:, 5])[:, None]], axis=1).astype(np.float32)
        thr = float(rng.choice([0.0, 0.05]))
        got, ref = ctx.voxel_filter(pts11, leaf, trace_threshold=thr), O.voxel_grid_cov(pts11, leaf, thr)
        if got.shape != ref.shape or not np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
            raise SystemExit(f"VOXEL covariance {what}, trace threshold {thr}: shapes {got.shape} / {ref.shape}")
        n_pts += n
    print(f"voxel: {trials} random clouds ({n_pts} points; face points, repeats): plain, pcl::VoxelGrid and covariance branches equal bit for bit  [{time.time() - t0:.0f} s]", flush=True)

if "select" in families:
    t0 = time.time(); n_sel = 0
    for trial in range(trials):
        sseed = int(rng.integers(1, 10 ** 6))
        case = conftest._make_case(synth, "50k", 16, int(rng.choice([1, 2])), seed=sseed)
        feats = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
        method = str(rng.choice(["rnd", "gd_fix", "gd_float", "fps"]))
        ratio = float(rng.choice([0.05, 0.2, 0.5, 0.9]))       # (0.9: more than match -- fps then fills up with feature 1, as the reference does)
        gseed = int(rng.integers(0, 1000))
        kind, ch = (mla.SURF, "s") if rng.integers(2) else (mla.CORNER, "c")
        cloud = case["surf_map"] if kind == mla.SURF else case["corner_map"]
        ctx.map_set(kind, cloud); ctx.features_set(kind, feats[kind])
        got = ctx.good_feature_matching(kind, case["p0"], gf_method=method, gf_ratio=ratio, seed=gseed)
        ref = O.good_feature_matching(O.Map(cloud), ch, feats[kind], case["p0"], O.mapper_params(gf_method=method, gf_ratio=ratio, seed=gseed))
        what = f"select trial {trial}: scene {sseed}, kind {ch}, {method}, ratio {ratio}, seed {gseed}"
        if not np.array_equal(got["sel"], ref["sel"]):
            raise SystemExit(f"SELECTION {what}: {len(got['sel'])} vs {len(ref['sel'])} picks")
        if float(np.abs(got["H"] - ref["H"]).max()) > 1e-9 * max(1.0, float(np.abs(ref["H"]).max())):
            raise SystemExit(f"SELECTION H {what}")
        n_sel += len(ref["sel"])
    print(f"select: {trials} random selections ({n_sel} picks): identical picks in identical order, information matrices within 1e-9  [{time.time() - t0:.0f} s]", flush=True)
if "odom_select" in families:
    t0 = time.time(); n_sel = 0
    for trial in range(trials):
        sseed = int(rng.integers(1, 10 ** 6))
        case = conftest._make_case(synth, "50k", 16, 1, seed=sseed)
        feats = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
        Tinv = np.linalg.inv(synth.pose_to_mat(case["gt"]))
        kind, ch = (mla.SURF, "s") if rng.integers(2) else (mla.CORNER, "c")
        mp = np.ascontiguousarray(synth.transform_points((case["surf_map"] if kind == mla.SURF else case["corner_map"])[:, :3], Tinv).astype(np.float32))
        m4 = np.zeros((len(mp), 4), np.float32); m4[:, :3] = mp
        f = np.ascontiguousarray(feats[kind][: int(rng.integers(30, len(feats[kind]) + 1))])
        pivot = case["gt"]
        pose_i = synth.perturbed_pose(case["gt"], seed=sseed + 1, dt=float(rng.choice([0.02, 0.1, 0.3])), drot_deg=float(rng.choice([0.2, 1.0])))
        q = rng.normal(size=4) * [0.01, 0.01, 0.02, 1.0]; q /= np.linalg.norm(q)
        ext = np.concatenate([rng.uniform(-0.05, 0.05, 3), q])
        T = np.linalg.inv(synth.pose_to_mat(pivot)) @ synth.pose_to_mat(pose_i) @ synth.pose_to_mat(ext)
        rel = np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
        ratio = float(rng.choice([1.0, 0.8, 0.8, 0.5, 0.3, 0.1]))
        gseed = int(rng.integers(1, 10 ** 5))
        ctx.map_set(kind, m4); ctx.features_set(kind, f)
        ctx.pure_odom_begin()
        got = ctx.pure_odom_add_matches_gf(kind, rel, pivot, pose_i, ext, 0, 0, gf_ratio=ratio, seed=gseed)
        ref = O.odom_good_feature_matching(O.Map(mp), ch, f, rel, pivot, pose_i, ext, ratio, gseed)
        if not np.array_equal(got, ref["sel"]):
            raise SystemExit(f"ODOM SELECT trial {trial}: scene {sseed}, kind {ch}, {len(f)} features, ratio {ratio}, seed {gseed}: {len(got)} vs {len(ref['sel'])} picks")
        n_sel += len(got)
    print(f"odom_select: {trials} random selections ({n_sel} picks): Estimator::goodFeatureMatching on the device path == the oracle's, pick for pick  [{time.time() - t0:.0f} s]", flush=True)
if "uct" in families:
    t0 = time.time(); n_pts = 0
    for trial in range(trials):
        n = int(rng.integers(10, 8000))
        kf = np.zeros((n, 11), np.float32)
        kf[:, :3] = rng.uniform(-50, 50, (n, 3)); kf[:, 2] *= 0.1
        kf[:, 3] = rng.integers(0, 2, n)
        q = rng.normal(size=4) * [0.05, 0.05, 0.5, 1.0]; q /= np.linalg.norm(q)
        pose_global = np.concatenate([rng.uniform(-5, 5, 3), q])
        A = rng.normal(size=(6, 6)) * float(rng.choice([1e-4, 1e-3, 1e-2]))
        cov_global = A @ A.T + np.eye(6) * 1e-6
        q2 = rng.normal(size=4) * [0.02, 0.02, 0.1, 1.0]; q2 /= np.linalg.norm(q2)
        ext = np.array([[0, 0, 0, 0, 0, 0, 1.0], np.concatenate([rng.uniform(-0.5, 0.5, 3), q2])])
        ext_cov = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3) * float(rng.choice([1.0, 10.0]))])
        cov_meas = np.diag([0.0025] * 3)
        with_ua = bool(rng.integers(3) > 0)
        ref_all = O.cloud_uct_associate_to_map(kf, pose_global, cov_global, ext, ext_cov, cov_meas, True, 1e30)
        tr = np.sort(ref_all[:, 10].astype(np.float64))
        # a gate that no trace sits on: the middle of the widest relative gap among the sorted traces' central half (a trace within rounding of the gate may fall either way)
        lo, hi = len(tr) // 4, max(len(tr) // 4 + 1, 3 * len(tr) // 4)
        gaps = (tr[lo + 1:hi + 1] - tr[lo:hi]) / np.maximum(tr[lo:hi], 1e-30) if hi > lo and hi < len(tr) else np.array([])
        thr = float(0.5 * (tr[lo + int(np.argmax(gaps))] + tr[lo + int(np.argmax(gaps)) + 1])) if len(gaps) and gaps.max() > 1e-4 else 1e30
        got = (O if os.environ.get("SOAK_DRY") else ctx).cloud_uct_associate_to_map(kf, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, thr)
        ref = O.cloud_uct_associate_to_map(kf, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, thr)
        what = f"uct trial {trial}: n {n}, with_ua {with_ua}, gate {thr:.3e}"
        if got.shape != ref.shape or not np.array_equal(got[:, :4].view(np.uint32), ref[:, :4].view(np.uint32)):
            raise SystemExit(f"UCT survivors / coordinates {what}: {got.shape} vs {ref.shape}")
        if len(ref) and float(np.abs(got[:, 4:] - ref[:, 4:]).max()) > 2e-5 * max(1e-12, float(np.abs(ref[:, 4:]).max())) + 1e-9:
            raise SystemExit(f"UCT covariances {what}")
        n_pts += n
    print(f"uct: {trials} random keyframe clouds ({n_pts} points): cloudUCTAssociateToMap on the device: survivors, order and f32 coordinates equal to the oracle's, covariances 2e-5  [{time.time() - t0:.0f} s]", flush=True)
if ctx is not None:
    ctx.close()
print(f"front-end parity soak: seed {seed}, {trials} trials per family, families {families}: all equal  [{time.time() - t_all:.0f} s]")
