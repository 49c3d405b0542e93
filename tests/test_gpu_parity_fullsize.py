"""Oracle-vs-HIP parity at BASELINE sizes: config 2 (2 x 64 rings vs the ~500k-point map), through the C-ABI,
against the LIVE oracle on the same seeded inputs -- labels / picked flags / index lists bit-exact per 64-ring scan and for the joint
128-ring upload, k-NN indices + f32 distances, correspondence validity + coefficient bits, H / g 1e-9, per-iteration counts and the
5-GN pose, scan2map's LM bookkeeping. Every lane width of the correspondence search is held to the same oracle results: 8 / 16 lanes per query chosen
per kind from the map's density (the default), and either width forced for both kinds (MLH_KNN_LANES)."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pose_err(a, b):
    dt = float(np.linalg.norm(np.asarray(a[:3]) - np.asarray(b[:3])))
    qa, qb = np.asarray(a[3:7]), np.asarray(b[3:7])
    # rotation angle between the two: 2 |qa -+ qb| for small angles (acos of a dot product next to 1 cannot resolve anything below 3e-8)
    dr = 2.0 * float(min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)))
    return dt, dr


@pytest.fixture(scope="module")
def cfg2(synth, orc):
    """BASELINE config 2 workload exactly as bench.py builds it, features from the ORACLE's extraction."""
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
    ex = [orc.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    surf, corner = bench.fuse_features(synth, scans, ex)
    return dict(surf_map=surf_map, corner_map=corner_map, gt=gt, scans=scans, ex=ex, surf=surf, corner=corner,
                p0=synth.perturbed_pose(gt, seed=43), oms=orc.Map(surf_map), omc=orc.Map(corner_map))


def _ctx_with_env(mla, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return mla.Context(0)          # the switches are read at mlh_create
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


MODES = {"lanes_per_kind": dict(), "lanes_8": dict(MLH_KNN_LANES=8), "lanes_16": dict(MLH_KNN_LANES=16)}


@pytest.fixture(scope="module", params=list(MODES))
def staged(request, mla, cfg2):
    c = _ctx_with_env(mla, **MODES[request.param])
    c.map_set(mla.SURF, cfg2["surf_map"])
    c.map_set(mla.CORNER, cfg2["corner_map"])
    c.features_set(mla.SURF, cfg2["surf"])
    c.features_set(mla.CORNER, cfg2["corner"])
    yield c
    c.close()


def test_extract_labels_bit_exact_64_rings(mla, orc, cfg2):
    """feature labels bit-exact (north star) for both 64-ring scans and for the joint 128-ring upload."""
    c = mla.Context(0)
    scans, ex = cfg2["scans"], cfg2["ex"]
    for s, ref in zip(scans, ex):
        assert ref["n_ties"] == 0
        got = c.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
        assert np.array_equal(got["curvature"].view(np.uint32), ref["curvature"].view(np.uint32))
        for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw"):
            assert np.array_equal(got[k], ref[k]), k
        assert got["less_flat_ds"].shape == ref["less_flat_ds"].shape
        np.testing.assert_array_equal(got["less_flat_ds"].view(np.uint32), ref["less_flat_ds"].view(np.uint32))   # summed along std::sort's member order: the reference's bits
    offs = np.cumsum([0] + [len(s.points) for s in scans])
    pts = np.concatenate([s.points for s in scans])
    st = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
    en = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
    both = c.extract(pts, st, en)
    assert np.array_equal(both["label"], np.concatenate([e["label"] for e in ex]))
    assert np.array_equal(both["picked"], np.concatenate([e["picked"] for e in ex]))
    for k in ("sharp", "less_sharp", "flat", "less_flat_raw"):
        assert np.array_equal(both[k], np.concatenate([e[k] + offs[i] for i, e in enumerate(ex)])), k
    c.close()


def test_knn_exact_full_size(mla, cfg2):
    """k-NN indices + f32 distances at the features' map-frame positions (inside the acceptance radius: exact)."""
    c = mla.Context(0)
    Tm = None
    import importlib
    synth = importlib.import_module("m-loam_amd.synth")
    Tm = synth.pose_to_mat(cfg2["p0"])
    for kind, feats, mp, om in ((mla.SURF, cfg2["surf"], cfg2["surf_map"], cfg2["oms"]), (mla.CORNER, cfg2["corner"], cfg2["corner_map"], cfg2["omc"])):
        c.map_set(kind, mp)
        q = synth.transform_points(feats[:, :3], Tm)
        idx, d2 = c.knn(kind, q)
        ridx, rd2 = om.knn(q)
        inside = rd2[:, 4] < 1.0
        assert inside.sum() > 0.3 * len(q)
        assert np.array_equal(d2[inside].view(np.uint32), rd2[inside].view(np.uint32))
        assert np.array_equal(idx[inside], ridx[inside])
        assert np.all(d2[~inside][:, 4] >= 1.0)
    c.close()


def test_match_and_linearise_full_size(staged, mla, orc, cfg2):
    """validity + coefficient bits for every feature, residuals / Jacobians / H / g at 1e-9, zero decision flips."""
    flips = 0
    for kind, ch, feats, om in ((mla.SURF, "s", cfg2["surf"], cfg2["oms"]), (mla.CORNER, "c", cfg2["corner"], cfg2["omc"])):
        got = staged.match_linearize(kind, cfg2["p0"])
        v, co = om.match(ch, feats, cfg2["p0"])
        flips += int(np.sum(got["valid"] != v))
        assert np.array_equal(got["valid"], v), f"{int(np.sum(got['valid'] != v))} decision flips ({ch})"
        m = v.astype(bool)
        assert np.array_equal(got["coeffs"][m].astype(np.float32).view(np.uint32), co[m].astype(np.float32).view(np.uint32))
        ref = orc.linearize(ch, feats, None, cfg2["p0"], v, co)
        np.testing.assert_allclose(got["r"][m], ref["r"][m], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(got["J"][m], ref["J"][m], rtol=1e-9, atol=1e-11)
        sc = max(1.0, float(np.abs(ref["H"]).max()))
        assert float(np.abs(got["H"] - ref["H"]).max()) <= 1e-9 * sc
        assert float(np.abs(got["g"] - ref["g"]).max()) <= 1e-9 * max(1.0, float(np.abs(ref["g"]).max()))
        assert got["count"] == ref["count"] and abs(got["cost"] - ref["cost"]) <= 1e-9 * max(1.0, ref["cost"])
    assert flips == 0


def test_gn5_full_size(staged, orc, cfg2):
    """per-iteration (n_surf, n_corner), H, g and the pose after 5 Gauss-Newton iterations."""
    pose, stats = staged.gn_solve(cfg2["p0"], 5)
    ref = orc.gn_iterations(cfg2["oms"], cfg2["omc"], cfg2["surf"], cfg2["corner"], cfg2["p0"], orc.mapper_params(), 5)
    for s, r in zip(stats, ref["iters"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"])
        assert float(np.abs(s["H"] - r["H"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["H"]).max()))
        assert float(np.abs(s["g"] - r["g"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["g"]).max()))
        assert s["is_degenerate"] == r["is_degenerate"]
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)                      # north star: 1e-4 m / 1e-4 rad
    pose2, _ = staged.gn_solve(cfg2["p0"], 5, want_stats=False)   # the path bench.py times
    dt, dr = _pose_err(pose2, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_scan2map_full_size(staged, orc, cfg2):
    """the reference's per-frame call: LM iteration counts, successful steps, termination, costs, pose."""
    pose, stats = staged.scan2map(cfg2["p0"])
    ref = orc.scan2map(cfg2["oms"], cfg2["omc"], cfg2["surf"], cfg2["corner"], cfg2["p0"], orc.mapper_params())
    for s, r in zip(stats, ref["outer"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf_sel"], r["n_corner_sel"])
        assert (s["lm_iterations"], s["successful_steps"], s["termination"]) == (r["lm_iterations"], r["successful_steps"], r["termination"])
        assert s["is_degenerate"] == r["is_degenerate"]
        assert abs(s["cost"] - r["initial_cost"]) <= 1e-9 * max(1.0, r["initial_cost"])
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)
    pose2, _ = staged.scan2map(cfg2["p0"], want_stats=False)
    dt, dr = _pose_err(pose2, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


# ---------------------------------------------------------------- BASELINE configs 4 and 5 at full scan size (1M-point map)
@pytest.fixture(scope="module")
def cfg45(synth, orc):
    """4 x 64-ring scans (config 4's frame; config 5 uses the first two) against the 1M-point scene; corner map from 4-LiDAR keyframes."""
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "1M", n_lidars=4)
    ex = [orc.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    return dict(surf_map=surf_map, corner_map=corner_map, gt=gt, scans=scans, ex=ex, p0=synth.perturbed_pose(gt, seed=43),
                oms=orc.Map(surf_map), omc=orc.Map(corner_map))


def test_config5_full_size(mla, orc, synth, cfg45):
    """config 5: uncertainty-weighted residuals (with_ua) + greedy good-feature selection (gd_fix, ratio 0.2, fixed seed) on 2 x 64 rings vs
    the 1M map: identical selections, identical LM bookkeeping, same pose; the same for rnd and for the plain (wo_gf) weighted solve."""
    import bench
    surf, corner = bench.fuse_features(synth, cfg45["scans"][:2], cfg45["ex"][:2])
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
    meas = np.diag([0.0025] * 3)
    c = mla.Context(0)
    c.map_set_pair(cfg45["surf_map"], cfg45["corner_map"])
    fs = c.downsample_current_scan(mla.SURF, surf, 0.4, ext, covs, meas, True, 0.6)       # (m, 11) records with cov_vec: both sides use these
    fc = c.downsample_current_scan(mla.CORNER, corner, 0.2, ext, covs, meas, True, 0.6)
    assert len(fs) > 5000 and len(fc) > 5000
    for kind, ch, f, om in ((mla.SURF, "s", fs, cfg45["oms"]), (mla.CORNER, "c", fc, cfg45["omc"])):
        got = c.good_feature_matching(kind, cfg45["p0"], gf_method="gd_fix", gf_ratio=0.2, seed=7)
        ref = orc.good_feature_matching(om, ch, f, cfg45["p0"], orc.mapper_params(with_ua=True, gf_method="gd_fix", gf_ratio=0.2, seed=7))
        assert np.array_equal(got["sel"], ref["sel"]), ch
        np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-9, atol=1e-9)
    for method in ("gd_fix", "rnd", "wo_gf"):
        opts = mla.default_opts(flags=mla.FLAG_WITH_UA, gf_method=mla.GF_METHODS[method], gf_ratio=0.2, gf_seed=7)
        pose, st = c.scan2map(cfg45["p0"], opts)
        ref = orc.scan2map(cfg45["oms"], cfg45["omc"], fs, fc, cfg45["p0"], orc.mapper_params(with_ua=True, gf_method=method, gf_ratio=0.2, seed=7))
        for s, r in zip(st, ref["outer"]):
            assert (s["n_surf"], s["n_corner"]) == (r["n_surf_sel"], r["n_corner_sel"]), method
            assert (s["lm_iterations"], s["successful_steps"], s["termination"]) == (r["lm_iterations"], r["successful_steps"], r["termination"]), method
        dt, dr = _pose_err(pose, ref["pose"])
        assert dt < 1e-7 and dr < 1e-7, (method, dt, dr)
    c.close()


def test_config4_scan_full_size(mla, orc, synth, cfg45):
    """config 4's frame: 4 x 64 rings, one pose block per LiDAR (block 0 = body pose, N_NEIGH 5; blocks 1..3 = extrinsics, N_NEIGH 10), CHECK_FOV,
    frozen when degenerate -- all blocks and both kinds in the same launches; every block must reproduce the oracle's iteration on its cloud."""
    surf_b, corner_b, poses0 = [], [], []
    for i, (sc, ex) in enumerate(zip(cfg45["scans"], cfg45["ex"])):
        cpts = np.zeros((len(ex["less_sharp"]), 4), np.float32)
        cpts[:, :3] = sc.points[ex["less_sharp"]][:, :3]
        surf_b.append(np.ascontiguousarray(synth.voxel_mean(ex["less_flat_ds"].copy(), 0.4)))
        corner_b.append(np.ascontiguousarray(synth.voxel_mean(cpts, 0.2)))
        r = synth.HERCULES_BODY_T_LASER[i]
        T = synth.pose_to_mat(cfg45["gt"]) @ synth.pose_to_mat(np.concatenate([r[4:7], r[:4] / np.linalg.norm(r[:4])]))
        from scipy.spatial.transform import Rotation as Rot
        gt_i = np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
        poses0.append(synth.perturbed_pose(gt_i, seed=50 + i, dt=0.1, drot_deg=1.0))
    poses0 = np.array(poses0)
    k_neigh, thre, freeze = [5, 10, 10, 10], [100.0, 70.0, 70.0, 70.0], [0, 1, 1, 1]
    c = mla.Context(0)
    c.map_set_pair(cfg45["surf_map"], cfg45["corner_map"])
    c.features_set_blocks(mla.SURF, surf_b)
    c.features_set_blocks(mla.CORNER, corner_b)
    opts = mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=1.0)
    n_it = 3
    poses, stats = c.gn_solve_blocks(poses0, n_it, k_neigh, thre, freeze, opts)
    for b in range(4):
        prm = orc.mapper_params(huber_delta=1.0, map_eig_thre=thre[b], n_neigh=k_neigh[b], check_fov=True, freeze_when_degenerate=bool(freeze[b]))
        ref = orc.gn_iterations(cfg45["oms"], cfg45["omc"], surf_b[b], corner_b[b], poses0[b], prm, n_it)
        for it in range(n_it):
            s, r = stats[it][b], ref["iters"][it]
            assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"]), (b, it)
            assert float(np.abs(s["H"] - r["H"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["H"]).max()))
            assert s["is_degenerate"] == r["is_degenerate"]
        dt, dr = _pose_err(poses[b], ref["pose"])
        assert dt < 1e-7 and dr < 1e-7, (b, dt, dr)
    c.close()


def test_scan2map_against_the_references_own_lines_at_config2(mla, orc, cfg2):
    """The per-frame call itself at BASELINE config 2's size, with no restatement in between: `scan2MapOptimization` compiled from the reference's OWN lines
    (lidar_mapper_keyframe.cpp:423-639 over the Ceres-shaped shim, oracle/_ref) on the 500 k map against the HIP path -- synchronous (mlh_scan2map) and
    submitted / collected separately (mlh_scan2map_begin / _end): the same number of residual blocks and LM iterations in both outer iterations, the same pose."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref/libmloam_ref.so has not been built (needs /root/reference once)")
    ref = orc.ref_scan2map(cfg2["surf_map"], cfg2["corner_map"], cfg2["surf"], cfg2["corner"], cfg2["p0"])
    c = mla.Context(0)
    try:
        c.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
        c.features_set(mla.SURF, cfg2["surf"])
        c.features_set(mla.CORNER, cfg2["corner"])
        pose, st = c.scan2map(cfg2["p0"])
        c.scan2map_begin(cfg2["p0"])
        pose_split, status = c.scan2map_end()
    finally:
        c.close()
    assert len(ref["solves"]) == len(st) == 2
    for g, h in zip(ref["solves"], st):
        assert g["n_blocks"] == h["n_surf"] + h["n_corner"] and g["n_blocks"] > 10000
        assert (g["lm_iterations"], g["successful_steps"], g["termination"]) == (h["lm_iterations"], h["successful_steps"], h["termination"])
        assert abs(g["initial_cost"] - h["cost"]) <= 1e-9 * max(1.0, g["initial_cost"])
    assert float(np.abs(pose - ref["pose"]).max()) < 1e-9
    assert status == 0 and np.array_equal(pose_split, pose)


def test_extract_against_the_references_own_lines(mla, orc, cfg2):
    """the HIP extractCloud against oracle/_ref -- FeatureExtract::extractCloud compiled from the reference's own source lines
    (oracle/ref/build_ref.py; the prebuilt library travels to the GPU box) -- on both 64-ring scans: the same feature points in the same order."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref/libmloam_ref.so has not been built (needs /root/reference once)")
    c = mla.Context(0)
    for s in cfg2["scans"]:
        got = c.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
        want = orc.ref_extract(s.points, s.scan_start, s.scan_end)
        for k in ("sharp", "less_sharp", "flat"):
            assert np.array_equal(want[k].view(np.uint32), np.ascontiguousarray(s.points[got[k]]).view(np.uint32)), k
        assert want["less_flat_ds"].shape == got["less_flat_ds"].shape
        np.testing.assert_array_equal(got["less_flat_ds"].view(np.uint32), want["less_flat_ds"].view(np.uint32))
    # ... and the correspondence decisions + coefficients of the config-2 features at the bench's initial pose
    c.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
    c.features_set(mla.SURF, cfg2["surf"]); c.features_set(mla.CORNER, cfg2["corner"])
    for kind, ch, feats, cloud in ((mla.SURF, "s", cfg2["surf"], cfg2["surf_map"]), (mla.CORNER, "c", cfg2["corner"], cfg2["corner_map"])):
        out = c.match_linearize(kind, cfg2["p0"], dense=False)
        v_ref, c_ref = orc.ref_match(ch, cloud, feats, cfg2["p0"])
        assert np.array_equal(out["valid"], v_ref), f"{int(np.sum(out['valid'] != v_ref))} decision flips vs the reference's lines ({ch})"
        m = v_ref.astype(bool)
        assert np.array_equal(out["coeffs"][m].astype(np.float32).view(np.uint32), c_ref[m].astype(np.float32).view(np.uint32))
    c.close()


@pytest.mark.parametrize("mode", ["map", "features"])
def test_sharded_solver_full_size(mla, synth, orc, cfg2, mode):
    """config 2's frame split over 4 ranks (contexts on one GPU, host all-reduce): the 5-GN solve equals the oracle's unsharded one, iteration by iteration"""
    import importlib
    import test_gpu_parity as T
    shard = importlib.import_module("m-loam_amd.shard")
    pose, counts, owned = T._sharded_gn(mla, shard, 4, mode, cfg2["surf_map"], cfg2["corner_map"], cfg2["surf"], cfg2["corner"], cfg2["p0"], 5)
    ref = orc.gn_iterations(cfg2["oms"], cfg2["omc"], cfg2["surf"], cfg2["corner"], cfg2["p0"], orc.mapper_params(), 5)
    assert counts == [(r["n_surf"], r["n_corner"]) for r in ref["iters"]]
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_downsample_current_scan_against_the_references_own_lines(mla, orc, synth, cfg2):
    """mlh_downsample_current_scan (default member order = the reference's) against downsampleCurrentScan COMPILED FROM THE REFERENCE'S OWN LINES
    (lidar_mapper_keyframe.cpp:356-421; its filter = the literal restatement of VoxelGridCovarianceMLOAM<PointI>, std::sort order), on the bench
    frame's fused two-LiDAR clouds -- half of whose surf voxels hold points of both LiDARs: the same features survive the trace gate, in the same
    order, with the same coordinates and LiDAR ids bit for bit; covariances to f32 rounding of a differently ordered f64 product."""
    if orc.ref_lib() is None:
        pytest.skip("no prebuilt oracle/_ref/libmloam_ref.so")
    scans = cfg2["scans"]
    ex = [orc.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    fused = {"s": [], "c": []}
    for i, (sc, e) in enumerate(zip(scans, ex)):
        T = np.eye(4)
        T[:3, :3] = synth.quat_to_rot(synth.HERCULES_BODY_T_LASER[i][:4])
        T[:3, 3] = synth.HERCULES_BODY_T_LASER[i][4:7]
        for key, xyz in (("c", sc.points[e["less_sharp"]][:, :3]), ("s", e["less_flat_ds"][:, :3])):
            a = np.zeros((len(xyz), 4), np.float32)
            a[:, :3] = synth.transform_points(xyz, T)
            a[:, 3] = i
            fused[key].append(a)
    surf, corner = np.concatenate(fused["s"]), np.concatenate(fused["c"])
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e_ in ext:
        e_[3:] /= np.linalg.norm(e_[3:])
    covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
    meas = np.diag([0.0025] * 3)
    rs, rc = orc.ref_downsample_current_scan(surf, corner, 0.4, 0.2, ext, covs, meas, True, 0.6)
    c = mla.Context(0)
    try:
        for kind, cloud, leaf, want in ((mla.SURF, surf, 0.4, rs), (mla.CORNER, corner, 0.2, rc)):
            got = c.downsample_current_scan(kind, cloud, leaf, ext, covs, meas, True, 0.6)
            assert got.shape == want.shape and len(got) > 5000
            assert np.array_equal(got[:, :4].view(np.uint32), want[:, :4].view(np.uint32))
            np.testing.assert_allclose(got[:, 4:], want[:, 4:], rtol=5e-5, atol=1e-9)
        # and the opt-in device-only (point-index) order really is a different feature set on this frame
        c.set_voxel_member_order(False)
        dflt = c.downsample_current_scan(mla.SURF, surf, 0.4, ext, covs, meas, True, 0.6)
        assert dflt.shape != rs.shape or not np.array_equal(dflt[:, 3], rs[:, 3])
    finally:
        c.close()


# ---------------------------------------------------------------- BASELINE configs 3 and 4 at their STATED map sizes
def _ranks_as_contexts(mla, shard, world, mode, surf_map, corner_map, centre):
    """one context per rank on this GPU: its map wedge + halo with half-space ownership ("map"), or the whole map with round-robin feature
    ownership ("features") -- what every rank of an N-GPU job stages (m-loam_amd/shard.py, csrc/comm.hip)"""
    ctxs = []
    for r in range(world):
        c = mla.Context(0)
        if mode == "map":
            lo, hi = shard.wedge_planes(centre, world, r)
            c.shard_set(lo, hi)
            ms = np.ascontiguousarray(surf_map[shard.shard_points_mask(surf_map, centre, world, r)])
            mc = np.ascontiguousarray(corner_map[shard.shard_points_mask(corner_map, centre, world, r)])
            far = np.full((1, 3), 1.0e6, np.float32)
            c.map_set_pair(ms if len(ms) else far, mc if len(mc) else far)
        else:
            c.shard_set_features(world, r)
            c.map_set_pair(surf_map, corner_map)
        ctxs.append(c)
    return ctxs


def _sharded_block_gn(mla, ctxs, surf, corner, p0, n_iters, k_neigh=5, flags=0, huber_delta=0.1, eig_thre=100.0, freeze=False):
    """the N-rank Gauss-Newton loop of one pose block with the all-reduce done by the host (sum of the ranks' packed J^T J / J^T r / counts),
    then the one solve every rank would repeat: evalDegenracy -> H d = -g -> Plus, or no update at all for a frozen degenerate block"""
    for c in ctxs:
        c.features_set(mla.SURF, surf)
        c.features_set(mla.CORNER, corner)
    pose = np.array(p0, np.float64)
    counts, owned = [], np.zeros((len(ctxs), 2), np.int64)
    for _ in range(n_iters):
        H, g, ns, nc = np.zeros((6, 6)), np.zeros(6), 0, 0
        for r, c in enumerate(ctxs):
            a = c.match_linearize(mla.SURF, pose, flags=flags, huber_delta=huber_delta, dense=False, k_neigh=k_neigh)
            b = c.match_linearize(mla.CORNER, pose, flags=flags, huber_delta=huber_delta, dense=False, k_neigh=k_neigh)
            H += a["H"] + b["H"]; g += a["g"] + b["g"]; ns += a["count"]; nc += b["count"]
            owned[r] = (a["count"], b["count"])
        deg = mla.eval_degeneracy(H, eig_thre)
        if not (freeze and deg["is_degenerate"]):
            pose = mla.pose_plus(pose, np.linalg.solve(H, -g), deg["V_update"] if deg["is_degenerate"] else None)
        counts.append((ns, nc))
    return pose, counts, owned


@pytest.fixture(scope="module")
def cfg3(synth, orc):
    """BASELINE config 3: the 2 x 64-ring frame against the 2 M-point map (the preset bench.py --map-preset 2M builds)"""
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "2M")
    assert len(surf_map) + len(corner_map) > 2_000_000
    ex = [orc.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    surf, corner = bench.fuse_features(synth, scans, ex)
    p0 = synth.perturbed_pose(gt, seed=43)
    oms, omc = orc.Map(surf_map), orc.Map(corner_map)
    ref = orc.gn_iterations(oms, omc, surf, corner, p0, orc.mapper_params(), 5)
    return dict(surf_map=surf_map, corner_map=corner_map, surf=surf, corner=corner, p0=p0, ref=ref)


def test_config3_unsharded_2M(mla, cfg3):
    """the 2 M map on ONE context: grid extents, cell_start sizes and the sticky grid box are what changes against the 500 k map"""
    c = mla.Context(0)
    try:
        c.map_set_pair(cfg3["surf_map"], cfg3["corner_map"])
        c.features_set(mla.SURF, cfg3["surf"]); c.features_set(mla.CORNER, cfg3["corner"])
        pose, stats = c.gn_solve(cfg3["p0"], 5)
        ref = cfg3["ref"]
        for s, r in zip(stats, ref["iters"]):
            assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"])
            assert float(np.abs(s["H"] - r["H"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["H"]).max()))
            assert float(np.abs(s["g"] - r["g"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["g"]).max()))
        dt, dr = _pose_err(pose, ref["pose"])
        assert dt < 1e-7 and dr < 1e-7, (dt, dr)
        # a second frame through the re-staging path (same box: the sticky geometry is reused) gives the same answer
        c.map_set_pair(cfg3["surf_map"], cfg3["corner_map"])
        pose2, _ = c.gn_solve(cfg3["p0"], 5, want_stats=False)
        assert np.array_equal(pose2, pose)
    finally:
        c.close()


@pytest.mark.parametrize("mode", ["map", "features"])
def test_config3_sharded_over_4_ranks_2M(mla, orc, cfg3, mode):
    """config 3 as BASELINE states it: the 2 M map voxel-sharded over 4 ranks (contexts on this GPU, the all-reduce summed by the host): per-iteration
    matched counts and the 5-GN pose equal the UNSHARDED oracle's, in both partitions"""
    import importlib
    shard = importlib.import_module("m-loam_amd.shard")
    ctxs = _ranks_as_contexts(mla, shard, 4, mode, cfg3["surf_map"], cfg3["corner_map"], cfg3["p0"][:2])
    try:
        pose, counts, owned = _sharded_block_gn(mla, ctxs, cfg3["surf"], cfg3["corner"], cfg3["p0"], 5)
    finally:
        for c in ctxs:
            c.close()
    ref = cfg3["ref"]
    assert counts == [(r["n_surf"], r["n_corner"]) for r in ref["iters"]]
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)
    assert (owned.sum(axis=1) > 0).all()                       # every rank really took part


@pytest.fixture(scope="module")
def cfg4(synth, orc):
    """BASELINE config 4: 4 x 64 rings against the 4 M-point map; per-LiDAR feature clouds (pose blocks) from the oracle's extraction"""
    import bench
    from scipy.spatial.transform import Rotation as Rot
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "4M", n_lidars=4)
    assert len(surf_map) + len(corner_map) > 4_000_000
    surf_b, corner_b, poses0, exts = [], [], [], []
    for i, s in enumerate(scans):
        ex = orc.extract(s.points, s.scan_start, s.scan_end)
        cpts = np.zeros((len(ex["less_sharp"]), 4), np.float32)
        cpts[:, :3] = s.points[ex["less_sharp"]][:, :3]
        surf_b.append(np.ascontiguousarray(synth.voxel_mean(ex["less_flat_ds"].copy(), 0.4)))
        corner_b.append(np.ascontiguousarray(synth.voxel_mean(cpts, 0.2)))
        r = synth.HERCULES_BODY_T_LASER[i]
        e = np.concatenate([r[4:7], r[:4] / np.linalg.norm(r[:4])])
        exts.append(e)
        T = synth.pose_to_mat(gt) @ synth.pose_to_mat(e)
        gt_i = np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
        poses0.append(synth.perturbed_pose(gt_i, seed=50 + i, dt=0.1, drot_deg=1.0))
    return dict(surf_map=surf_map, corner_map=corner_map, gt=gt, surf_b=surf_b, corner_b=corner_b, poses0=np.array(poses0), exts=np.array(exts),
                oms=orc.Map(surf_map), omc=orc.Map(corner_map))


CFG4_K, CFG4_THRE, CFG4_FREEZE = [5, 10, 10, 10], [100.0, 70.0, 70.0, 70.0], [0, 1, 1, 1]


def _cfg4_refs(orc, cfg4, n_it):
    refs = []
    for b in range(4):
        prm = orc.mapper_params(huber_delta=1.0, map_eig_thre=CFG4_THRE[b], n_neigh=CFG4_K[b], check_fov=True, freeze_when_degenerate=bool(CFG4_FREEZE[b]))
        refs.append(orc.gn_iterations(cfg4["oms"], cfg4["omc"], cfg4["surf_b"][b], cfg4["corner_b"][b], cfg4["poses0"][b], prm, n_it))
    return refs


def test_config4_pose_blocks_4M(mla, orc, cfg4):
    """config 4's frame on the 4 M map, ONE context: four pose blocks (N_NEIGH 5 / 10 / 10 / 10, CHECK_FOV, freeze-on-degenerate) in the same launches;
    every block reproduces the oracle's iterations on its own cloud"""
    n_it = 3
    c = mla.Context(0)
    try:
        c.map_set_pair(cfg4["surf_map"], cfg4["corner_map"])
        c.features_set_blocks(mla.SURF, cfg4["surf_b"])
        c.features_set_blocks(mla.CORNER, cfg4["corner_b"])
        opts = mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=1.0)
        poses, stats = c.gn_solve_blocks(cfg4["poses0"], n_it, CFG4_K, CFG4_THRE, CFG4_FREEZE, opts)
    finally:
        c.close()
    for b, ref in enumerate(_cfg4_refs(orc, cfg4, n_it)):
        for it in range(n_it):
            s, r = stats[it][b], ref["iters"][it]
            assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"]), (b, it)
            assert float(np.abs(s["H"] - r["H"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["H"]).max()))
            assert s["is_degenerate"] == r["is_degenerate"]
        assert ref["iters"][0]["n_surf"] + ref["iters"][0]["n_corner"] > 2000, b
        dt, dr = _pose_err(poses[b], ref["pose"])
        assert dt < 1e-7 and dr < 1e-7, (b, dt, dr)


@pytest.mark.parametrize("mode", ["map", "features"])
def test_config4_pose_blocks_over_8_ranks_4M(mla, orc, cfg4, mode):
    """config 4 as BASELINE states it -- the 4 M map over 8 ranks: every pose block's sharded iteration (8 contexts on this GPU, host all-reduce,
    the redundant per-rank solve with the block's threshold / freeze policy) equals the unsharded oracle's"""
    import importlib
    shard = importlib.import_module("m-loam_amd.shard")
    n_it = 3
    refs = _cfg4_refs(orc, cfg4, n_it)
    ctxs = _ranks_as_contexts(mla, shard, 8, mode, cfg4["surf_map"], cfg4["corner_map"], cfg4["gt"][:2])
    try:
        for b in range(4):
            pose, counts, owned = _sharded_block_gn(mla, ctxs, cfg4["surf_b"][b], cfg4["corner_b"][b], cfg4["poses0"][b], n_it, k_neigh=CFG4_K[b],
                                                    flags=mla.FLAG_CHECK_FOV, huber_delta=1.0, eig_thre=CFG4_THRE[b], freeze=bool(CFG4_FREEZE[b]))
            assert counts == [(r["n_surf"], r["n_corner"]) for r in refs[b]["iters"]], b
            dt, dr = _pose_err(pose, refs[b]["pose"])
            assert dt < 1e-7 and dr < 1e-7, (b, dt, dr)
    finally:
        for c in ctxs:
            c.close()


def _cfg4_window(synth, cfg4):
    from scipy.spatial.transform import Rotation as Rot
    frame0 = synth.perturbed_pose(cfg4["gt"], seed=70, dt=0.05, drot_deg=0.5)
    exts0 = np.array([e if i == 0 else synth.perturbed_pose(e, seed=80 + i, dt=0.03, drot_deg=0.3) for i, e in enumerate(cfg4["exts"])])
    to_pose = lambda T: np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
    rels = [to_pose(synth.pose_to_mat(frame0) @ synth.pose_to_mat(exts0[i])) for i in range(4)]        # T_pivot^-1 T_frame T_ext with T_pivot = I
    return frame0, exts0, rels


def test_config4_coupled_window_4M(mla, orc, synth, cfg4):
    """config 4 proper on the 4 M map: Estimator::optimizeMap's coupled problem [pivot | 1 frame | 4 extrinsics] = 36 local parameters. The factor table is
    built ON THE DEVICE from eight match passes against the resident map (mlh_pure_odom_begin / _add_matches, N_NEIGH 5, CHECK_FOV); its J^T J / J^T r /
    cost equal the oracle's factor-by-factor accumulation over the ORACLE's own matches (1e-9), the coupled Gauss-Newton step with the pivot and the
    reference extrinsic held constant (estimator.cpp:636, 642) is the same, and so are five coupled iterations."""
    frame0, exts0, rels = _cfg4_window(synth, cfg4)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    tab = [[], [], [], [], []]
    for i in range(4):
        for ty, ch, om, f in ((0, "s", cfg4["oms"], cfg4["surf_b"][i]), (1, "c", cfg4["omc"], cfg4["corner_b"][i])):
            v, co = om.match(ch, f, rels[i], n_neigh=5, check_fov=True)
            m = v.astype(bool)
            tab[0].append(np.full(m.sum(), ty, np.int32)); tab[1].append(f[m, :3].astype(np.float64)); tab[2].append(co[m])
            tab[3].append(np.zeros(m.sum(), np.int32)); tab[4].append(np.full(m.sum(), i, np.int32))
    tab = [np.concatenate(a) for a in tab]
    assert len(tab[0]) > 20000
    D = 36
    free = np.r_[6:12, 18:36]                                     # the frame + extrinsics 1..3
    c = mla.Context(0)
    try:
        c.map_set_pair(cfg4["surf_map"], cfg4["corner_map"])
        c.pure_odom_begin()
        for i in range(4):
            for kind, f in ((mla.SURF, cfg4["surf_b"][i]), (mla.CORNER, cfg4["corner_b"][i])):
                c.features_set(kind, f)
                c.pure_odom_add_matches(kind, rels[i], 0, i, k_neigh=5, flags=mla.FLAG_CHECK_FOV)
        got = c.pure_odom_normal_eq(ident, frame0[None, :], exts0, huber_delta=1.0)
        ref = orc.pure_odom_normal_eq(tab[0], tab[1], tab[2], None, tab[3], tab[4], ident, frame0[None, :], exts0, 1.0)
        assert got["H"].shape == (D, D) and got["count"] == ref["count"] == len(tab[0])
        sc = float(np.abs(ref["H"]).max())
        assert float(np.abs(got["H"] - ref["H"]).max()) <= 1e-9 * sc
        assert float(np.abs(got["g"] - ref["g"]).max()) <= 1e-9 * float(np.abs(ref["g"]).max())
        assert abs(got["cost"] - ref["cost"]) <= 1e-9 * ref["cost"]
        step_g = np.linalg.solve(got["H"][np.ix_(free, free)], -got["g"][free])
        step_r = np.linalg.solve(ref["H"][np.ix_(free, free)], -ref["g"][free])
        np.testing.assert_allclose(step_g, step_r, rtol=1e-6, atol=1e-9)
        assert np.linalg.norm(step_r[:3]) > 1e-3
        # five coupled Gauss-Newton iterations on the fixed table (what ceres::Solve iterates on), device J^T J vs the oracle's
        def window_gn(neq):
            fr, ex = frame0.copy()[None, :], exts0.copy()
            for _ in range(5):
                ne = neq(fr, ex)
                step = np.zeros(D); step[free] = np.linalg.solve(ne["H"][np.ix_(free, free)], -ne["g"][free])
                fr[0] = mla.pose_plus(fr[0], step[6:12])
                for k in range(1, 4):
                    ex[k] = mla.pose_plus(ex[k], step[12 + 6 * k:18 + 6 * k])
            return fr, ex
        fr_g, ex_g = window_gn(lambda fr, ex: c.pure_odom_normal_eq(ident, fr, ex, huber_delta=1.0))
        fr_c, ex_c = window_gn(lambda fr, ex: orc.pure_odom_normal_eq(tab[0], tab[1], tab[2], None, tab[3], tab[4], ident, fr, ex, 1.0))
        # ... and the same five iterations without the host in the loop (mlh_pure_odom_gn_solve on the device-built table)
        sol = c.pure_odom_gn_solve(ident, frame0[None, :], exts0, n_iters=5, huber_delta=1.0)
        assert sol["status"] == 0 and sol["count"] == len(tab[0])
        dt, dr = _pose_err(sol["frames"][0], fr_c[0])
        assert dt < 1e-9 and dr < 1e-9, (dt, dr)
        for k in range(1, 4):
            dt, dr = _pose_err(sol["exts"][k], ex_c[k])
            assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)
        dt, dr = _pose_err(fr_g[0], fr_c[0])
        assert dt < 1e-9 and dr < 1e-9, (dt, dr)
        for k in range(1, 4):
            dt, dr = _pose_err(ex_g[k], ex_c[k])
            assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)
    finally:
        c.close()
    # ... and the same system from 8 ranks (replicated map, round-robin ownership; every rank builds its own part of the table on the device):
    # the summed records are the unsharded system
    import importlib
    shard = importlib.import_module("m-loam_amd.shard")
    ctxs = _ranks_as_contexts(mla, shard, 8, "features", cfg4["surf_map"], cfg4["corner_map"], cfg4["gt"][:2])
    try:
        H, g, cost, cnt = np.zeros((D, D)), np.zeros(D), 0.0, 0
        for cr in ctxs:
            cr.pure_odom_begin()
            for i in range(4):
                for kind, f in ((mla.SURF, cfg4["surf_b"][i]), (mla.CORNER, cfg4["corner_b"][i])):
                    cr.features_set(kind, f)
                    cr.pure_odom_add_matches(kind, rels[i], 0, i, k_neigh=5, flags=mla.FLAG_CHECK_FOV)
            ne = cr.pure_odom_normal_eq(ident, frame0[None, :], exts0, huber_delta=1.0)
            H += ne["H"]; g += ne["g"]; cost += ne["cost"]; cnt += ne["count"]
        assert cnt == ref["count"]
        assert float(np.abs(H - ref["H"]).max()) <= 1e-9 * sc
        assert float(np.abs(g - ref["g"]).max()) <= 1e-9 * float(np.abs(ref["g"]).max())
        assert abs(cost - ref["cost"]) <= 1e-9 * ref["cost"]
    finally:
        for cr in ctxs:
            cr.close()
