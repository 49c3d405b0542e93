"""Oracle-vs-HIP parity at BASELINE sizes: config 2 (2 x 64 rings vs the ~500k-point map), through the C-ABI,
against the LIVE oracle on the same seeded inputs -- labels / picked flags / index lists bit-exact per 64-ring scan and for the joint
128-ring upload, k-NN indices + f32 distances, correspondence validity + coefficient bits, H / g 1e-9, per-iteration counts and the
5-GN pose, scan2map's LM bookkeeping. Every match path of the library is held to the same oracle results: the fused single-launch
kernel (strided and consecutive feature assignment) and the two-kernel path (MLH_FUSED=0)."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pose_err(a, b):
    dt = float(np.linalg.norm(np.asarray(a[:3]) - np.asarray(b[:3])))
    qa, qb = np.asarray(a[3:7]), np.asarray(b[3:7])
    dr = 2.0 * float(np.arccos(min(1.0, abs(float(np.dot(qa, qb))))))
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


MODES = {"fused_strided": dict(MLH_FUSED=1, MLH_FUSED_STRIDED=1), "fused_consecutive": dict(MLH_FUSED=1, MLH_FUSED_STRIDED=0),
         "two_kernel": dict(MLH_FUSED=0)}


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
