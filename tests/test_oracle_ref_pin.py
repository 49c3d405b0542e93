"""The oracle pinned against the reference's OWN SOURCE LINES (oracle/_ref, built by oracle/ref/build_ref.py from the files under
/root/reference): FeatureExtract::extractCloud and match*PointFromMap, the six factor classes, PoseLocalParameterization::Plus, ImageSegmenter,
evalPointUncertainty / compoundPoseWithCov (associate_uct.hpp), cloudUCTAssociateToMap and evalDegenracy (lidar_mapper_keyframe.cpp), the
tracker's match*FromScan / TransformToEnd / scan factors, and ActiveFeatureSelection::{evalFullHessian, goodFeatureMatching} with
common::logDet (lidar_mapper.h:130-573, math.hpp:172-202).
CPU only; skipped where neither the reference tree nor a prebuilt oracle/_ref/libmloam_ref.so exists."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(orc):
    if orc.ref_lib() is None:
        pytest.skip("no /root/reference and no prebuilt oracle/_ref/libmloam_ref.so")
    return orc


def _same_cloud(ref_cloud, pts, idx):
    assert len(ref_cloud) == len(idx)
    assert np.array_equal(ref_cloud.view(np.uint32), np.ascontiguousarray(pts[idx]).view(np.uint32))


def _check_scan(ref, pts, ss, se, tie_rule=0):
    got = ref.extract(pts, ss, se, tie_rule=tie_rule)
    want = ref.ref_extract(pts, ss, se)
    for k in ("sharp", "less_sharp", "flat"):
        _same_cloud(want[k], pts, got[k])          # same points, same order: the labels (2 / 1 / -1) and the pick order are the reference's
    assert want["less_flat_ds"].shape == got["less_flat_ds"].shape
    assert np.array_equal(want["less_flat_ds"].view(np.uint32), got["less_flat_ds"].view(np.uint32))
    return got


def test_extract_cloud_is_the_references(ref, synth, case16):
    sc = case16["scans"][0]
    got = _check_scan(ref, sc.points, sc.scan_start, sc.scan_end)
    assert len(got["sharp"]) > 100 and len(got["flat"]) > 200
    # labels follow from the lists: sharp -> 2, less sharp \ sharp -> 1, flat -> -1 (feature_extract.cpp:171-231)
    lab = np.zeros(len(sc.points), np.int32)
    lab[got["less_sharp"]] = 1; lab[got["sharp"]] = 2; lab[got["flat"]] = -1
    assert np.array_equal(lab, got["label"])


def test_extract_cloud_64_rings_and_ragged(ref, synth):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scn = synth.make_scene(seed=5, **synth.SCENE_PRESETS["50k"])
        sc = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[1], 64, seed=21)
    _check_scan(ref, sc.points, sc.scan_start, sc.scan_end)
    # ragged ring table: some rings too short to be processed (end - start < 6, feature_extract.cpp:155), some empty
    ss, se = sc.scan_start.copy(), sc.scan_end.copy()
    se[3] = ss[3] + 4
    se[10] = ss[10] - 1
    _check_scan(ref, sc.points, ss, se)


def test_extract_cloud_with_exact_ties_matches_std_sort(ref, case16):
    """quantised coordinates -> thousands of exactly equal curvatures. The reference's comparator + std::sort on the same data in the same
    order is deterministic, and the oracle (tie_rule 0: the same comparator, the same libstdc++ introsort) reproduces it pick for pick."""
    sc = case16["scans"][0]
    pts = sc.points.copy()
    pts[:, :3] = np.round(pts[:, :3] * 32.0) / 32.0
    got = _check_scan(ref, pts, sc.scan_start, sc.scan_end, tie_rule=0)
    assert got["n_ties"] > 500


def test_map_factors_are_the_references(ref):
    rng = np.random.default_rng(17)
    for i in range(200):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pose = np.concatenate([rng.uniform(-30, 30, 3), q])
        p = rng.uniform(-40, 40, 3)
        sd = rng.uniform(0.01, 0.6, 3)
        cov = np.diag(sd ** 2)
        cov[0, 1] = cov[1, 0] = 0.1 * sd[0] * sd[1]
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        plane = np.concatenate([n, [rng.uniform(-5, 5)]])
        c = rng.uniform(-40, 40, 3); v = rng.normal(size=3); v /= np.linalg.norm(v)
        line = np.concatenate([c + 0.1 * v, c - 0.1 * v])
        for kind, coeff in (("s", plane), ("c", line)):
            r_ref, J_ref = ref.ref_map_factor(kind, p, coeff, cov, pose)
            r_orc, J_orc = ref.factor_eval(kind, p, coeff, float(np.trace(cov)), pose)
            assert abs(r_ref - r_orc) <= 1e-12 * max(1.0, abs(r_orc)), (kind, i)
            np.testing.assert_allclose(J_ref, J_orc, rtol=1e-11, atol=1e-11)
            assert J_ref[6] == 0.0
            r2, _ = ref.ref_map_factor(kind, p, coeff, cov, pose, want_jacobian=False)     # jacobians == NULL
            assert r2 == r_ref
    # the weight rule of the constructors: sqrt(1 / trace) >= 3 -> 1, else / 3 (lidar_map_factor.hpp:35, 41)
    for tr, w in ((0.0075, 1.0), (1.0 / 9.0, 1.0), (0.5, np.sqrt(2.0) / 3.0)):
        r_ref, _ = ref.ref_map_factor("s", [1.0, 2.0, 3.0], [0.0, 0.0, 1.0, 0.5], np.eye(3) * tr / 3.0, [0, 0, 0, 0, 0, 0, 1.0])
        assert abs(r_ref - w * 3.5) < 1e-12


def _rand_pose(rng, scale):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    return np.concatenate([rng.uniform(-scale, scale, 3), q])


def test_odometry_and_calibration_factors_are_the_references(ref):
    """LidarPureOdom{PlaneNorm,Edge}Factor (lidar_pure_odom_factor.hpp:27-102, 198-282: residual + three 1x7 rows over pivot / frame / extrinsic)
    and LidarOnlineCalib{PlaneNorm,Edge}Factor (lidar_online_calib_factor.hpp:24-62, 125-165) against the oracle's restatements -- including the
    two Jacobian columns of the reference that are not exact derivatives (they must be reproduced, not corrected)."""
    rng = np.random.default_rng(23)
    for i in range(150):
        pivot, pose_i, ext = _rand_pose(rng, 20.0), _rand_pose(rng, 20.0), _rand_pose(rng, 1.0)
        p = rng.uniform(-40, 40, 3)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        plane = np.concatenate([n, [rng.uniform(-5, 5)]])
        c = rng.uniform(-40, 40, 3); v = rng.normal(size=3); v /= np.linalg.norm(v)
        line = np.concatenate([c + 0.1 * v, c - 0.1 * v])
        s = rng.uniform(0.3, 1.0)
        for kind, coeff in (("s", plane), ("c", line)):
            r_ref, J_ref = ref.ref_pure_odom(kind, p, coeff, s, pivot, pose_i, ext)
            r_orc, J_orc = ref.pure_odom_eval(kind, p, coeff, pivot, pose_i, ext, s)
            assert abs(r_ref - r_orc) <= 1e-11 * max(1.0, abs(r_orc)), (kind, i)
            np.testing.assert_allclose(J_ref, J_orc, rtol=1e-10, atol=1e-10)
            assert not J_ref[:, 6].any()
            # the calibration factors: the map factor's form on the extrinsic alone, weight handed in (1.0 at estimator.cpp:757, 813)
            rc, Jc = ref.ref_online_calib(kind, p, coeff, 1.0, ext)
            ro, Jo = ref.factor_eval(kind, p, coeff, 0.0075, ext)        # trace 0.0075 -> weight 1 in the map factor
            assert abs(rc - ro) <= 1e-12 * max(1.0, abs(ro))
            np.testing.assert_allclose(Jc, Jo, rtol=1e-11, atol=1e-11)


def test_pose_local_parameterization_plus_is_the_references(ref):
    """PoseLocalParameterization::Plus (pose_local_parameterization.cpp:26-45): dx' = V_update dx, t += dx'_t, q = (q * [dx'_theta / 2, 1]).normalized()"""
    rng = np.random.default_rng(29)
    for i in range(100):
        x = _rand_pose(rng, 50.0)
        d = rng.normal(0, 0.05, 6)
        a = ref.ref_pose_plus(x, d)
        b = ref.pose_plus(x, d)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-15)
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        V = Q[:, 2:] @ Q[:, 2:].T                    # a projector, as evalDegenracy leaves in V_update_
        np.testing.assert_allclose(ref.ref_pose_plus(x, d, V), ref.pose_plus(x, d, V), rtol=0, atol=1e-15)


@pytest.mark.parametrize("n_neigh,check_fov", [(5, False), (5, True), (10, True)])
def test_match_point_from_map_is_the_references(ref, case16, feats16, n_neigh, check_fov):
    """FeatureExtract::matchSurfPointFromMap / matchCornerPointFromMap (feature_extract.hpp:645-883) compiled from the reference's lines (k-NN,
    eigen solver and QR behind them are the oracle's restatements): the 5th-neighbour gate, the line / plane tests, the FOV check (whose
    sqrt(3.0f) resolves to the float overload -- the check stays in f32), d = 1 / |n| before the normalisation, the endpoint arithmetic."""
    for kind, feats, cloud in (("s", feats16[0], case16["surf_map"]), ("c", feats16[1], case16["corner_map"])):
        v_ref, c_ref = ref.ref_match(kind, cloud, feats, case16["p0"], n_neigh, check_fov)
        v_orc, c_orc = ref.Map(cloud).match(kind, feats, case16["p0"], n_neigh=n_neigh, check_fov=check_fov)
        assert np.array_equal(v_ref, v_orc), (kind, int(np.sum(v_ref != v_orc)))
        assert v_ref.sum() > 50
        m = v_ref.astype(bool)
        assert np.array_equal(c_ref[m], c_orc[m])          # f32 values widened to double: identical bits
    if check_fov:
        # a pose pitched by 50 degrees: part of the scan now lies outside the +-60 degree cone around the sensor's z axis, and both sides
        # reject the same features (the features are matched where the tilted pose puts them, so fewer find neighbours at all)
        import importlib
        synth = importlib.import_module("m-loam_amd.synth")
        tilt = case16["p0"].copy()
        tilt[3:] = synth.quat_mul(case16["p0"][3:], synth.rotvec_to_quat(np.deg2rad([0.0, 50.0, 0.0])))
        rng = np.random.default_rng(5)
        cloud = (case16["surf_map"][rng.choice(len(case16["surf_map"]), 20000, replace=False)]).astype(np.float32)
        Tm = synth.pose_to_mat(tilt)
        local = synth.transform_points(cloud[:4000], np.linalg.inv(Tm))          # features that land exactly on map points under `tilt`
        f4 = np.zeros((len(local), 4), np.float32); f4[:, :3] = local
        v_all, _ = ref.ref_match("s", cloud, f4, tilt, n_neigh, False)
        v_fov, c_fov = ref.ref_match("s", cloud, f4, tilt, n_neigh, True)
        v_orc, c_orc = ref.Map(cloud).match("s", f4, tilt, n_neigh=n_neigh, check_fov=True)
        assert np.array_equal(v_fov, v_orc) and np.array_equal(c_fov[v_fov.astype(bool)], c_orc[v_fov.astype(bool)])
        assert 0 < v_fov.sum() < v_all.sum()               # the check really removes something here


@pytest.mark.parametrize("n_neigh,check_fov", [(5, True), (10, True), (10, False)])
def test_whole_cloud_match_functions_are_the_references(ref, case16, feats16, n_neigh, check_fov):
    """FeatureExtract::matchSurfFromMap / matchCornerFromMap -- the whole-cloud forms (feature_extract.hpp:378-643) buildCalibMap calls with N_NEIGH 10 and CHECK_FOV
    for the non-reference LiDARs (estimator.cpp:1130-1150) -- compiled from the reference's own lines: the same features survive, with the same coefficients, as in the
    per-point forms the oracle restates (and the device's mlh_pure_odom_add_matches is held to)."""
    for kind, feats, cloud in (("s", feats16[0], case16["surf_map"]), ("c", feats16[1], case16["corner_map"])):
        v_ref, c_ref = ref.ref_match_cloud(kind, cloud, feats, case16["p0"], n_neigh, check_fov)
        v_orc, c_orc = ref.Map(cloud).match(kind, feats, case16["p0"], n_neigh=n_neigh, check_fov=check_fov)
        assert np.array_equal(v_ref, v_orc), (kind, int(np.sum(v_ref != v_orc)))
        assert v_ref.sum() > 50
        m = v_ref.astype(bool)
        assert np.array_equal(c_ref[m], c_orc[m])


def _raw_cloud(synth, n_rings, seed, clutter):
    """an UNORDERED cloud as a driver delivers it: a simulated scan, shuffled, part of the points pulled off their surfaces along the ray"""
    scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    s = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[0], n_rings, seed=seed)
    rng = np.random.default_rng(seed)
    pts = s.points.copy()
    pts[:, 3] = 0
    m = rng.random(len(pts)) < clutter
    pts[m, :3] *= rng.uniform(0.5, 1.3, (int(m.sum()), 1)).astype(np.float32)
    return pts[rng.permutation(len(pts))]


@pytest.mark.parametrize("vs,rings", [(16, 16), (32, 64), (64, 64)])
def test_image_segmenter_is_the_references(ref, synth, vs, rings, capfd):
    """ImageSegmenter::segmentCloud (image_segmenter.hpp:88-393) compiled from the reference's lines against the restatement: the ring-major cloud,
    ScanInfo and the outlier cloud bit for bit, with and without clutter (hundreds to thousands of rejected clusters exercise the stale-position
    erasures and the next-queue-entry rule), both thresholds the configs use, several ROI ranges, segmentation on and off. The reference's
    undefined spots are pinned the same way on both sides (oracle/image_segmenter.hpp U1-U3; the shim context bounds-checks what the oracle skips)."""
    n_checked = 0
    for seed in range(3):
        for clutter in (0.0, 0.1, 0.4):
            pts = _raw_cloud(synth, rings, seed, clutter)
            for flag in (True, False):
                prm = ref.seg_params(vertical_scans=vs, segment_theta=1.047 if seed % 2 else 0.53, roi_range=[1.0, 0.5, 6.0][seed], segment_flag=flag)
                a, b = ref.segment_cloud(pts, prm), ref.ref_segment_cloud(pts, prm)
                assert a["cloud"].shape == b["cloud"].shape and np.array_equal(a["cloud"].view(np.uint32), b["cloud"].view(np.uint32))
                assert np.array_equal(a["scan_start"], b["scan_start"]) and np.array_equal(a["scan_end"], b["scan_end"])
                assert a["outlier"].shape == b["outlier"].shape and np.array_equal(a["outlier"].view(np.uint32), b["outlier"].view(np.uint32))
                n_checked += 1
                if flag and clutter > 0:
                    assert (a["label_mat"] == 999999).sum() > 500
    capfd.readouterr()        # the reference's setParameter prints a line per call
    assert n_checked == 18


def test_image_segmenter_known_answers(orc):
    """hand-checkable cases of the restatement: the projection's row / column rule, first-point-wins, the +5 / -6 insets of ScanInfo, the
    intensity += row convention, a 3-pixel cluster that is rejected and a vertical 5-pixel pole that the line rule keeps."""
    prm = orc.seg_params(vertical_scans=16, horizon_scans=1800, segment_flag=True)
    def ray(row, col, r):
        el = np.deg2rad(-15.0 + 2.0 * row)
        ha = np.deg2rad(90.0 - (col - 900) * 0.2)            # column_id = -round((ha - 90) / 0.2) + 900
        return [r * np.cos(el) * np.sin(ha), r * np.cos(el) * np.cos(ha), r * np.sin(el), 0.25]
    pts = []
    for col in range(200, 260):                              # a wall: rows 8..12, constant range -> one big cluster
        for row in range(8, 13):
            pts.append(ray(row, col, 12.0))
    pts += [ray(9, 600, 7.0), ray(9, 601, 7.0), ray(9, 602, 7.0)]        # 3 pixels on one ring: fewer than segment_valid_point_num -> outliers
    pts += [ray(r, 1000, 9.0) for r in range(8, 13)]                      # a pole: 5 pixels over 5 rings -> kept by the line rule (>= 5 points, >= 3 rings)
    pts.append(ray(10, 220, 3.0))                                          # second hit on an occupied pixel: the first point keeps it
    pts.append(ray(4, 50, 0.5))                                            # inside roi_range = 1: dropped
    pts = np.array(pts, np.float32)
    out = orc.segment_cloud(pts, prm)
    lab = out["label_mat"]
    assert out["pixel_of_point"][-1] == -1 and out["pixel_of_point"][-2] == -1
    assert out["pixel_of_point"][0] == 8 * 1800 + 200
    assert (lab[9, 600:603] == 999999).all() and (lab[8:13, 1000] > 1).all() and (lab[8:13, 1000] < 999999).all()
    assert len(np.unique(lab[8:13, 200:260])) == 1
    # the three rejected pixels sit at positions 60, 61, 62 of ring 9's list (after the wall's 60 points, before the pole's): the first
    # erasure (60) shifts the rest down, the stale position 61 then removes what was recorded at 62, and the stale 62 points past the end --
    # the reference's erase is undefined there (U2), here it erases nothing. So ONE rejected point (column 601) survives in the output:
    assert len(out["cloud"]) == 60 * 5 + 5 + 1
    survivors = out["cloud"][np.isclose(np.linalg.norm(out["cloud"][:, :3], axis=1), 7.0, atol=1e-3)]
    assert len(survivors) == 1 and np.allclose(survivors[0, :3], ray(9, 601, 7.0)[:3], atol=1e-5)
    assert np.array_equal(out["cloud"][:, 3], np.floor(out["cloud"][:, 3]) + 0.25)           # intensity + row id
    rows = np.floor(out["cloud"][:, 3]).astype(int)
    assert np.all(np.diff(rows) >= 0)                                     # ring-major
    for r in range(16):
        n_r = int((rows == r).sum()); first = int((rows < r).sum())
        assert out["scan_start"][r] == first + 5 and out["scan_end"][r] == first + n_r - 6
    # outlier cloud: outlier pixels whose column is a multiple of 5 (600), plus the first point of the output cloud (hpp:391)
    assert len(out["outlier"]) == 2 and np.array_equal(out["outlier"][1], out["cloud"][0])


def test_eval_point_uncertainty_is_the_references(ref, synth):
    """evalPointUncertainty + pointToFS (estimator/src/lidarMapper/associate_uct.hpp:150-215), both overloads, compiled from the reference's own
    lines: the oracle's restatement (which the HIP path's point_uncertainty_kernel is held against) gives the same 3x3 covariance -- the
    formula G diag(cov_pose, COV_MEASUREMENT) G^T with G = [(T p)^odot | T D] is pinned; the products' rounding is the shim's, so 1e-12."""
    rng = np.random.default_rng(8)
    for trial in range(40):
        pose = np.concatenate([rng.uniform(-3, 3, 3), (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))])
        A = rng.normal(size=(6, 6)) * 0.02
        cov_pose = A @ A.T + np.diag([0.0025] * 3 + [0.0003] * 3)
        cov_meas = np.diag(rng.uniform(0.001, 0.01, 3))
        p = rng.uniform(-40, 40, 3).astype(np.float32)
        T4 = synth.pose_to_mat(pose)
        r1, r2 = ref.ref_eval_point_uncertainty(p, T4, cov_pose, cov_meas)
        assert np.array_equal(r1, r2)                                      # the two overloads are the same arithmetic
        got = ref.eval_point_uncertainty(p[None, :], pose, cov_pose, cov_meas)[0]
        sc = float(np.abs(r1).max())
        assert float(np.abs(got - r1).max()) <= 1e-12 * sc, (trial, got, r1)
        assert np.allclose(r1, r1.T, rtol=0, atol=1e-12 * sc)


def test_compound_pose_with_cov_is_the_references(ref):
    """compoundPoseWithCov (associate_uct.hpp:9-86: adjointMatrix, covop1, covop2, Barfoot's fourth-order compound, method 2) compiled from the
    reference's own lines vs the oracle's restatement (held by the HIP path's mlh_compound_pose_with_cov / cloudUCTAssociateToMap tests)."""
    rng = np.random.default_rng(9)
    for trial in range(40):
        poses, covs = [], []
        for _ in range(2):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            poses.append(np.concatenate([rng.uniform(-5, 5, 3), q]))
            A = rng.normal(size=(6, 6)) * rng.uniform(0.005, 0.05)
            covs.append(A @ A.T)
        rp, rc = ref.ref_compound_pose_with_cov(poses[0], covs[0], poses[1], covs[1])
        gp, gc = ref.compound_pose_with_cov(poses[0], covs[0], poses[1], covs[1])
        assert np.allclose(gp, rp, rtol=0, atol=1e-13)
        sc = float(np.abs(rc).max())
        assert float(np.abs(gc - rc).max()) <= 1e-12 * sc, (trial, float(np.abs(gc - rc).max()), sc)


def _with_cov(f, rng):
    a = np.zeros((len(f), 11), np.float32)
    a[:, :4] = f[:, :4]
    sd = rng.uniform(0.03, 0.2, (len(f), 3)).astype(np.float32)
    a[:, 4] = sd[:, 0] ** 2; a[:, 7] = sd[:, 1] ** 2; a[:, 9] = sd[:, 2] ** 2
    a[:, 10] = a[:, 4] + a[:, 7] + a[:, 9]
    return a


def test_eval_full_hessian_and_logdet_are_the_references(ref, case16, feats16):
    """ActiveFeatureSelection::evalFullHessian (lidar_mapper.h:176-227) with evaluateFeatJacobianMatching (:130-174) and common::logDet
    (math.hpp:172-202) compiled from the reference's own lines: same matched count, same information matrix (1e-9: f64 sums of the same rows),
    same gf_deg_factor."""
    rng = np.random.default_rng(4)
    Hr, nr = None, 0
    Ho, no = None, 0
    for ch, cloud, f in (("s", case16["surf_map"], feats16[0]), ("c", case16["corner_map"], feats16[1])):
        f11 = _with_cov(f, rng)
        Hr, nr, ld = ref.ref_eval_full_hessian(cloud, ch, f11, case16["p0"], Hr, nr)
        Ho, no = ref.eval_full_hessian(ref.Map(cloud), ch, f11, case16["p0"], Ho, no)
    assert nr == no and nr > 1000
    assert float(np.abs(Hr - Ho).max()) <= 1e-9 * float(np.abs(Hr).max())
    assert abs(ld - ref.logdet(Ho)) <= 1e-9 * abs(ld)


@pytest.mark.parametrize("method", ["wo_gf", "rnd", "fps", "gd_fix", "gd_float"])
def test_good_feature_matching_is_the_references(ref, case16, feats16, method):
    """ActiveFeatureSelection::goodFeatureMatching (lidar_mapper.h:229-573) compiled from the reference's own lines -- the draw / retry / erase
    bookkeeping, the per-pick heap, the subset size, the early terminations -- against the oracle's restatement (which the HIP path's
    selection is held to): the SAME features in the SAME order and the same sub_mat_H, for both kinds and several engine seeds. (The
    reference seeds its mt19937 from std::random_device; both sides are given the same seed here.)"""
    rng = np.random.default_rng(6)
    for ch, cloud, f in (("s", case16["surf_map"], feats16[0][:1500]), ("c", case16["corner_map"], feats16[1][:1200])):
        f11 = _with_cov(f, rng)
        om = ref.Map(cloud)
        for seed in (1, 7, 12345):
            # wo_gf: the reference sizes sel_feature_idx by num_all * gf_ratio and then stores EVERY match (lidar_mapper.h:247-248, 290): anything
            # below 1.0 writes past the vector -- the mapper always runs wo_gf at 1.0 (gf_ratio policy, lidar_mapper_keyframe.cpp:476)
            for ratio in ((0.2, 0.8) if method.startswith("gd") else ((1.0,) if method == "wo_gf" else (0.2,))):
                r = ref.ref_good_feature_matching(cloud, ch, f11, case16["p0"], method, ratio, seed)
                o = ref.good_feature_matching(om, ch, f11, case16["p0"], ref.mapper_params(with_ua=True, gf_method=method, gf_ratio=ratio, seed=seed))
                Ho = np.eye(6) * 1e-6 + 0 * o["H"]
                assert len(r["sel"]) > 20
                assert np.array_equal(r["sel"], o["sel"]), (ch, seed, ratio, int(np.sum(r["sel"][:min(len(r["sel"]), len(o["sel"]))] != o["sel"][:min(len(r["sel"]), len(o["sel"]))])))
                assert float(np.abs(r["H"] - o["H"]).max()) <= 1e-9 * float(np.abs(r["H"]).max())


@pytest.mark.parametrize("variant", ["lattice", "duplicates"])
def test_fps_equal_distances_are_decided_as_the_reference_decides_them(ref, case16, feats16, variant):
    """The farthest-point loop of goodFeatureMatching ('fps', lidar_mapper.h:352-408) takes the FIRST index among equal distances (a strict `>` over
    ascending indices). The device kernel that runs this loop (select.hip: fps_order_kernel) is held to the oracle on inputs that produce such ties by the
    thousand (tests/test_gpu_parity.py::test_fps_selection_on_the_device_ties_and_edge_sizes); here the oracle is held to the reference's own lines on the
    same kind of input: features snapped to a 0.25 m lattice, and features repeated verbatim."""
    rng = np.random.default_rng(16)
    f = feats16[0][:1500].copy()
    if variant == "lattice":
        f[:, :3] = np.round(f[:, :3] * 4.0) / 4.0
    else:
        f = np.ascontiguousarray(np.concatenate([f[:500], f[:500], f[:200][::-1], f[500:1000]]))
    f11 = _with_cov(f, rng)
    # how many exact ties the loop meets on its first step alone (distances from one point to all others, in the loop's f32 arithmetic)
    d0 = np.sqrt(((f[:, :3] - f[0, :3]).astype(np.float32) ** 2).sum(1, dtype=np.float32))
    assert len(d0) - len(np.unique(d0)) > 100
    om = ref.Map(case16["surf_map"])
    for seed in (1, 7, 12345):
        r = ref.ref_good_feature_matching(case16["surf_map"], "s", f11, case16["p0"], "fps", 0.2, seed)
        o = ref.good_feature_matching(om, "s", f11, case16["p0"], ref.mapper_params(with_ua=True, gf_method="fps", gf_ratio=0.2, seed=seed))
        assert len(r["sel"]) > 20
        assert np.array_equal(r["sel"], o["sel"]), (variant, seed)
        assert float(np.abs(r["H"] - o["H"]).max()) <= 1e-9 * float(np.abs(r["H"]).max())


def test_fps_asked_for_more_features_than_match(ref, case16, feats16):
    """'fps' with gf_ratio * N above the number of features that match (found by scripts/soak_ref_pin.py): the reference's loop has no "everything visited" exit
    (lidar_mapper.h:391-399, the test is commented out), so after the last point its scan leaves best_j = 1 and feature 1 is matched -- and, matched, appended with
    its J^T J -- again and again until the count is reached; unmatched, the loop spins into its 20 ms cut-off (the shim's clock advances per reading, so it ends) and
    returns what it has. The oracle restates both outcomes; the HIP path is held to it (tests/test_gpu_parity.py::test_fps_asked_for_more_features_than_match)."""
    rng = np.random.default_rng(8)
    p0 = case16["p0"]
    seen = set()
    for ch, cloud, f in (("c", case16["corner_map"], feats16[1][:600]), ("s", case16["surf_map"], feats16[0][:600])):
        f = f.copy()
        f[::2, :3] += 500.0                      # every second feature (feature 0, 2, ...) far from the map: at most half can match; feature 1 stays where it is
        for first_matches in (True, False):
            g = f.copy()
            if not first_matches:
                g[1, :3] += 500.0               # ... and now feature 1 cannot match either
            f11 = _with_cov(g, rng)
            r = ref.ref_good_feature_matching(cloud, ch, f11, p0, "fps", 0.8, 5)
            o = ref.good_feature_matching(ref.Map(cloud), ch, f11, p0, ref.mapper_params(with_ua=True, gf_method="fps", gf_ratio=0.8, seed=5))
            assert np.array_equal(r["sel"], o["sel"]) and float(np.abs(r["H"] - o["H"]).max()) <= 1e-9 * float(np.abs(r["H"]).max())
            n_use, uniq = int(len(g) * 0.8), len(set(r["sel"].tolist()))
            assert uniq <= len(g) // 2 + 1 < n_use
            if len(r["sel"]) == n_use:           # feature 1 matched: the list was filled up with it
                assert (r["sel"] == 1).sum() == n_use - uniq + 1 and np.all(r["sel"][uniq:] == 1)
                seen.add("filled")
            else:                                # feature 1 did not match: the loop ran into its cut-off
                assert len(r["sel"]) == uniq and 1 not in r["sel"]
                seen.add("cut off")
    assert seen == {"filled", "cut off"}


@pytest.mark.parametrize("method", ["gd_fix", "rnd"])
def test_selection_with_repeated_features_is_the_references(ref, case16, feats16, method):
    """Features repeated verbatim give the stochastic-greedy loop subsets whose members score EXACTLY alike: which of them std::priority_queue leaves on top is
    the reference's container's business, and the oracle (and through it the HIP path, which replays such subsets through the same heap) has to agree with the
    reference's own lines there too. `rnd` rides along: repeated features do not change its draws, only what they hit."""
    rng = np.random.default_rng(26)
    f = feats16[0][:1200]
    f = np.ascontiguousarray(np.concatenate([f[:400], f[:400], f[:400], f[400:800], f[400:800]]))
    f11 = _with_cov(f, rng)
    f11[400:800] = f11[:400]; f11[800:1200] = f11[:400]; f11[1600:2000] = f11[1200:1600]     # the same covariance columns too: identical rows
    om = ref.Map(case16["surf_map"])
    for seed in (3, 99):
        for ratio in (0.2, 0.5):
            r = ref.ref_good_feature_matching(case16["surf_map"], "s", f11, case16["p0"], method, ratio, seed)
            o = ref.good_feature_matching(om, "s", f11, case16["p0"], ref.mapper_params(with_ua=True, gf_method=method, gf_ratio=ratio, seed=seed))
            assert len(r["sel"]) > 50
            assert np.array_equal(r["sel"], o["sel"]), (method, seed, ratio)
            assert float(np.abs(r["H"] - o["H"]).max()) <= 1e-9 * float(np.abs(r["H"]).max())


def test_track_matching_is_the_references(ref, track_case):
    """LidarTracker's correspondence search -- FeatureExtract::matchCornerFromScan / matchSurfFromScan (feature_extract.hpp:131-376: nearest
    previous-frame point, then the two scan-line walks with NEARBY_SCAN and DISTANCE_SQ_THRESHOLD) over TransformToStart (utility.h:54-77) --
    compiled from the reference's own lines: the oracle's restatement (which the HIP tracker is held to) accepts the same features with the
    same coefficient bits, at the identity and at a moved pose estimate."""
    tc = track_case
    poses = [np.array([0, 0, 0, 0, 0, 0, 1.0]), tc["motion"] if "motion" in tc else np.array([0.3, -0.1, 0.0, 0, 0, 0.0131, 0.99991])]
    for pose in poses:
        for kind, prev, cur in (("c", tc["corner_last"], tc["corner_sharp"]), ("s", tc["surf_last"], tc["surf_flat"])):
            rv, rc = ref.ref_track_match(kind, prev, cur, pose)
            ov, oc = ref.track_match(kind, prev, cur, pose)
            assert rv.sum() > 50
            assert np.array_equal(rv, ov), (kind, int(np.sum(rv != ov)))
            m = rv.astype(bool)
            assert np.array_equal(rc[m].astype(np.float32).view(np.uint32), oc[m].astype(np.float32).view(np.uint32)), kind
            np.testing.assert_allclose(rc[m], oc[m], rtol=1e-12, atol=1e-12)


def test_scan_factors_and_transform_to_end_are_the_references(ref):
    """LidarScanPlaneNormFactor / LidarScanEdgeFactorVector (lidar_scan_factor.hpp:25-62, 236-279) and TransformToEnd (utility.h:79-100) from the
    reference's own lines vs the oracle (slerp from the identity is the shim's restatement of Eigen 3.3 on one side, the oracle's on the other)."""
    rng = np.random.default_rng(10)
    for trial in range(30):
        q = rng.normal(size=4) * np.array([0.05, 0.05, 0.05, 0]) + np.array([0, 0, 0, 1.0]); q /= np.linalg.norm(q)
        pose = np.concatenate([rng.uniform(-0.5, 0.5, 3), q])
        p = rng.uniform(-20, 20, 3)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        for kind, coeff in (("S", np.concatenate([n, [rng.uniform(-3, 3)]])), ("E", np.concatenate([p + rng.normal(0, 0.5, 3), p + rng.normal(0, 0.5, 3)]))):
            for s_ in (1.0, 0.37):
                rr, rJ = ref.ref_scan_factor_eval(kind, p, coeff, pose, s_)
                orr, oJ = ref.scan_factor_eval(kind, p, coeff, pose, s_)
                np.testing.assert_allclose(orr, rr, rtol=1e-12, atol=1e-12)
                np.testing.assert_allclose(oJ, rJ, rtol=1e-11, atol=1e-11)
    pts = np.zeros((500, 4), np.float32)
    pts[:, :3] = rng.uniform(-30, 30, (500, 3))
    pts[:, 3] = rng.integers(0, 16, 500) + rng.uniform(0, 0.0999, 500).astype(np.float32)
    pose = np.array([0.4, -0.1, 0.02, 0.0, 0.0, np.sin(0.01), np.cos(0.01)])
    for dist in (True, False):
        a = ref.ref_transform_to_end(pts, pose, dist, 0.1)
        b = ref.transform_to_end(pts, pose, dist, 0.1)
        assert np.array_equal(a[:, 3], b[:, 3])
        np.testing.assert_allclose(a[:, :3], b[:, :3], rtol=0, atol=4e-6)              # f64 math stored to f32: at most an ulp at 30 m
        assert np.mean(a.view(np.uint32) == b.view(np.uint32)) > 0.99


def test_cloud_uct_associate_to_map_is_the_references(ref, feats16):
    """cloudUCTAssociateToMap (lidar_mapper_keyframe.cpp:1116-1158) over the pose.cov_ overload of compoundPoseWithCov (associate_uct.hpp:88-147),
    Pose::inverse / update (pose.cpp:99-108) and updateCov, from the reference's own lines: the same points survive the trace gate, in the same
    order, with the same map-frame coordinates (f32 bits) and covariances (1e-6 relative in f32)."""
    rng = np.random.default_rng(12)
    f = feats16[0][:4000]
    kf = np.zeros((len(f), 11), np.float32)
    kf[:, :3] = f[:, :3]
    kf[:, 3] = rng.integers(0, 2, len(f))
    q = np.array([0.01, -0.02, 0.3, 1.0]); q /= np.linalg.norm(q)
    pose_global = np.concatenate([[1.5, -0.7, 0.2], q])
    A = rng.normal(size=(6, 6)) * 0.001
    cov_global = A @ A.T + np.eye(6) * 1e-5
    ext = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0.1, -0.5, 0.02, 0, 0, 0.0998334166468, 0.995004165278]])
    ext_cov = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
    cov_meas = np.diag([0.0025] * 3)
    for with_ua, thr in ((True, 0.6), (True, 0.035), (False, 0.6)):
        r = ref.ref_cloud_uct_associate_to_map(kf, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, thr)
        o = ref.cloud_uct_associate_to_map(kf, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, thr)
        assert r.shape == o.shape and 0 < len(r) <= len(kf)
        assert np.array_equal(r[:, :4].view(np.uint32), o[:, :4].view(np.uint32))
        np.testing.assert_allclose(o[:, 4:], r[:, 4:], rtol=2e-6, atol=1e-12)
    assert len(ref.ref_cloud_uct_associate_to_map(kf, pose_global, cov_global, ext, ext_cov, cov_meas, True, 0.035)) < len(kf)   # the gate really cuts


def test_eval_degeneracy_is_the_references(ref):
    """evalDegenracy (lidar_mapper_keyframe.cpp:1172-1204) from the reference's own lines: which eigen-directions fall under MAP_EIG_THRE (the loop
    leaves at the first one that does not), the flag, and the projector V_update = V_f^-T V_p^T the solver's Plus then applies. The 6x6
    eigen-decomposition underneath is the oracle's on both sides; the projector does not depend on the eigenvectors' signs."""
    rng = np.random.default_rng(13)
    n_deg = 0
    for trial in range(60):
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        k = trial % 4                                                   # 0..3 eigenvalues under the threshold
        ev = np.concatenate([rng.uniform(1.0, 80.0, k), rng.uniform(150.0, 5000.0, 6 - k)])
        H = (Q * ev) @ Q.T
        H = 0.5 * (H + H.T)
        r = ref.ref_eval_degeneracy(H, 100.0)
        o = ref.eval_degeneracy(H, 100.0)
        assert r["is_degenerate"] == o["is_degenerate"] == (k > 0)
        np.testing.assert_allclose(np.sort(r["eigval"]), np.sort(ev), rtol=1e-9)
        if k > 0:
            n_deg += 1
            np.testing.assert_allclose(o["V_update"], r["V_update"], rtol=0, atol=1e-9)
            # the projector keeps exactly the non-degenerate subspace
            keep = Q[:, k:] @ Q[:, k:].T
            np.testing.assert_allclose(r["V_update"], keep, rtol=0, atol=1e-9)
    assert n_deg >= 40


def _mixed_lidar_clouds(synth, feats16, rng):
    clouds = []
    for f in (feats16[0], feats16[1]):
        xyz = np.concatenate([f[:, :3], f[:, :3] + rng.normal(0, 0.08, f[:, :3].shape).astype(np.float32)])
        a = np.zeros((len(xyz), 4), np.float32)
        a[:, :3] = xyz
        a[:, 3] = rng.integers(0, 2, len(xyz))                 # both LiDARs' points inside the same voxels
        clouds.append(a)
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.diag([0.0004] * 3 + [0.0001] * 3), np.diag([0.0025] * 3 + [0.00030461] * 3) * 30])
    return clouds, ext, covs, np.diag([0.0025] * 3)


def test_downsample_current_scan_is_the_references(ref, synth, feats16):
    """downsampleCurrentScan (lidar_mapper_keyframe.cpp:356-421) from the reference's own lines: thinned point -> int(intensity) picks the LiDAR ->
    pointAssociateToMap with the INVERSE extrinsic -> evalPointUncertainty through pose_ext[idx] (its cov_) -> trace gate -> PointIWithCov(point,
    cov.cast<float>()). Against the composition the HIP path's mlh_downsample_current_scan is tested with (the oracle's plain voxel filter in the
    reference's std::sort member order, then its evalPointUncertainty and the gate), on clouds whose voxels mix both LiDARs."""
    rng = np.random.default_rng(21)
    (surf, corner), ext, covs, meas = _mixed_lidar_clouds(synth, feats16, rng)
    for with_ua, thr in ((True, 0.05), (True, 0.6), (False, 0.6)):
        rs, rc = ref.ref_downsample_current_scan(surf, corner, 0.4, 0.2, ext, covs, meas, with_ua, thr)
        for got, cloud, leaf in ((rs, surf, 0.4), (rc, corner, 0.2)):
            ds = ref.voxel_grid_mloam_plain(cloud, leaf, member_order=0)
            rows = []
            for p in ds:
                n = int(p[3])
                cov = np.zeros((3, 3))
                if with_ua:
                    R = synth.quat_to_rot(ext[n][3:])
                    sel = ((p[:3].astype(np.float64) - ext[n][:3]) @ R).astype(np.float32)
                    cov = ref.eval_point_uncertainty(sel[None, :], ext[n], covs[n], meas)[0]
                    if np.trace(cov) > thr:
                        continue
                c32 = cov.astype(np.float32)
                rows.append(np.concatenate([p[:4], [c32[0, 0], c32[0, 1], c32[0, 2], c32[1, 1], c32[1, 2], c32[2, 2]], [c32[0, 0] + c32[1, 1] + c32[2, 2]]]).astype(np.float32))
            want = np.array(rows, np.float32).reshape(-1, 11)
            assert got.shape == want.shape and len(got) > 100
            assert np.array_equal(got[:, :4].view(np.uint32), want[:, :4].view(np.uint32))
            np.testing.assert_allclose(got[:, 4:], want[:, 4:], rtol=2e-6, atol=1e-12)
        if with_ua and thr == 0.05:
            assert len(rs) < len(ref.voxel_grid_mloam_plain(surf, 0.4, member_order=0))       # the gate cuts


@pytest.mark.parametrize("frame_cnt,estimate_extrinsic", [(20, True), (23, True), (20, False)])
def test_estimator_eval_degeneracy_is_the_references(ref, frame_cnt, estimate_extrinsic):
    """Estimator::evalDegenracy (estimator.cpp:1598-1680) compiled from the reference's own lines against the restatement
    (oracle/mapper.cpp: window_eval_degeneracy), on a Jacobian with the window problem's shape -- every row three 1 x 6 blocks: pivot, one frame,
    one extrinsic. The pose blocks go through the mapper's rule with per-block thresholds (one frame is made degenerate in two directions);
    the extrinsic blocks through all three calibration branches (lambda >= LAMBDA_THRE_CALIB; between the running threshold and it; below the
    running threshold), the every-N-frames gate and the ESTIMATE_EXTRINSIC switch."""
    rng = np.random.default_rng(17)
    W, L_, n_rows = 3, 3, 900                       # OPT_WINDOW_SIZE = 3 -> 4 pose blocks; 3 LiDARs
    n_pose = W + 1
    D = 6 * (n_pose + L_)
    rows, cols, vals = [0], [], []
    ext_scale = [3.0, 0.9, 0.05]                     # lambda_min / N of the three extrinsic blocks: above 70, between, below the running threshold
    for r in range(n_rows):
        f, e = 1 + r % W, r % L_
        for b, sc in ((0, 1.0), (f, 1.0), (n_pose + e, ext_scale[e])):
            j = rng.normal(0, 1, 6) * sc
            if b == 2:
                j[[1, 4]] *= 1e-3                    # frame 2: two weak directions
            cols += [6 * b + k for k in range(6)]
            vals += list(j)
        rows.append(len(cols))
    J = np.zeros((n_rows, D))
    for r in range(n_rows):
        J[r, cols[rows[r]:rows[r + 1]]] = vals[rows[r]:rows[r + 1]]
    thr = np.array([100.0] * n_pose + [0.0, 5.0, 5.0])
    got = ref.ref_estimator_eval_degeneracy(rows, cols, vals, D, W, L_, thr, estimate_extrinsic, frame_cnt, 10, 70.0)
    want = ref.window_eval_degeneracy(J.T @ J, n_pose, thr, estimate_extrinsic, frame_cnt, 10, 70.0)
    np.testing.assert_array_equal(got["is_degenerate"], want["is_degenerate"])
    np.testing.assert_allclose(got["eig_thre"], want["eig_thre"], rtol=1e-9)
    np.testing.assert_allclose(got["d_factor_calib"], want["d_factor_calib"], rtol=1e-9)
    np.testing.assert_allclose(got["V_update"], want["V_update"], rtol=0, atol=1e-8)
    assert got["is_degenerate"][2] and not got["is_degenerate"][:2].any() and not got["is_degenerate"][3]
    assert np.linalg.matrix_rank(got["V_update"][2], tol=1e-6) == 4                     # the two weak directions are projected out
    if estimate_extrinsic and frame_cnt % 10 == 0:
        assert list(got["is_degenerate"][n_pose:]) == [False, False, True]
        assert got["eig_thre"][n_pose] == 70.0 and got["d_factor_calib"][0] > 70.0      # branch 1
        assert 5.0 < got["eig_thre"][n_pose + 1] < 70.0 and got["d_factor_calib"][1] == 0.0   # branch 2: the running threshold rises
        assert not got["V_update"][n_pose + 2].any()                                    # branch 3: frozen
    elif estimate_extrinsic:
        assert got["is_degenerate"][n_pose:].all() and not got["V_update"][n_pose:].any()      # not a calibration frame: every extrinsic frozen
    else:
        assert not got["is_degenerate"][n_pose:].any()


def test_eval_hessian_is_the_references(ref, case16, feats16):
    """evalHessian (lidar_mapper_keyframe.cpp:1160-1169) compiled from the reference's own lines -- CRSMatrix2EigenMatrix, J^T J, the leading 6 x 6 -- on
    the Jacobian rows of a real frame (loss switched off, so the rows are what problem.Evaluate would hand it): the restatement's normal-equation
    accumulation, which the HIP reduction is held against, gives the same H."""
    Js, Hsum = [], np.zeros((6, 6))
    for kind, m, f in (("s", ref.Map(case16["surf_map"]), feats16[0]), ("c", ref.Map(case16["corner_map"]), feats16[1])):
        valid, coeffs = m.match(kind, f, case16["p0"])[:2]
        lin = ref.linearize(kind, f, None, case16["p0"], valid, coeffs, huber_delta=1e9)      # rows as they are: no loss correction to undo
        Js.append(lin["J"][np.asarray(valid, bool)])
        Hsum += lin["H"]
    J = np.concatenate(Js)
    assert len(J) > 3000
    rows = np.arange(len(J) + 1) * 6
    cols = np.tile(np.arange(6), len(J))
    H = ref.ref_eval_hessian(rows, cols, J.ravel())
    Hn = J.T @ J
    sc = np.abs(Hn).max()
    np.testing.assert_allclose(H, Hn, rtol=1e-11, atol=1e-11 * sc)
    np.testing.assert_allclose(H, Hsum, rtol=1e-9, atol=1e-9 * sc)


def _cov_cloud(rng, n, extent, trace_lo, trace_hi, n_lidar=2):
    """PointXYZIWithCov records the way the mapper holds them: a few thousand points on a handful of surfaces (so voxels have many members),
    intensity = LiDAR id, a diagonal-dominant covariance whose trace spreads around the gate; some records share their trace exactly"""
    xyz = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    xyz[:, 2] = (0.05 * rng.standard_normal(n)).astype(np.float32)                # a ground sheet: dense voxels
    half = n // 2
    xyz[half:, 0] = (extent * 0.5 + 0.05 * rng.standard_normal(n - half)).astype(np.float32)   # and a wall
    xyz[half:, 2] = rng.uniform(0, 3, n - half).astype(np.float32)
    tr = rng.uniform(trace_lo, trace_hi, n)
    tr[rng.choice(n, n // 10, replace=False)] = 0.5 * (trace_lo + trace_hi)       # exact weight ties inside voxels
    d = rng.dirichlet([2.0, 2.0, 2.0], n) * tr[:, None]
    off = 0.1 * rng.standard_normal((n, 3)) * np.sqrt(d[:, [0, 0, 1]] * d[:, [1, 2, 2]])
    rec = np.zeros((n, 11), np.float32)
    rec[:, :3] = xyz
    rec[:, 3] = rng.integers(0, n_lidar, n)
    rec[:, 4] = d[:, 0]; rec[:, 5] = off[:, 0]; rec[:, 6] = off[:, 1]; rec[:, 7] = d[:, 1]; rec[:, 8] = off[:, 2]; rec[:, 9] = d[:, 2]
    rec[:, 10] = rec[:, 4] + rec[:, 7] + rec[:, 9]
    return rec


def test_voxel_grid_covariance_mloam_is_the_references(ref):
    """VERDICT r02 (row c, Missing #1): VoxelGridCovarianceMLOAM<PointT>::applyFilter compiled from the reference's OWN file
    (mloam_pcl/.../voxel_grid_covariance_mloam_impl.hpp:68-457) against the oracle's restatement -- the one the HIP filters are held to.
    Plain branch (PointXYZI: xyz mean over the members in std::sort's order, the LAST member's intensity = which LiDAR id a mixed voxel keeps) and
    covariance branch (PointXYZIWithCov: the |trace| >= threshold gate, w = threshold - trace, weighted mean, w^2-weighted covariance over (sum w)^2,
    the first-heaviest member's intensity, trace recomputed): same voxels in the same order, every output field bit for bit."""
    rng = np.random.default_rng(77)
    for n, extent, leaf in ((6000, 6.0, 0.4), (6000, 3.0, 0.2), (200, 2.0, 1.0), (1, 1.0, 0.4)):
        rec = _cov_cloud(rng, n, extent, 0.01, 0.9)
        plain_ref = ref.ref_voxel_filter(rec[:, :4], leaf)
        plain_orc = ref.voxel_grid_mloam_plain(rec[:, :4], leaf, member_order=0)
        assert plain_ref.shape == plain_orc.shape and len(plain_ref) >= 1
        assert np.array_equal(plain_ref.view(np.uint32), plain_orc.view(np.uint32))
        if n >= 6000:
            assert len(plain_ref) < n // 2                                              # voxels really have several members ...
            other = ref.voxel_grid_mloam_plain(rec[:, :4], leaf, member_order=1)
            assert not np.array_equal(other[:, 3], plain_ref[:, 3])                     # ... and the member order really decides the surviving id
        for thr in (0.6, 0.3, 2.0):
            cov_ref = ref.ref_voxel_filter(rec, leaf, thr)
            cov_orc = ref.voxel_grid_cov(rec, leaf, thr)
            assert cov_ref.shape == cov_orc.shape
            assert np.array_equal(cov_ref.view(np.uint32), cov_orc.view(np.uint32)), (n, leaf, thr)
    # a voxel whose members are ALL gated out keeps its place in the output with zero weight (valid_cnt -> 1, weight_total -> 1: impl.hpp:314-325)
    rec = _cov_cloud(rng, 500, 3.0, 0.7, 0.9)
    a, b = ref.ref_voxel_filter(rec, 0.4, 0.6), ref.voxel_grid_cov(rec, 0.4, 0.6)
    assert len(a) > 10 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and not a[:, :3].any()


def test_voxel_grid_points_on_voxel_faces_and_repeated_points(ref):
    """The same filter on input built to sit on the decision boundaries: coordinates that are exact multiples of the leaf size (which voxel a point on a face
    belongs to is floor(x * inverse_leaf_size) in f32, negative side included), a cloud whose minimum is itself such a multiple, and points repeated verbatim
    (equal keys by the hundred inside std::sort's unstable order, which decides the surviving intensity of the plain branch and the order of the f32 sums)."""
    rng = np.random.default_rng(78)
    for leaf in (0.4, 0.2, 0.25):
        n = 4000
        rec = _cov_cloud(rng, n, 5.0, 0.01, 0.9)
        k = np.round(rec[:, :3] / np.float32(leaf))                                     # snap a third of the points onto the faces, another third onto a 4x finer lattice
        rec[: n // 3, :3] = (k[: n // 3] * np.float32(leaf)).astype(np.float32)
        rec[n // 3: 2 * n // 3, :3] = (np.round(rec[n // 3: 2 * n // 3, :3] * (4.0 / leaf)) * np.float32(leaf / 4.0)).astype(np.float32)
        rec = np.ascontiguousarray(np.concatenate([rec, rec[:700], rec[300:900][::-1]]))       # and repeat 1300 of them verbatim
        rec[:, 3] = rng.integers(0, 2, len(rec)).astype(np.float32)                     # two LiDAR ids, mixed inside voxels
        plain_ref = ref.ref_voxel_filter(rec[:, :4], leaf)
        plain_orc = ref.voxel_grid_mloam_plain(rec[:, :4], leaf, member_order=0)
        assert plain_ref.shape == plain_orc.shape and len(plain_ref) > 100
        assert np.array_equal(plain_ref.view(np.uint32), plain_orc.view(np.uint32)), leaf
        for thr in (0.6, 2.0):
            cov_ref = ref.ref_voxel_filter(rec, leaf, thr)
            cov_orc = ref.voxel_grid_cov(rec, leaf, thr)
            assert cov_ref.shape == cov_orc.shape
            assert np.array_equal(cov_ref.view(np.uint32), cov_orc.view(np.uint32)), (leaf, thr)


def _features11(synth, feats, cov_scale, rng):
    """(m, 4) features -> (m, 11) PointXYZIWithCov records with a small per-point covariance (so that with_ua weighs them differently)"""
    out = np.zeros((len(feats), 11), np.float32)
    out[:, :4] = feats[:, :4]
    d = rng.uniform(0.2, 1.0, (len(feats), 3)) * cov_scale
    out[:, 4] = d[:, 0]; out[:, 7] = d[:, 1]; out[:, 9] = d[:, 2]
    out[:, 10] = out[:, 4] + out[:, 7] + out[:, 9]
    return out


@pytest.mark.parametrize("with_ua,gf_method,gf_ratio,frame_cnt", [(False, "wo_gf", 1.0, 1), (True, "wo_gf", 1.0, 0), (True, "gd_fix", 0.3, 0), (True, "rnd", 0.4, 7)])
def test_scan2map_optimization_is_the_references(ref, synth, case16, feats16, with_ua, gf_method, gf_ratio, frame_cnt):
    """VERDICT r02 (row c, Missing #2): the mapper's driver loop -- scan2MapOptimization, lidar_mapper_keyframe.cpp:423-639 -- compiled from the reference's
    OWN lines over a Ceres-shaped shim: kd-tree set-up, two outer iterations, the every-tenth-frame evalFullHessian + gf_ratio policy, goodFeatureMatching for
    corners then surfs, block assembly with extractCov / COV_MEASUREMENT, problem.Evaluate -> evalHessian -> evalDegenracy, ceres::Solve (30 iterations), the
    final covariance. The oracle's restatement ("2 outer x match once x LM", which the HIP mlh_scan2map is held to) must give the same number of residual
    blocks, the same LM bookkeeping and costs in both outer iterations, the same pose and the same cov_mapping."""
    rng = np.random.default_rng(5)
    surf, corner = feats16
    f_s, f_c = _features11(synth, surf, 0.01, rng), _features11(synth, corner, 0.01, rng)
    p0 = case16["p0"]
    got = ref.ref_scan2map(case16["surf_map"], case16["corner_map"], f_s, f_c, p0, with_ua=with_ua, gf_method=gf_method, gf_ratio=gf_ratio, seed=11, frame_cnt=frame_cnt)
    prm = ref.mapper_params(with_ua=with_ua, gf_method=gf_method, gf_ratio=gf_ratio, seed=11)
    want = ref.scan2map(ref.Map(case16["surf_map"]), ref.Map(case16["corner_map"]), f_s, f_c, p0, prm)
    assert len(got["solves"]) == len(want["outer"]) == 2
    for g, w in zip(got["solves"], want["outer"]):
        assert g["n_blocks"] == w["n_surf_sel"] + w["n_corner_sel"] and g["n_blocks"] > 300
        assert (g["lm_iterations"], g["successful_steps"], g["termination"]) == (w["lm_iterations"], w["successful_steps"], w["termination"])
        assert abs(g["initial_cost"] - w["initial_cost"]) <= 1e-12 * max(1.0, w["initial_cost"])
        assert abs(g["final_cost"] - w["final_cost"]) <= 1e-12 * max(1.0, w["final_cost"])
    assert np.linalg.norm(got["pose"] - want["pose"]) < 1e-12
    assert np.linalg.norm(got["pose"][:3] - p0[:3]) > 1e-3                                  # the optimisation really moved the pose
    if with_ua:                                                                             # cov_mapping = (J^T J)^-1 after the last solve (cpp:600-610)
        np.testing.assert_allclose(got["cov"], np.linalg.inv(want["H_final"]), rtol=1e-8, atol=1e-14)
    else:
        assert not got["cov"].any()


def test_scan2map_optimization_is_the_references_at_config2(ref, synth):
    """the same pin at BASELINE config 2's size (2 x 64 rings, the 500 k map: 12-14 k residual blocks per outer iteration): the reference's own lines and the
    oracle's restatement agree to the last bit of the pose (the GPU suite holds the HIP path against the same call, tests/test_gpu_parity_fullsize.py)"""
    import warnings
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
    ex = [ref.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    surf, corner = bench.fuse_features(synth, scans, ex)
    p0 = synth.perturbed_pose(gt, seed=43)
    got = ref.ref_scan2map(surf_map, corner_map, surf, corner, p0)
    want = ref.scan2map(ref.Map(surf_map), ref.Map(corner_map), surf, corner, p0, ref.mapper_params())
    assert [s["n_blocks"] for s in got["solves"]] == [o["n_surf_sel"] + o["n_corner_sel"] for o in want["outer"]] and got["solves"][0]["n_blocks"] > 10000
    assert [s["lm_iterations"] for s in got["solves"]] == [o["lm_iterations"] for o in want["outer"]]
    assert np.linalg.norm(got["pose"] - want["pose"]) < 1e-12


def test_track_cloud_is_the_references(ref, track_case):
    """LidarTracker::trackCloud (lidar_tracker.cpp:23-129) from the reference's own lines: two rounds of {matchCornerFromScan + matchSurfFromScan at the
    current estimate, LidarScanPlaneNormFactor / LidarScanEdgeFactorVector blocks under Huber(0.1), ceres::Solve with 4 iterations}; the oracle's
    restatement (which the HIP tracker is held to) walks the same path: block counts, LM bookkeeping, costs, pose."""
    tc = track_case
    for p0 in (np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.1, -0.05, 0.0, 0, 0, 0.004, 0.999992])):
        got = ref.ref_track_cloud(tc["corner_last"], tc["surf_last"], tc["corner_sharp"], tc["surf_flat"], p0)
        want = ref.track_cloud(tc["corner_last"], tc["surf_last"], tc["corner_sharp"], tc["surf_flat"], p0)
        solved = [o for o in want["outer"] if o["solved"]]
        assert len(got["solves"]) == len(solved) == 2
        for g, w in zip(got["solves"], solved):
            assert g["n_blocks"] == w["n_corner"] + w["n_surf"] and g["n_blocks"] > 100
            assert (g["lm_iterations"], g["termination"]) == (w["lm_iterations"], w["termination"])
            # 1e-9: the scan factors' f64 expressions are associated differently in the restatement (pinned at that level factor by factor above)
            assert abs(g["initial_cost"] - w["initial_cost"]) <= 1e-9 * max(1.0, w["initial_cost"])
            assert abs(g["final_cost"] - w["final_cost"]) <= 1e-9 * max(1.0, w["final_cost"])
        assert np.linalg.norm(got["pose"] - want["pose"]) < 1e-9
    # too few correspondences (a handful of current features): both rounds are skipped, the pose comes back unchanged (cpp:66-70)
    p0 = np.array([0.02, 0.0, 0.0, 0, 0, 0, 1.0])
    got = ref.ref_track_cloud(tc["corner_last"], tc["surf_last"], tc["corner_sharp"][:3], tc["surf_flat"][:4], p0)
    assert got["solves"] == [] and np.array_equal(got["pose"], p0)


def test_mappers_pose_chain_is_the_references(orc):
    """transformUpdate + transformAssociateToMap over Pose::operator* / Pose::inverse (lidar_mapper_keyframe.cpp:145-160, pose.cpp:99-113): the restatement
    (which the device kernel behind mlh_gn_solve_begin_chained follows operation for operation) against the reference's own lines, bit for bit."""
    if orc.ref_lib() is None:
        pytest.skip("no reference build")
    rng = np.random.default_rng(5)

    def rp(s):
        q = rng.normal(size=4)
        return np.concatenate([rng.normal(size=3) * s, q / np.linalg.norm(q)])

    for s in (0.01, 1.0, 50.0, 1e3):
        for _ in range(20):
            a, b, c = rp(s), rp(s), rp(s)
            np.testing.assert_array_equal(orc.pose_chain(a, b, c), orc.ref_pose_chain(a, b, c))
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    a = rp(3.0)
    assert np.abs(orc.pose_chain(a, ident, ident) - a).max() < 1e-15


def _window_rows(w, laser_filter=None):
    rows = np.zeros((len(w["types"]), 12))
    rows[:, 0] = w["ei"]; rows[:, 1] = w["fi"] + 1; rows[:, 2] = w["types"]; rows[:, 3:6] = w["points"]; rows[:, 6:12] = w["coeffs"]
    return rows if laser_filter is None else rows[laser_filter(rows)]


def test_odometry_window_assembly_is_the_references_pure_odometry(orc, synth):
    """Estimator::optimizeMap with ESTIMATE_EXTRINSIC = 0 (estimator.cpp:593-866 from the reference's own lines over the Ceres-shaped shim): one
    LidarPureOdom factor per selected feature on (para_pose_[0], para_pose_[i - pivot_idx], para_ex_pose_[n]), Huber(1.0), the pivot pose and EVERY extrinsic
    constant. The Jacobian evalResidual evaluates (-> evalDegenracy) and its cost equal the oracle's window normal equations -- the table the device reduces
    (mlh_pure_odom_normal_eq) is built the same way --, constant blocks with zero columns; ceres::Solve then lowers the cost and lands where Gauss-Newton on the
    oracle's normal equations lands (what mlh_pure_odom_gn_solve iterates)."""
    if orc.ref_lib() is None:
        pytest.skip("no reference build")
    import conftest
    n_frames, n_lidars = 2, 2
    w = conftest.make_window_case(synth, orc, n_frames, n_lidars)
    poses = np.vstack([w["pivot"][None, :], w["frames"]])
    got = orc.ref_optimize_map(poses, w["exts"], _window_rows(w), estimate_extrinsic=0, num_iterations=10)
    want = orc.pure_odom_normal_eq(w["types"], w["points"], w["coeffs"], None, w["fi"], w["ei"], w["pivot"], w["frames"], w["exts"], 1.0)
    D = 6 * (1 + n_frames + n_lidars)
    free = np.zeros(D, bool); free[6:6 * (1 + n_frames)] = True
    Hm = want["H"] * np.outer(free, free)
    assert got["n_blocks"] == len(w["types"])
    assert abs(got["cost"] - want["cost"]) <= 1e-12 * want["cost"]
    assert float(np.abs(got["H"] - Hm).max()) <= 1e-11 * float(np.abs(Hm).max())
    assert not got["H"][~free].any() and not got["H"][:, ~free].any()
    # the solve: constants untouched, cost lowered, and the minimiser Gauss-Newton on the oracle's normal equations reaches
    assert np.array_equal(got["poses"][0], w["pivot"]) and np.array_equal(got["exts"], w["exts"])
    assert got["solve"]["final_cost"] < 0.9 * got["solve"]["initial_cost"] and abs(got["solve"]["initial_cost"] - want["cost"]) <= 1e-12 * want["cost"]
    fr = w["frames"].copy()
    for _ in range(8):
        ne = orc.pure_odom_normal_eq(w["types"], w["points"], w["coeffs"], None, w["fi"], w["ei"], w["pivot"], fr, w["exts"], 1.0)
        idx = np.flatnonzero(free)
        d = np.linalg.solve(ne["H"][np.ix_(idx, idx)], -ne["g"][idx])
        fr = np.stack([orc.pose_plus(fr[i], d[6 * i:6 * i + 6]) for i in range(n_frames)])
    assert float(np.abs(got["poses"][1:, :3] - fr[:, :3]).max()) < 2e-4 and float(np.abs(got["poses"][1:, 3:] - fr[:, 3:]).max()) < 2e-5
    end = orc.pure_odom_normal_eq(w["types"], w["points"], w["coeffs"], None, w["fi"], w["ei"], w["pivot"], got["poses"][1:], w["exts"], 1.0)
    assert abs(end["cost"] - got["solve"]["final_cost"]) <= 1e-9 * end["cost"]


def test_odometry_window_assembly_is_the_references_online_calibration(orc, synth):
    """ESTIMATE_EXTRINSIC = 1: only the reference LiDAR's features become window factors (on para_ex_pose_[IDX_REF], which is constant); the other LiDARs'
    pivot-frame features accumulate in cumu_*_map_features_ and turn into LidarOnlineCalib factors on their own extrinsic every N_CUMU_FEATURE-th frame."""
    if orc.ref_lib() is None:
        pytest.skip("no reference build")
    import conftest
    n_frames, n_lidars = 1, 2
    w = conftest.make_window_case(synth, orc, n_frames, n_lidars)
    poses = np.vstack([w["pivot"][None, :], w["frames"]])
    rows = _window_rows(w)
    is_ref = rows[:, 0] == 0
    calib = rows[~is_ref].copy()
    calib[:, 1] = 0                                                     # LiDAR 1's features play the pivot frame's (frame index pivot_idx) features of LiDAR 1
    D = 6 * (1 + n_frames + n_lidars)
    m = is_ref
    win = orc.pure_odom_normal_eq(w["types"][m], w["points"][m], w["coeffs"][m], None, w["fi"][m], w["ei"][m], w["pivot"], w["frames"], w["exts"], 1.0)
    free = np.ones(D, bool); free[:6] = False; free[6 * (1 + n_frames):6 * (2 + n_frames)] = False      # pivot and the reference extrinsic are constant
    for frame_cnt, with_calib in ((10, True), (7, False)):
        got = orc.ref_optimize_map(poses, w["exts"], np.vstack([rows[is_ref], calib]), estimate_extrinsic=1, frame_cnt=frame_cnt, n_cumu_feature=10, num_iterations=6)
        H = win["H"] * np.outer(free, free)
        cost = win["cost"]
        n_blocks = int(is_ref.sum())
        if with_calib:
            e1 = slice(6 * (2 + n_frames), 6 * (3 + n_frames))
            for kind, k in (("s", 0), ("c", 1)):
                sel = calib[:, 2] == k
                f4 = np.zeros((int(sel.sum()), 4), np.float32)            # the calibration factors take the point as a double: keep the rows exactly representable
                pts = calib[sel, 3:6]
                assert np.array_equal(pts.astype(np.float32).astype(np.float64), pts)
                f4[:, :3] = pts
                lin = orc.linearize(kind, f4, np.full(len(f4), 0.0075), w["exts"][1], np.ones(len(f4), np.uint8), calib[sel, 6:12], huber_delta=1.0)
                H[e1, e1] += lin["H"]; cost += lin["cost"]; n_blocks += int(sel.sum())
        assert got["n_blocks"] == n_blocks, frame_cnt
        assert abs(got["cost"] - cost) <= 1e-11 * cost
        assert float(np.abs(got["H"] - H).max()) <= 1e-10 * float(np.abs(H).max())
        assert np.array_equal(got["poses"][0], w["pivot"]) and np.array_equal(got["exts"][0], w["exts"][0])
        assert (not np.array_equal(got["exts"][1], w["exts"][1])) == with_calib           # the extrinsic moves only when its factors were added
        assert got["solve"]["final_cost"] < got["solve"]["initial_cost"]


def test_odometry_good_feature_matching_is_the_references(ref, synth):
    """Estimator::goodFeatureMatching + evaluateFeatJacobian (estimator.cpp:1273-1517: the ODOMETRY's selection in front of the window's residual blocks, ODOM_GF_RATIO
    = 0.8 in every shipped configuration) compiled from the reference's own lines against the oracle's restatement: the same features in the same order for surf and
    corner features, ratios 1.0 (match everything), 0.8 (subsets of one: random picks among the matching features until the count is reached or the pool is empty),
    0.3 / 0.05 (scored subsets of 3 / 20: the heap decides; the estimator's ten-draw limit does not end the selection, its loop starts over)."""
    import conftest
    import importlib
    w = conftest.make_window_case(synth, ref, 1, 2)
    case = conftest._make_case(synth, "50k", 16, 2)
    Tinv = np.linalg.inv(synth.pose_to_mat(case["gt"]))
    maps = [np.ascontiguousarray(synth.transform_points(m[:, :3], Tinv).astype(np.float32)) for m in (case["surf_map"], case["corner_map"])]      # the local map in the pivot frame
    feats = conftest.features_from_extraction(synth, case["scans"][:1], lambda s: ref.extract(s.points, s.scan_start, s.scan_end))
    pivot, pose_i, ext = w["pivot"], w["frames"][0], w["exts"][1]
    n_checked = 0
    for kind, mp, f in (("s", maps[0], feats[0]), ("c", maps[1], feats[1])):
        om = ref.Map(mp)
        for ratio in (1.0, 0.8, 0.3, 0.05):
            for seed in (1, 7):
                r = ref.ref_odom_good_feature_matching(kind, mp, f, pivot, pose_i, ext, ratio, seed)
                o = ref.odom_good_feature_matching(om, kind, f, r["rel_pose"], pivot, pose_i, ext, ratio, seed)
                assert np.array_equal(r["sel"], o["sel"]), (kind, ratio, seed, len(r["sel"]), len(o["sel"]))
                assert len(r["sel"]) > 50 and len(set(r["sel"].tolist())) == len(r["sel"])
                if ratio == 1.0:
                    assert np.array_equal(r["sel"], np.flatnonzero(o["matched"]))
                n_checked += 1
    assert n_checked == 16
    # the row that is scored: a surf feature's LidarPureOdomPlaneNormFactor frame block, a corner feature's (1 0 0 0 0 0)
    o = ref.odom_good_feature_matching(ref.Map(maps[1]), "c", feats[1], r["rel_pose"], pivot, pose_i, ext, 0.3, 1)
    assert np.array_equal(o["jaco"][o["matched"].astype(bool)], np.tile([1.0, 0, 0, 0, 0, 0], (int(o["matched"].sum()), 1)))


def test_random_problems_against_the_references_own_lines(ref):
    """scripts/soak_ref_pin.py, three random problems per family (extractCloud, match*PointFromMap, segmentCloud, applyFilter, scan2MapOptimization, trackCloud,
    goodFeatureMatching): the oracle equals the reference's own lines beyond the fixed cases above. The long runs (1 000 per family) are in profiles/r04_soak.txt."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_ref_pin.py"), "3", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all equal" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_random_problems_of_the_hard_scene_family_against_the_references_own_lines(ref):
    """The same on the "hard" scene family (synth.scene_family: poles and trunks, noisy vegetation blobs, slabs at grazing incidence, maps with exactly duplicated
    points and a patch of four-fold density -- where the line test lambda_2 > 3 lambda_1, feature_extract.hpp:688-693, and the plane gate, :823-840, sit near their
    thresholds): two random problems per scene-based family. The long runs are in profiles/r06_soak.txt."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MLOAM_SCENE_FAMILY="hard")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_ref_pin.py"), "2", "17", "match,scan2map,track,select,downsample"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "all equal" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
