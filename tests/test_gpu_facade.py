"""The C++ facade (m-loam_amd/host/mloam_facade.hpp: FeatureExtract / MapIndex / scan2MapOptimization with the reference's
signatures) gives the same answers as the oracle, through a real C++ executable linked against libmloam_hip.so."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_selftest(tmp_path, orc, case16, feats16, track_case):
    exe = os.path.join(ROOT, "m-loam_amd", "host", "facade_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe), "-s"], check=True)
    sc = case16["scans"][0]
    d = str(tmp_path)
    sc.points.astype(np.float32).tofile(os.path.join(d, "scan.f32"))
    np.concatenate([sc.scan_start, sc.scan_end]).astype(np.int32).tofile(os.path.join(d, "rings.i32"))
    case16["surf_map"].astype(np.float32).tofile(os.path.join(d, "surf_map.f32"))
    case16["corner_map"].astype(np.float32).tofile(os.path.join(d, "corner_map.f32"))
    feats16[0].astype(np.float32).tofile(os.path.join(d, "surf.f32"))
    feats16[1].astype(np.float32).tofile(os.path.join(d, "corner.f32"))
    case16["p0"].astype(np.float64).tofile(os.path.join(d, "pose.f64"))
    for name in ("corner_last", "surf_last", "corner_sharp", "surf_flat"):
        track_case[name].astype(np.float32).tofile(os.path.join(d, f"trk_{name}.f32"))
    for name, scn in zip(("prev", "cur"), track_case["scans"]):
        scn.points.astype(np.float32).tofile(os.path.join(d, f"trk_scan_{name}.f32"))
        np.concatenate([scn.scan_start, scn.scan_end]).astype(np.int32).tofile(os.path.join(d, f"trk_rings_{name}.i32"))
    import importlib
    synth_mod = importlib.import_module("m-loam_amd.synth")
    rng = np.random.default_rng(3)
    raw = sc.points.copy()
    raw[:, 3] = 0.0
    mclut = rng.random(len(raw)) < 0.1
    raw[mclut, :3] *= rng.uniform(0.5, 1.3, (int(mclut.sum()), 1)).astype(np.float32)
    raw = raw[rng.permutation(len(raw))]
    raw.astype(np.float32).tofile(os.path.join(d, "raw_cloud.f32"))
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    labels = np.fromfile(os.path.join(d, "out_labels.i32"), np.int32)
    ref = orc.extract(sc.points, sc.scan_start, sc.scan_end)
    assert np.array_equal(labels, ref["label"])
    lf = np.fromfile(os.path.join(d, "out_less_flat.f32"), np.float32).reshape(-1, 4)      # extractCloud's thinned less-flat cloud
    assert lf.shape == ref["less_flat_ds"].shape
    np.testing.assert_array_equal(lf.view(np.uint32), ref["less_flat_ds"].view(np.uint32))      # summed along std::sort's member order: the reference's bits
    valid = np.fromfile(os.path.join(d, "out_valid_surf.u8"), np.uint8)
    v, _ = orc.Map(case16["surf_map"]).match("s", feats16[0], case16["p0"])
    assert np.array_equal(valid, v)
    pose = np.fromfile(os.path.join(d, "out_pose.f64"), np.float64)
    s2m = orc.scan2map(orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"]), feats16[0], feats16[1], case16["p0"], orc.mapper_params())
    assert np.linalg.norm(pose[:3] - s2m["pose"][:3]) < 1e-7 and np.linalg.norm(pose[3:] - s2m["pose"][3:]) < 1e-7
    counts = np.fromfile(os.path.join(d, "out_counts.i32"), np.int32).reshape(-1, 3)
    for c, o in zip(counts, s2m["outer"]):
        assert tuple(c) == (o["n_surf_sel"], o["n_corner_sel"], o["lm_iterations"])
    # ---- round 4: the reference's own front-end loop (estimator.cpp:249-263) -- ONE FeatureExtract and ONE ImageSegmenter without a context of their own,
    #      4 OpenMP threads on 4 different scans, three frames -- gives every LiDAR exactly what the same calls give one after the other
    rv = np.fromfile(os.path.join(d, "out_reentrant.i32"), np.int32)
    assert len(rv) == 13 and rv[12] == 4, rv                 # four LiDARs were really served by four threads
    for i in range(4):
        assert rv[3 * i] == 1 and rv[3 * i + 1] == 1 and rv[3 * i + 2] > 10000, (i, rv)
    # round 6: the same loop from ONE thread through the facade's own lanes (FrontEndLanes) -- every LiDAR's clouds as the one-after-the-other run
    lv = np.fromfile(os.path.join(d, "out_lanes.i32"), np.int32)
    assert lv.tolist() == [4, 4, 1, 1, 1, 1], lv            # ... FeatureExtract::sendAhead: same clouds, the look-ahead served the upload; the lanes' device-resident front end == one Device's
    # FeatureExtract::calTimestamp (feature_extract.cpp:54-114) of LiDAR 0 against a plain restatement of its unwrapping and against the reference's own lines
    st = np.fromfile(os.path.join(d, "out_timestamps.f32"), np.float32)
    assert len(st) == len(raw)
    f32 = np.float32
    ori_all = (-np.arctan2(raw[:, 1], raw[:, 0])).astype(np.float32)
    PI = float(np.pi)
    # float variables against double constants, as the reference's expressions evaluate: sums and comparisons in double, results stored back into floats
    start, end = ori_all[0], f32(float(ori_all[-1]) + 2 * PI)
    if float(f32(end - start)) > 3 * PI:
        end = f32(float(end) - 2 * PI)
    elif float(f32(end - start)) < PI:
        end = f32(float(end) + 2 * PI)
    exp = np.empty(len(raw), np.float32)
    half = False
    for i, o in enumerate(ori_all):
        if not half:
            if float(o) < float(start) - PI / 2:
                o = f32(float(o) + 2 * PI)
            elif float(o) > float(start) + PI * 3 / 2:
                o = f32(float(o) - 2 * PI)
            if float(f32(o - start)) > PI:
                half = True
        else:
            o = f32(float(o) + 2 * PI)
            if float(o) < float(end) - PI * 3 / 2:
                o = f32(float(o) + 2 * PI)
            elif float(o) > float(end) + PI / 2:
                o = f32(float(o) - 2 * PI)
        exp[i] = f32(f32(o - start) / f32(end - start)) * f32(0.1)
    # a sanity bound only: numpy's arctan2 differs from libm's atan2f in the last ulp here and there, and a point (or the sweep's end angle) within an ulp of one of
    # the unwrapping thresholds then takes the other branch -- on this cloud twelve points of one azimuth do. The pin is the reference's own lines, below.
    assert np.mean(np.abs(st - exp) > 2e-6) < 2e-3
    if orc.ref_lib() is not None:                                # ... the reference's own lines are the pin: the same float for every point
        assert np.array_equal(st.view(np.uint32), orc.ref_cal_timestamp(raw[:, :3], 0.1).view(np.uint32))
    # ---- round 4: PipelinedMapper (the overlap's precondition as code): every pose equals the one-frame-at-a-time loop's -- including the frame whose keyframe
    #      prediction was wrong and that was therefore solved again on the rebuilt map -- and all three paths were taken
    pm = np.fromfile(os.path.join(d, "out_pipelined_mapper.f64"), np.float64)
    worst, overlapped, waited, redone, n_kf, n_kf_ref, travelled = pm[56:63]
    assert worst < 1e-9, pm[56:]
    assert overlapped >= 2 and waited >= 1 and redone >= 1 and n_kf == n_kf_ref >= 3, pm[56:]
    assert abs(travelled - 3 * 0.34) < 0.02                      # the solved poses follow the sensor's motion (0.34 m per frame), not the drifting odometry (0.31)
    # the same on six random 40-frame sequences (random steps, drift, keyframe distances): every pose the plain synchronous loop's, every path of the loop taken
    pr = np.fromfile(os.path.join(d, "out_pipelined_mapper_random.f64"), np.float64)
    assert pr[0] < 1e-9 and pr[5] == 0, pr
    assert pr[1] >= 20 and pr[2] >= 10 and pr[4] >= 20, pr        # staged beside the solve / waited for the pose / keyframes: all well exercised
    # ---- round 2: ActiveFeatureSelection::evalFullHessian -> logDet -> gf_ratio policy through the facade
    afs = np.fromfile(os.path.join(d, "out_afs.f64"), np.float64)
    cov6 = np.array([0.01, 0, 0, 0.02, 0, 0.03], np.float32)
    def with_cov(f):
        a = np.zeros((len(f), 11), np.float32); a[:, :4] = f[:, :4]; a[:, 4:10] = cov6; a[:, 10] = 0.06
        return a
    fs11, fc11 = with_cov(feats16[0]), with_cov(feats16[1])
    Href, nref = orc.eval_full_hessian(orc.Map(case16["surf_map"]), "s", fs11, case16["p0"])
    Href, nref = orc.eval_full_hessian(orc.Map(case16["corner_map"]), "c", fc11, case16["p0"], Href, nref)
    assert int(afs[36]) == nref
    assert float(np.abs(afs[:36].reshape(6, 6) - Href).max()) <= 1e-9 * float(np.abs(Href).max())
    assert abs(afs[37] - orc.logdet(Href)) < 1e-9 * abs(orc.logdet(Href))
    k = 38
    for method in ("wo_gf", "rnd", "fps", "gd_fix", "gd_float"):
        for thre in (afs[37] - 1.0, afs[37] + 1.0):
            assert afs[k] == orc.gf_ratio_policy(method, 0.2, afs[37], thre, -1.0)
            k += 1
    # goodFeatureMatching through the facade + the reference's per-feature factor objects on the selection
    refsel = orc.good_feature_matching(orc.Map(case16["surf_map"]), "s", fs11, case16["p0"], orc.mapper_params(with_ua=True, gf_method="gd_fix", gf_ratio=0.2, seed=11))
    rows = np.fromfile(os.path.join(d, "out_sel_rows.f64"), np.float64).reshape(-1, 9)
    assert np.array_equal(rows[:, 0].astype(int), refsel["sel"])
    np.testing.assert_allclose(np.fromfile(os.path.join(d, "out_sel_H.f64"), np.float64).reshape(6, 6), refsel["H"], rtol=1e-9, atol=1e-9)
    vs_, cs_ = orc.Map(case16["surf_map"]).match("s", feats16[0], case16["p0"])
    for row in rows[::7]:
        i = int(row[0])
        rr, JJ = orc.factor_eval("s", feats16[0][i, :3].astype(np.float64), cs_[i, :4], 0.06, case16["p0"])
        assert abs(row[1] - rr) < 1e-12 and np.allclose(row[2:], JJ, rtol=1e-11, atol=1e-12)
    crows = np.fromfile(os.path.join(d, "out_corner_rows.f64"), np.float64).reshape(-1, 9)
    vc_, cc_ = orc.Map(case16["corner_map"]).match("c", feats16[1], case16["p0"])
    assert np.array_equal(crows[:, 0].astype(int), np.flatnonzero(vc_))
    for row in crows[::5]:
        i = int(row[0])
        rr, JJ = orc.factor_eval("c", feats16[1][i, :3].astype(np.float64), cc_[i], 0.0075, case16["p0"])
        assert abs(row[1] - rr) < 1e-12 and np.allclose(row[2:], JJ, rtol=1e-11, atol=1e-12)
    # round 3: FramePipeline -- frame 0 submitted with the start pose, frame 1 chained on the device while its maps are staged beside frame 0's solve
    pl = np.fromfile(os.path.join(d, "out_pipeline.f64"), np.float64).reshape(4, 7)
    import importlib
    mla = importlib.import_module("m-loam_amd")
    cpl = mla.Context(0)
    try:
        cpl.map_set_pair(case16["surf_map"], case16["corner_map"])
        cpl.features_set(mla.SURF, feats16[0]); cpl.features_set(mla.CORNER, feats16[1])
        w0, _ = cpl.gn_solve(case16["p0"], 3, want_stats=False)
        w1, _ = cpl.gn_solve(orc.pose_chain(w0, pl[2], pl[3]), 3, want_stats=False)
    finally:
        cpl.close()
    assert np.array_equal(pl[0], w0) and np.array_equal(pl[1], w1)
    # WindowFactorTable + evalWindowNormalEquations: surf matched as (frame 1, LiDAR 0, N_NEIGH 5), corner as (frame 1, LiDAR 1, N_NEIGH 10), CHECK_FOV
    wn = np.fromfile(os.path.join(d, "out_window_ne.f64"), np.float64)
    D = 24
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    tabs = [[], [], [], [], []]
    for ch, om_cloud, f, kn, lid in (("s", case16["surf_map"], feats16[0], 5, 0), ("c", case16["corner_map"], feats16[1], 10, 1)):
        v_w, co_w = orc.Map(om_cloud).match(ch, f, case16["p0"], n_neigh=kn, check_fov=True)
        m_w = v_w.astype(bool)
        tabs[0].append(np.full(m_w.sum(), 0 if ch == "s" else 1, np.int32)); tabs[1].append(f[m_w, :3].astype(np.float64)); tabs[2].append(co_w[m_w])
        tabs[3].append(np.zeros(m_w.sum(), np.int32)); tabs[4].append(np.full(m_w.sum(), lid, np.int32))
    tabs = [np.concatenate(a) for a in tabs]
    wref = orc.pure_odom_normal_eq(tabs[0], tabs[1], tabs[2], None, tabs[3], tabs[4], ident, case16["p0"][None, :], np.stack([ident, ident]), 1.0)
    assert int(wn[D * D + D + 1]) == wref["count"] == len(tabs[0])
    assert float(np.abs(wn[:D * D].reshape(D, D) - wref["H"]).max()) <= 1e-9 * float(np.abs(wref["H"]).max())
    assert float(np.abs(wn[D * D:D * D + D] - wref["g"]).max()) <= 1e-9 * float(np.abs(wref["g"]).max())
    # WindowFactorTable::goodFeatureMatching: the odometry's selection (Estimator::goodFeatureMatching) -- the same picks in the same order as the oracle, and the
    # table holds exactly the selected correspondences
    so = np.fromfile(os.path.join(d, "out_odom_selection.i32"), np.int32)
    ns, nc, nres = int(so[0]), int(so[1]), int(so[2])
    sel_s, sel_c = so[3:3 + ns], so[3 + ns:3 + ns + nc]
    assert nres == ns + nc and ns > 100 and nc > 30
    rs = orc.odom_good_feature_matching(orc.Map(case16["surf_map"]), "s", feats16[0], case16["p0"], ident, case16["p0"], ident, 0.8, 11)
    rc_ = orc.odom_good_feature_matching(orc.Map(case16["corner_map"]), "c", feats16[1], case16["p0"], ident, case16["p0"], ident, 0.3, 12)
    assert np.array_equal(sel_s, rs["sel"]) and np.array_equal(sel_c, rc_["sel"])
    # ImageSegmenter facade
    seg = orc.segment_cloud(raw, orc.seg_params())
    so = np.fromfile(os.path.join(d, "out_seg_cloud.f32"), np.float32).reshape(-1, 4)
    assert so.shape == seg["cloud"].shape and np.array_equal(so.view(np.uint32), seg["cloud"].view(np.uint32))
    si = np.fromfile(os.path.join(d, "out_seg_info.i32"), np.int32)
    assert np.array_equal(si[:16], seg["scan_start"]) and np.array_equal(si[16:], seg["scan_end"])
    # VoxelGridCovarianceMLOAM facade: 48-byte PointXYZIWithCov records in and out
    ds = np.fromfile(os.path.join(d, "out_map_ds.f32"), np.float32).reshape(-1, 11)
    m11 = np.zeros((len(case16["surf_map"]), 11), np.float32)
    m11[:, :3] = case16["surf_map"][:, :3]
    ref = orc.voxel_grid_cov(m11, 0.8, 1.0)
    assert ds.shape == ref.shape
    np.testing.assert_array_equal(ds.view(np.uint32), ref.view(np.uint32))
    # cloudUCTAssociateToMap facade (the self-test labels the surf features alternately LiDAR 0 / 1)
    km = np.fromfile(os.path.join(d, "out_kf_map.f32"), np.float32).reshape(-1, 11)
    kf = np.zeros((len(feats16[0]), 11), np.float32)
    kf[:, :3] = feats16[0][:, :3]
    kf[:, 3] = np.arange(len(kf)) & 1
    ext = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0.1, -0.5, 0, 0, 0, 0.0998334166, 0.9950041653]])
    ext_cov = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
    ref = orc.cloud_uct_associate_to_map(kf, pose, np.eye(6) * 1e-5,   # the self-test uses the pose scan2MapOptimization just refined
                                         ext, ext_cov, np.diag([0.0025] * 3), True, 0.6)
    assert km.shape == ref.shape
    assert np.array_equal(km[:, :4].view(np.uint32), ref[:, :4].view(np.uint32))
    np.testing.assert_allclose(km[:, 4:], ref[:, 4:], rtol=2e-5, atol=1e-9)
    # LidarPureOdomBatchFactor: Ceres-shaped Evaluate over [pivot, frame 1, frame 2, ext 0, ext 1]
    res = np.fromfile(os.path.join(d, "out_odom_res.f64"), np.float64)
    nr = len(res)
    jac = np.fromfile(os.path.join(d, "out_odom_jac.f64"), np.float64).reshape(5, nr, 7)
    vidx = np.flatnonzero(v)
    assert nr == len(vidx)
    _, coeffs = orc.Map(case16["surf_map"]).match("s", feats16[0], case16["p0"])
    pivot = np.array([0.3, -0.2, 0.1, 0.0, 0.0, 0.0499791693, 0.9987502604])
    f1 = pose.copy(); f2 = pose.copy(); f2[0] += 0.25
    par = [pivot, f1, f2, np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.1, -0.5, 0.02, 0.0, 0.0, 0.0998334166, 0.9950041653])]
    for i in range(0, nr, 11):
        fb, eb = 1 + i % 2, 3 + (i // 2) % 2
        rr, JJ = orc.pure_odom_eval("s", feats16[0][vidx[i], :3].astype(np.float64), coeffs[vidx[i], :4], pivot, par[fb], par[eb])
        assert abs(res[i] - rr) < 1e-10
        np.testing.assert_allclose(jac[0, i], JJ[0], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(jac[fb, i], JJ[1], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(jac[eb, i], JJ[2], rtol=1e-10, atol=1e-10)
        for b in range(1, 5):
            if b not in (fb, eb):
                assert not jac[b, i].any()
    # LidarTracker::trackCloud facade
    tp = np.fromfile(os.path.join(d, "out_track_pose.f64"), np.float64)
    rt = orc.track_cloud(track_case["corner_last"], track_case["surf_last"], track_case["corner_sharp"], track_case["surf_flat"],
                         np.array([0, 0, 0, 0, 0, 0, 1.0]))
    assert np.linalg.norm(tp[:3] - rt["pose"][:3]) < 1e-9 and np.linalg.norm(tp[3:] - rt["pose"][3:]) < 1e-9
    # window-map facade: pcl::transformPointCloud + pcl::VoxelGrid<PointI> with the reference's call sequence
    wm = np.fromfile(os.path.join(d, "out_window_map.f32"), np.float32).reshape(-1, 4)
    ref_wm = orc.voxel_grid(orc.transform_point_cloud(track_case["scans"][0].points, ext[1]), 0.3)
    assert wm.shape == ref_wm.shape
    np.testing.assert_array_equal(wm.view(np.uint32), ref_wm.view(np.uint32))
    # TransformToEnd facade (the tracker scans carry ring ids only: frac(intensity) = 0 -> s = 0, the point goes through T^-1)
    und = np.fromfile(os.path.join(d, "out_undistorted.f32"), np.float32).reshape(-1, 4)
    pu = np.array([0.35, -0.12, 0.02, 0.0, 0.0, 0.0130895956, 0.9999143276])
    ref_u = orc.transform_to_end(track_case["scans"][1].points, pu, True)
    assert und.shape == ref_u.shape
    np.testing.assert_allclose(und[:, :3], ref_u[:, :3], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(und[:, 3], ref_u[:, 3])
    # the device-resident front end of the facade == the same calls through the Python binding
    import importlib
    mla = importlib.import_module("m-loam_amd")
    prev, cur = track_case["scans"]
    c = mla.Context(0)
    try:
        c.scan_upload(prev.points, prev.scan_start, prev.scan_end); c.extract_run(); c.extract_voxel_run(0.2); c.track_set_from_scan(1)
        c.scan_upload(cur.points, cur.scan_start, cur.scan_end); c.extract_run(); c.extract_voxel_run(0.2); c.track_set_from_scan(0)
        pose_dev, _ = c.track_cloud(np.array([0, 0, 0, 0, 0, 0, 1.0]))
        np.testing.assert_array_equal(np.fromfile(os.path.join(d, "out_track_pose_dev.f64"), np.float64), pose_dev)
        assert np.linalg.norm(pose_dev[:3] - track_case["motion"][:3]) < 0.08
        ext_cov[1] = np.diag([0.0025] * 3 + [0.00030461] * 3)
        c.fuse_reset(); c.fuse_add_scan(1, ext[1])
        c.scan_upload(prev.points, prev.scan_start, prev.scan_end); c.extract_run(); c.extract_voxel_run(0.2); c.fuse_add_scan(0, ext[0])
        kept = [c.downsample_current_scan(k, c.fused_cloud(k), leaf, ext, ext_cov, np.diag([0.0025] * 3), True, 0.6, fetch=False)
                for k, leaf in ((mla.SURF, 0.4), (mla.CORNER, 0.2))]
        assert list(np.fromfile(os.path.join(d, "out_fused_kept.i32"), np.int32)) == kept + kept and min(kept) > 50      # single calls, then the pair call
    finally:
        c.close()


def test_reference_call_sites_on_the_gpu(tmp_path, orc, case16, feats16):
    """The reference's own call sites, cut verbatim and compiled against the facade (tests/host/refcut; built where /root/reference exists, the executable travels):
    estimator.cpp:248-270 -- ONE FeatureExtract + ONE ImageSegmenter, NUM_OF_LASER OpenMP threads -- on four scans; lidar_mapper_keyframe.cpp:433-434 through
    MapIndex::Ptr; the match functions filling the REFERENCE's PointPlaneFeature at the REFERENCE's Pose (validity == the oracle's); cpp:537-571 building one residual
    block per matched feature in a ceres::Problem that owns them, each block's r / J == the batched device evaluation (mlh_linearize) to 1e-9."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_refcut", os.path.join(ROOT, "tests", "host", "refcut", "build_refcut.py"))
    rc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rc)
    exe = rc.build()
    if exe is None:
        if os.environ.get("MLOAM_REQUIRE_REF") == "1":
            pytest.fail("MLOAM_REQUIRE_REF=1 but neither /root/reference nor the prebuilt tests/host/refcut/_build/refcut_selftest is here")
        pytest.skip("no reference tree and no prebuilt refcut_selftest")
    sc = case16["scans"][0]
    d = str(tmp_path)
    case16["surf_map"].astype(np.float32).tofile(os.path.join(d, "surf_map.f32"))
    case16["corner_map"].astype(np.float32).tofile(os.path.join(d, "corner_map.f32"))
    feats16[0].astype(np.float32).tofile(os.path.join(d, "surf.f32"))
    feats16[1].astype(np.float32).tofile(os.path.join(d, "corner.f32"))
    case16["p0"].astype(np.float64).tofile(os.path.join(d, "pose.f64"))
    rng = np.random.default_rng(3)
    raw = sc.points.copy()
    raw[:, 3] = 0.0
    mclut = rng.random(len(raw)) < 0.1
    raw[mclut, :3] *= rng.uniform(0.5, 1.3, (int(mclut.sum()), 1)).astype(np.float32)
    raw = raw[rng.permutation(len(raw))]
    raw.astype(np.float32).tofile(os.path.join(d, "raw_cloud.f32"))
    r = subprocess.run([exe, "gpu", d], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr + r.stdout
    v = np.fromfile(os.path.join(d, "refcut_verdict.i32"), np.int32)
    assert len(v) == 13, v
    for i in range(4):
        assert v[2 * i] == 1 and v[2 * i + 1] > 10000, (i, v)            # every LiDAR's clouds == the sequential, bound-object calls
    assert v[8] == 4, v                                                   # four LiDARs were really served by four threads
    assert v[9] > 0 and v[10] > 0, v                                      # total_corner_feature_, total_surf_feature_ of the reference's collection loop
    vs = np.fromfile(os.path.join(d, "refcut_valid_surf.u8"), np.uint8)
    vc = np.fromfile(os.path.join(d, "refcut_valid_corner.u8"), np.uint8)
    want_s, _ = orc.Map(case16["surf_map"]).match("s", feats16[0], case16["p0"])
    want_c, _ = orc.Map(case16["corner_map"]).match("c", feats16[1], case16["p0"])
    assert np.array_equal(vs, want_s) and np.array_equal(vc, want_c)
    assert v[11] == int(want_s.sum() + want_c.sum()) and v[12] == 1, v    # one residual block per matched feature, all released with the Problem
    dr, dJ = np.fromfile(os.path.join(d, "refcut_diff.f64"), np.float64)
    assert dr < 1e-9 and dJ < 1e-9, (dr, dJ)
