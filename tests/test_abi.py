"""The C-ABI library loads on a CPU-only box and exports every symbol include/mloam_hip.h declares; host-only entry points
agree with the oracle; with no GPU the context constructor fails loudly (no CPU fallback in the product path)."""
import ctypes
import os
import re

import numpy as np
import pytest


def _declared_symbols(header_path):
    txt = open(header_path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mlh_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(mla):
    lib = mla.load_library()
    declared = _declared_symbols(mla.HEADER_PATH)
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(mla.EXPORTED_SYMBOLS) == declared


def test_version_and_structs(mla):
    lib = mla.load_library()
    assert b"gfx950" in lib.mlh_version()
    o = mla.default_opts()
    assert (o.min_match_sq_dis, o.min_plane_dis, o.huber_delta, o.map_eig_thre, o.max_outer, o.max_lm_iterations) == (1.0, pytest.approx(0.2), 0.1, 100.0, 2, 30)
    assert ctypes.sizeof(mla.IterStat) == 6 * 4 + 8 * (2 + 6 + 36 + 6 + 7)


def test_host_helpers_match_oracle(mla, orc):
    rng = np.random.default_rng(0)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x = np.concatenate([rng.normal(size=3), q])
    d = rng.normal(size=6) * 0.1
    np.testing.assert_allclose(mla.pose_plus(x, d), orc.pose_plus(x, d), atol=1e-15)
    Qm, _ = np.linalg.qr(rng.normal(size=(6, 6)))
    H = Qm @ np.diag([3.0, 50.0, 200.0, 1e3, 1e4, 1e5]) @ Qm.T
    a, b = mla.eval_degeneracy(H, 100.0), orc.eval_degeneracy(H, 100.0)
    assert a["is_degenerate"] and b["is_degenerate"]
    np.testing.assert_allclose(a["eigval"], b["eigval"], rtol=1e-10)
    np.testing.assert_allclose(a["V_update"], b["V_update"], atol=1e-9)
    np.testing.assert_allclose(mla.pose_plus(x, d, a["V_update"]), orc.pose_plus(x, d, b["V_update"]), atol=1e-12)


def test_no_cpu_fallback(mla):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mla.MlhError):
        mla.Context(0)


def test_product_does_not_reference_the_oracle():
    """the product path must never import, link or load anything under oracle/"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "m-loam_amd")
    for dp, _, files in os.walk(pkg):
        if os.sep + "build" in dp or os.sep + "lib" in dp or "__pycache__" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("oracle-free", ""), os.path.join(dp, f)


def test_compound_pose_with_cov_host(mla, orc):
    """mlh_compound_pose_with_cov is host arithmetic (no GPU needed): it must agree with the oracle's restatement of
    compoundPoseWithCov (associate_uct.hpp:90-147), which test_oracle_numerics pins against Monte-Carlo sampling."""
    import numpy as np
    rng = np.random.default_rng(2)
    for _ in range(5):
        p1 = np.concatenate([rng.uniform(-5, 5, 3), rng.normal(size=4)]); p1[3:] /= np.linalg.norm(p1[3:])
        p2 = np.concatenate([rng.uniform(-1, 1, 3), rng.normal(size=4)]); p2[3:] /= np.linalg.norm(p2[3:])
        A, B = rng.normal(size=(6, 6)), rng.normal(size=(6, 6))
        c1, c2 = A @ A.T * 1e-4, B @ B.T * 1e-3
        got, ref = mla.compound_pose_with_cov(p1, c1, p2, c2), orc.compound_pose_with_cov(p1, c1, p2, c2)
        np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=1e-14)
        np.testing.assert_allclose(got[1], ref[1], rtol=1e-12, atol=1e-18)


def test_std_sort_mt_equals_std_sort(tmp_path):
    """The voxel filters' default member order is libstdc++'s std::sort order, produced on several host threads by
    m-loam_amd/csrc/std_sort_mt.hpp (libstdc++'s own partition steps, the recursion's independent halves forked). Equal to std::sort element
    for element -- duplicates, patterned inputs, every fork depth -- or the reference's voxel results are not reproduced."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "tests", "host", "std_sort_mt_check.cpp")
    exe = str(tmp_path / "std_sort_mt_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "m-loam_amd", "csrc"), "-o", exe, src], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert " 0 mismatches" in out, out


def test_facade_window_degeneracy_policy(tmp_path, orc):
    """The facade's evalDegenracy(local_param_ids, window normal equations, frame_cnt, state) = Estimator::evalDegenracy (estimator.cpp:1598-1680):
    host code over mlh_eval_degeneracy, so it runs without a GPU. Held against the CPU restatement (which tests/test_oracle_ref_pin.py holds
    against the reference's own lines): flags, projectors, the updated thresholds and d_factor_calib, on calibration and non-calibration frames."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libmloam_hip.so")):
        pytest.skip("libmloam_hip.so not built")
    exe = str(tmp_path / "window_degeneracy_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "m-loam_amd", "host"), "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "window_degeneracy_check.cpp"), "-L", lib, "-lmloam_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
                    "-L/opt/rocm/lib"], check=True)
    rng = np.random.default_rng(5)
    W, L_ = 3, 3
    n_pose, D = W + 1, 6 * (W + 1 + L_)
    J = np.zeros((900, D))
    scale = [3.0, 0.9, 0.05]
    for r in range(900):
        f, e = 1 + r % W, r % L_
        for b, sc in ((0, 1.0), (f, 1.0), (n_pose + e, scale[e])):
            j = rng.normal(0, 1, 6) * sc
            if b == 2:
                j[[1, 4]] *= 1e-3
            J[r, 6 * b:6 * b + 6] = j
    H = J.T @ J
    thr = np.array([100.0] * n_pose + [0.0, 5.0, 5.0])
    for frame_cnt, est_ext in ((20, 1), (23, 1), (20, 0)):
        H.tofile(tmp_path / "JtJ.f64")
        thr.tofile(tmp_path / "eig_thre.f64")
        subprocess.run([exe, str(tmp_path), str(D), str(W), str(L_), str(est_ext), str(frame_cnt), "10", "70.0"], check=True)
        out = np.fromfile(tmp_path / "out.f64")
        nb = D // 6
        rec = out[:nb * 38].reshape(nb, 38)
        want = orc.window_eval_degeneracy(H, n_pose, thr, bool(est_ext), frame_cnt, 10, 70.0)
        np.testing.assert_array_equal(rec[:, 0] != 0, want["is_degenerate"])
        np.testing.assert_allclose(rec[:, 1], want["eig_thre"], rtol=1e-9)
        np.testing.assert_allclose(rec[:, 2:].reshape(nb, 6, 6), want["V_update"], atol=1e-8)
        if est_ext:
            np.testing.assert_allclose(out[nb * 38:], want["d_factor_calib"], rtol=1e-9)
        assert want["is_degenerate"][2]



def test_facade_front_end_lanes(tmp_path):
    """FrontEndLanes (the facade's own workers behind estimator.cpp:248-263 for a caller without OpenMP), without a GPU: stand-in segmenter / extractor types record
    who ran what where -- per LiDAR calTimestamp -> segmentCloud -> extractCloud once and in order, passes when there are more LiDARs than lanes, the same worker threads
    frame after frame and never the caller's, two lanes really overlap, a job's exception is rethrown at wait() and the lane goes on, one lane's jobs run in order."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libmloam_hip.so")):
        pytest.skip("libmloam_hip.so not built")
    exe = str(tmp_path / "front_end_lanes_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "m-loam_amd", "host"), "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "front_end_lanes_check.cpp"), "-L", lib, "-lmloam_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
                    "-L/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout


def test_facade_keyframe_policy_and_pose_chain(tmp_path, orc):
    """The host side of the pipelined mapper loop, without a GPU: poseMul / poseInverse compose the next frame's start pose as the reference's own lines do
    (Pose::operator*, Pose::inverse, transformUpdate, transformAssociateToMap through oracle/_ref -- bit for bit, as the device's chain is), and KeyframePolicy makes
    saveKeyframe's decisions (lidar_mapper_keyframe.cpp:641-657) and extractSurroundingKeyFrames' selection (cpp:266-272: within the radius, nearest first)."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libmloam_hip.so")):
        pytest.skip("libmloam_hip.so not built")
    exe = str(tmp_path / "keyframe_policy_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "m-loam_amd", "host"), "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "keyframe_policy_check.cpp"), "-L", lib, "-lmloam_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
                    "-L/opt/rocm/lib"], check=True)
    rng = np.random.default_rng(12)
    n = 40
    # a trajectory: 0.3-0.5 m and up to 0.8 degrees per frame, one sharper turn
    poses = np.zeros((n, 7))
    t = np.zeros(3)
    yaw = 0.0
    for i in range(n):
        yaw += np.radians(rng.uniform(-0.8, 0.8) + (3.0 if i == 17 else 0.0))
        t = t + np.array([np.cos(yaw), np.sin(yaw), 0.02 * rng.normal()]) * rng.uniform(0.3, 0.5)
        q = np.array([0.01 * rng.normal(), 0.01 * rng.normal(), np.sin(yaw / 2), np.cos(yaw / 2)])
        poses[i] = np.concatenate([t, q / np.linalg.norm(q)])
    poses.tofile(tmp_path / "poses.f64")
    dist_kf, ori_kf, radius = 1.0, 1.0, 4.0
    subprocess.run([exe, str(tmp_path), str(n), str(dist_kf), str(ori_kf), str(radius)], check=True)
    out = np.fromfile(tmp_path / "out.f64")
    chain = out[:7 * (n - 2)].reshape(n - 2, 7)
    dec = out[7 * (n - 2):].reshape(n, 4)
    if orc.ref_lib() is not None:
        for i in range(n - 2):
            assert np.array_equal(chain[i], orc.ref_pose_chain(poses[i], poses[i + 1], poses[i + 2])), i
    for i in range(n - 2):
        assert np.array_equal(chain[i], orc.pose_chain(poses[i], poses[i + 1], poses[i + 2])), i
    # saveKeyframe / radius search restated: f32 positions (PointI), angularDistance = 2 acos(|q1 . q2|)
    kf_pos, kf_ids, prev_p, prev_q = [], [], None, None
    n_saved = 0
    for i in range(n):
        p32 = poses[i, :3].astype(np.float32)
        if prev_p is None:
            save = True
        else:
            d = float(np.sqrt(np.float64(np.sum((p32 - prev_p) ** 2, dtype=np.float32))))
            ang = np.degrees(2.0 * np.arccos(min(1.0, abs(float(poses[i, 3:] @ prev_q)))))
            save = d > dist_kf or ang > ori_kf
        assert bool(dec[i, 0]) == save, i
        if save:
            assert int(dec[i, 1]) == n_saved
            kf_pos.append(p32); prev_p, prev_q = p32, poses[i, 3:].copy(); n_saved += 1
        else:
            assert int(dec[i, 1]) == -1
        kp = np.array(kf_pos)
        d2 = np.sum((kp - p32) ** 2, axis=1, dtype=np.float32)
        ids = [int(j) for j in np.lexsort((np.arange(len(kp)), d2)) if d2[j] <= np.float32(radius) ** 2]
        h = 0.0
        for j in ids:
            h = h * 31.0 + (j + 1)
        assert int(dec[i, 2]) == len(ids) and dec[i, 3] == h, (i, ids)
    assert 5 < n_saved < n - 5                      # both decisions occur
    # ... and the decisions are those of saveKeyframe's own lines (oracle/_ref: lidar_mapper_keyframe.cpp:641-683 over the shim)
    if orc.ref_lib() is not None:
        assert np.array_equal(orc.ref_save_keyframes(poses, dist_kf, ori_kf), dec[:, 0].astype(np.uint8))


def test_facade_cal_timestamp_is_the_references(tmp_path, orc, synth):
    """The facade's FeatureExtract::calTimestamp (host code: the sweep's two-phase azimuth unwrapping, float variables against double constants) against
    findStartEndAngle + calTimestamp compiled from the reference's own lines (oracle/_ref, feature_extract.cpp:54-114): the same float for every point of raw scans in
    firing order, sweeps starting anywhere, clockwise and counter-clockwise."""
    import subprocess
    if orc.ref_lib() is None:
        pytest.skip("no reference build")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libmloam_hip.so")):
        pytest.skip("libmloam_hip.so not built")
    exe = str(tmp_path / "cal_timestamp_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-I", os.path.join(ROOT, "m-loam_amd", "host"), "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "cal_timestamp_check.cpp"), "-L", lib, "-lmloam_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
                    "-L/opt/rocm/lib"], check=True)
    rng = np.random.default_rng(4)
    scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    s = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[0], 16, seed=3)
    n_checked = 0
    for start in (0.0, 1.0, 3.0, -2.5):
        for direction in (1.0, -1.0):
            pts = s.points[:, :3].copy()
            az = np.mod(direction * (np.arctan2(pts[:, 1], pts[:, 0]) - start), 2 * np.pi)       # firing order: azimuth step by azimuth step from `start`, either way round
            pts = np.ascontiguousarray(pts[np.argsort(az, kind="stable")], np.float32)
            pts.tofile(tmp_path / "cloud.f32")
            subprocess.run([exe, str(tmp_path), "0.1"], check=True)
            got = np.fromfile(tmp_path / "rel_time.f32", np.float32)
            want = orc.ref_cal_timestamp(pts, 0.1)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (start, direction, int(np.sum(got != want)))
            n_checked += len(pts)
    assert n_checked > 100000


def test_facade_map_factors_are_the_references(tmp_path, orc):
    """The facade's per-factor host classes (LidarMapPlaneNormFactor / LidarMapEdgeFactor: what a caller that keeps the reference's AddResidualBlock loop constructs,
    lidar_mapper_keyframe.cpp:537-571) against the reference's own lines (oracle/_ref, lidar_map_factor.hpp:28-174): residual and 1 x 7 Jacobian on 2 000 random
    factors, the weight rule of the constructors included."""
    import subprocess
    if orc.ref_lib() is None:
        pytest.skip("no reference build")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libmloam_hip.so")):
        pytest.skip("libmloam_hip.so not built")
    exe = str(tmp_path / "map_factor_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-I", os.path.join(ROOT, "m-loam_amd", "host"), "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "map_factor_check.cpp"), "-L", lib, "-lmloam_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
                    "-L/opt/rocm/lib"], check=True)
    rng = np.random.default_rng(31)
    n = 2000
    rows = np.zeros((n, 26))
    for i in range(n):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pose = np.concatenate([rng.uniform(-30, 30, 3), q])
        sd = rng.uniform(0.01, 0.6, 3)
        cov = np.diag(sd ** 2); cov[0, 1] = cov[1, 0] = 0.1 * sd[0] * sd[1]
        kind = i % 2
        if kind == 0:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            coeff = np.concatenate([nrm, [rng.uniform(-5, 5)], [0, 0]])
        else:
            c = rng.uniform(-40, 40, 3); v = rng.normal(size=3); v /= np.linalg.norm(v)
            coeff = np.concatenate([c + 0.1 * v, c - 0.1 * v])
        rows[i] = np.concatenate([[kind], rng.uniform(-40, 40, 3), coeff, cov.ravel(), pose])
    rows.tofile(tmp_path / "factors.f64")
    subprocess.run([exe, str(tmp_path)], check=True)
    out = np.fromfile(tmp_path / "factors_out.f64").reshape(n, 8)
    for i in range(n):
        kind = "s" if rows[i, 0] == 0 else "c"
        r_ref, J_ref = orc.ref_map_factor(kind, rows[i, 1:4], rows[i, 4:10][: 4 if kind == "s" else 6], rows[i, 10:19].reshape(3, 3), rows[i, 19:26])
        assert abs(out[i, 0] - r_ref) <= 1e-12 * max(1.0, abs(r_ref)), (i, kind)
        np.testing.assert_allclose(out[i, 1:], J_ref, rtol=1e-11, atol=1e-11)


def test_facade_odometry_factors_are_the_references(tmp_path, orc):
    """The facade's per-factor host classes of the odometry window and the calibration (LidarPureOdom{PlaneNorm,Edge}Factor over [pivot, frame, extrinsic],
    LidarOnlineCalib{PlaneNorm,Edge}Factor over the extrinsic: what a caller that keeps Estimator::optimizeMap's AddResidualBlock loops constructs, estimator.cpp:733-813)
    against the reference's own lines (oracle/_ref): residuals and every Jacobian row on 1 500 random factors -- the two columns of the reference that are not exact
    derivatives included."""
    import subprocess
    if orc.ref_lib() is None:
        pytest.skip("no reference build")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libmloam_hip.so")):
        pytest.skip("libmloam_hip.so not built")
    exe = str(tmp_path / "odom_factor_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-I", os.path.join(ROOT, "m-loam_amd", "host"), "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "odom_factor_check.cpp"), "-L", lib, "-lmloam_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
                    "-L/opt/rocm/lib"], check=True)
    rng = np.random.default_rng(37)

    def rand_pose(scale):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        return np.concatenate([rng.uniform(-scale, scale, 3), q])
    n = 1500
    rows = np.zeros((n, 32))
    for i in range(n):
        kind = i % 2
        if kind == 0:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            coeff = np.concatenate([nrm, [rng.uniform(-5, 5)], [0, 0]])
        else:
            c = rng.uniform(-40, 40, 3); v = rng.normal(size=3); v /= np.linalg.norm(v)
            coeff = np.concatenate([c + 0.1 * v, c - 0.1 * v])
        rows[i] = np.concatenate([[kind], rng.uniform(-40, 40, 3), coeff, [rng.uniform(0.3, 1.0)], rand_pose(20.0), rand_pose(20.0), rand_pose(1.0)])
    rows.tofile(tmp_path / "ofactors.f64")
    subprocess.run([exe, str(tmp_path)], check=True)
    out = np.fromfile(tmp_path / "ofactors_out.f64").reshape(n, 30)
    for i in range(n):
        kind = "s" if rows[i, 0] == 0 else "c"
        coeff = rows[i, 4:10][: 4 if kind == "s" else 6]
        r_ref, J_ref = orc.ref_pure_odom(kind, rows[i, 1:4], coeff, rows[i, 10], rows[i, 11:18], rows[i, 18:25], rows[i, 25:32])
        assert abs(out[i, 0] - r_ref) <= 1e-11 * max(1.0, abs(r_ref)), (i, kind)
        np.testing.assert_allclose(out[i, 1:22].reshape(3, 7), J_ref, rtol=1e-10, atol=1e-10)
        rc, Jc = orc.ref_online_calib(kind, rows[i, 1:4], coeff, rows[i, 10], rows[i, 25:32])
        assert abs(out[i, 22] - rc) <= 1e-12 * max(1.0, abs(rc)), (i, kind)
        np.testing.assert_allclose(out[i, 23:30], Jc, rtol=1e-11, atol=1e-11)


def test_every_entry_point_refuses_a_null_context(mla):
    """Every C-ABI function that takes a context must hand back an error (not crash, not touch the GPU) when the context is null and every other argument is
    zero / null -- the first thing a binding gets wrong. Run in a child process so that a crash names its function instead of taking the test run down."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(ROOT, "include", "mloam_hip.h")).read()
    names = sorted(set(re.findall(r"\b(mlh_\w+)\s*\(\s*(?:const\s+)?mlh_ctx\s*\*", hdr)))
    assert len(names) > 60
    code = r'''
import ctypes as C, importlib, sys
sys.path.insert(0, %r)
mla = importlib.import_module("m-loam_amd")
lib = mla.load_library()
names = %r
bad = []
for nm in names:
    fn = getattr(lib, nm)
    at = fn.argtypes
    assert at is not None, nm + ": no argtypes declared in the Python binding"
    args = []
    for t in at:
        if t in (C.c_float, C.c_double):
            args.append(0.0)
        elif t in (C.c_int, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_longlong, C.c_ulonglong, C.c_size_t):
            args.append(0)
        else:
            args.append(None)
    print("calling", nm, flush=True)
    rc = fn(*args)
    if fn.restype in (None, C.c_char_p, C.c_void_p):
        continue
    if rc == 0:
        bad.append(nm)
print("ACCEPTED A NULL CONTEXT:", bad)
sys.exit(1 if bad else 0)
''' % (ROOT, names)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-600:], r.stderr[-1500:])
