"""Independent pins for the restated LIBRARY arithmetic of the CPU oracle (Eigen / FLANN / Ceres are absent here and the reference has no golden
vectors for this path; the reference's own logic is pinned separately, tests/test_oracle_ref_pin.py).
 - analytic vs numeric Jacobians: the reference's own check() recipe (lidar_map_factor.hpp:98-118, 204-227), as assertions
 - independent cross-checks: numpy.linalg (eigh, lstsq), scipy.spatial.cKDTree, scipy.optimize.least_squares
 - the 4-point covariance-voxel example of mloam_test/src/test_pointiwithcov.cpp with hand-derived expected values"""
import numpy as np
import pytest
from scipy.spatial import cKDTree


def _rand_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.uniform(-3, 3, 3), q])


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def _perturb(pose, k, eps):
    """the reference's check(): translation += eps*e_k, or q <- q * deltaQ(eps*e_k) (NOT re-normalised)"""
    p = pose.copy()
    if k < 3:
        p[k] += eps
    else:
        d = np.zeros(3)
        d[k - 3] = eps
        p[3:7] = _quat_mul(pose[3:7], np.array([d[0] / 2, d[1] / 2, d[2] / 2, 1.0]))
    return p


@pytest.mark.parametrize("kind", ["s", "c"])
def test_factor_jacobian_matches_finite_differences(orc, kind):
    rng = np.random.default_rng(1)
    for trial in range(20):
        pose = _rand_pose(rng)
        point = rng.uniform(-20, 20, 3)
        if kind == "s":
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            coeff = np.concatenate([n, [rng.uniform(-5, 5)]])
        else:
            c = rng.uniform(-20, 20, 3)
            v = rng.normal(size=3)
            v /= np.linalg.norm(v)
            coeff = np.concatenate([c + 0.1 * v, c - 0.1 * v])
        trace = [0.0075, 0.5, 30.0][trial % 3]
        r, J = orc.factor_eval(kind, point, coeff, trace, pose)
        assert J[6] == 0.0
        eps = 1e-6
        num = np.array([(orc.factor_eval(kind, point, coeff, trace, _perturb(pose, k, eps))[0] - r) / eps for k in range(6)])
        np.testing.assert_allclose(J[:6], num, rtol=2e-4, atol=2e-4)


def test_weight_from_covariance_trace(orc):
    # sqrt(1/tr) >= 3 -> 1 ; else /3  (lidar_map_factor.hpp:35,41)
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
    n = np.array([0, 0, 1.0, -1.0])
    p = np.array([0.3, 0.2, 3.0])
    r1, _ = orc.factor_eval("s", p, n, 0.0075, pose)       # sqrt(133.3)=11.5 >= 3 -> w = 1
    assert abs(r1 - 2.0) < 1e-12
    r2, _ = orc.factor_eval("s", p, n, 1.0, pose)          # sqrt(1) = 1 < 3 -> w = 1/3
    assert abs(r2 - 2.0 / 3.0) < 1e-12
    r3, _ = orc.factor_eval("s", p, n, 1.0 / 9.0, pose)    # sqrt(9) = 3 >= 3 -> w = 1
    assert abs(r3 - 2.0) < 1e-12
    r4, _ = orc.factor_eval("s", p, n, 0.0, pose)          # with_ua off + zero cov: sqrt(inf) -> w = 1
    assert abs(r4 - 2.0) < 1e-12


def test_huber_matches_ceres_definition(orc):
    a = 0.1
    rho = orc.huber(a, 0.005)      # s <= a^2
    assert tuple(rho) == (0.005, 1.0, 0.0)
    s = 0.04
    rho = orc.huber(a, s)
    assert abs(rho[0] - (2 * a * 0.2 - a * a)) < 1e-15 and abs(rho[1] - a / 0.2) < 1e-15 and abs(rho[2] + rho[1] / (2 * s)) < 1e-15


def test_pose_plus_is_right_multiplicative_and_normalised(orc):
    rng = np.random.default_rng(2)
    x = _rand_pose(rng)
    d = rng.normal(size=6) * 0.05
    y = orc.pose_plus(x, d)
    np.testing.assert_allclose(y[:3], x[:3] + d[:3], atol=1e-15)
    q = _quat_mul(x[3:7], np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0]))
    np.testing.assert_allclose(y[3:7], q / np.linalg.norm(q), atol=1e-15)
    assert abs(np.linalg.norm(y[3:7]) - 1) < 1e-15
    # degeneracy projection: only the component along kept eigen-directions is applied
    V = np.eye(6)
    V[0, 0] = 0.0
    y2 = orc.pose_plus(x, d, V)
    assert y2[0] == x[0] and abs(y2[1] - (x[1] + d[1])) < 1e-15


def test_eig3f_against_numpy(orc):
    rng = np.random.default_rng(3)
    for _ in range(200):
        pts = rng.normal(size=(5, 3)) * rng.uniform(0.01, 1.0, 3)
        c = pts - pts.mean(0)
        A = (c.T @ c).astype(np.float32)
        val, vec, rc = orc.eig3f(A)
        assert rc == 0
        w, v = np.linalg.eigh(A.astype(np.float64))
        np.testing.assert_allclose(val, w, rtol=2e-4, atol=2e-6 * max(1.0, abs(w).max()))
        assert val[0] <= val[1] <= val[2]
        if w[2] > 3 * w[1] and w[2] - w[1] > 1e-3 * w[2]:
            assert abs(abs(float(vec[:, 2] @ v[:, 2])) - 1.0) < 1e-3
        np.testing.assert_allclose(vec.T @ vec, np.eye(3), atol=1e-5)


def test_qr_plane_fit_against_numpy(orc):
    rng = np.random.default_rng(4)
    for _ in range(200):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        d = rng.uniform(1, 30)
        # 5 points near the plane n.x + d = 0
        base = rng.normal(size=(5, 3)) * 0.4
        pts = base - np.outer(base @ n + d, n) + rng.normal(size=(5, 3)) * 0.01
        A = pts.astype(np.float32)
        x = orc.qr_solve(A, -np.ones(5, np.float32))
        ref = np.linalg.lstsq(A.astype(np.float64), -np.ones(5), rcond=None)[0]
        np.testing.assert_allclose(x, ref, rtol=5e-3, atol=5e-4)


def test_kdtree_knn_against_scipy(orc):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-30, 30, (20000, 3)).astype(np.float32)
    q = rng.uniform(-30, 30, (2000, 3)).astype(np.float32)
    m = orc.Map(pts)
    idx, d2 = m.knn(q, 5)
    tree = cKDTree(pts.astype(np.float64))
    dd, ii = tree.query(q.astype(np.float64), k=5)
    assert np.array_equal(np.sort(idx, 1), np.sort(ii, 1))
    np.testing.assert_allclose(d2, dd ** 2, rtol=1e-5)
    assert np.all(np.diff(d2, axis=1) >= 0)
    # k = 10 (N_NEIGH of buildCalibMap)
    idx10, _ = m.knn(q[:200], 10)
    _, ii10 = tree.query(q[:200].astype(np.float64), k=10)
    assert np.array_equal(np.sort(idx10, 1), np.sort(ii10, 1))


def test_voxel_grid_cov_known_answer(orc):
    """mloam_test/src/test_pointiwithcov.cpp:23-38: four points, leaf 3 m, trace threshold 2.
    Hand-derived from voxel_grid_covariance_mloam_impl.hpp:296-333: w = thr - tr = (2,2,2,1), W = 7,
    mu = (2*p1 + 2*p2 + 2*p3 + p4)/7 = (3/7, 3/7, 0); cov = sum w^2 cov_i / W^2 -> cxx = 1/49; trace = 1/49."""
    pts = np.zeros((4, 11), np.float32)
    pts[1, 0] = 1
    pts[2, 1] = 1
    pts[3, 0] = pts[3, 1] = 1
    pts[3, 4] = 1      # cxx
    pts[3, 10] = 1     # trace
    out = orc.voxel_grid_cov(pts, 3.0, 2.0)
    assert out.shape == (1, 11)
    exp = np.zeros(11, np.float32)
    exp[0] = exp[1] = np.float32(3.0) / np.float32(7.0)
    exp[4] = exp[10] = np.float32(1.0) / np.float32(49.0)
    np.testing.assert_allclose(out[0], exp, rtol=1e-6, atol=1e-8)
    # a point with trace >= threshold is dropped from the voxel
    pts[3, 4] = 2.5
    out2 = orc.voxel_grid_cov(pts, 3.0, 2.0)
    np.testing.assert_allclose(out2[0, :3], [1 / 3, 1 / 3, 0], rtol=1e-6)
    assert out2[0, 4] == 0


def test_point_uncertainty_against_numpy(orc):
    rng = np.random.default_rng(6)
    pose = _rand_pose(rng)
    cov_pose = np.diag([0.0025, 0.0025, 0.0025, 0.00030461, 0.00030461, 0.00030461])
    cov_meas = np.diag([0.0025] * 3)
    pts = rng.uniform(-40, 40, (50, 3)).astype(np.float32)
    got = orc.eval_point_uncertainty(pts, pose, cov_pose, cov_meas)
    x, y, z, w = pose[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    for i, p in enumerate(pts.astype(np.float64)):
        tp = R @ p + pose[:3]
        S = np.array([[0, -tp[2], tp[1]], [tp[2], 0, -tp[0]], [-tp[1], tp[0], 0]])
        G = np.hstack([np.eye(3), -S, R])
        C = np.zeros((9, 9))
        C[:6, :6] = cov_pose
        C[6:, 6:] = cov_meas
        np.testing.assert_allclose(got[i], G @ C @ G.T, rtol=1e-10, atol=1e-12)


def test_degeneracy_projection(orc):
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
    lam = np.array([5.0, 60.0, 300.0, 4000.0, 5e4, 6e5])
    H = Q @ np.diag(lam) @ Q.T
    d = orc.eval_degeneracy(H, 100.0)
    assert d["is_degenerate"]
    np.testing.assert_allclose(d["eigval"], lam, rtol=1e-9)
    P = Q[:, 2:] @ Q[:, 2:].T      # projector on the well-constrained subspace
    np.testing.assert_allclose(d["V_update"], P, atol=1e-9)
    d2 = orc.eval_degeneracy(H, 1.0)
    assert not d2["is_degenerate"]
    np.testing.assert_allclose(d2["V_update"], np.eye(6), atol=0)


@pytest.mark.parametrize("kind", ["s", "c"])
def test_pure_odom_factor_jacobians(orc, kind):
    """LidarPureOdom{PlaneNorm,Edge}Factor (lidar_pure_odom_factor.hpp:38-102, 209-282): analytic vs numeric, per block, with the
    reference's perturbation (t += eps e_k ; q <- q * deltaQ(eps e_k)). The edge factor's extrinsic-rotation column is restated
    as the reference writes it (Rext [p]x + [t_ext]x), which is NOT the derivative of the residual: that column is only
    checked against the formula, not against finite differences."""
    rng = np.random.default_rng(8)
    for trial in range(15):
        pivot, pose_i, ext = _rand_pose(rng), _rand_pose(rng), _rand_pose(rng)
        point = rng.uniform(-10, 10, 3)
        if kind == "s":
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            coeff = np.concatenate([n, [rng.uniform(-5, 5)]])
        else:
            c = rng.uniform(-10, 10, 3)
            v = rng.normal(size=3)
            v /= np.linalg.norm(v)
            coeff = np.concatenate([c + 0.1 * v, c - 0.1 * v])
        r, J = orc.pure_odom_eval(kind, point, coeff, pivot, pose_i, ext)
        assert np.all(J[:, 6] == 0)
        eps = 1e-6
        blocks = [pivot, pose_i, ext]
        for b in range(3):
            for k in range(6):
                pb = [x.copy() for x in blocks]
                pb[b] = _perturb(blocks[b], k, eps)
                num = (orc.pure_odom_eval(kind, point, coeff, *pb)[0] - r) / eps
                if b == 2 and k >= 3:
                    continue   # extrinsic rotation: plane uses [Rext p]x, edge uses Rext [p]x + [t_ext]x -- see docstring
                if b == 0 and k >= 3 and kind == "s":
                    continue   # plane factor, pivot rotation: the reference writes Rp^T [v]x (a LEFT perturbation), not [Rp^T v]x
                assert abs(J[b, k] - num) < 5e-4 * max(1.0, abs(num)), (b, k, J[b, k], num)
        # with identity pivot and identity extrinsic the frame-i block equals the map factor's Jacobian
        ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
        r2, J2 = orc.pure_odom_eval(kind, point, coeff, ident, pose_i, ident)
        r3, J3 = orc.factor_eval(kind, point, coeff, 0.0, pose_i)
        assert abs(r2 - r3) < 1e-12
        np.testing.assert_allclose(J2[1], J3, rtol=1e-10, atol=1e-12)


def _so3_exp(phi):
    th = np.linalg.norm(phi, axis=1)[:, None, None]
    K = np.zeros((len(phi), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -phi[:, 2], phi[:, 1], phi[:, 2], -phi[:, 0], -phi[:, 1], phi[:, 0]
    th = np.maximum(th, 1e-12)
    I = np.eye(3)[None]
    R = I + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
    J = I + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    return R, J


def test_compound_pose_with_cov_matches_monte_carlo(orc, synth):
    """compoundPoseWithCov (associate_uct.hpp:90-147) is the 4th-order SE(3) pose-compounding formula with perturbations
    xi = [rho, phi] applied on the left, T = exp(xi^) T_mean. Pin the restatement against sampling: the empirical covariance
    of T1 T2 must agree with the formula (to sampling error), and clearly better than the first-order term alone would."""
    rng = np.random.default_rng(5)
    p1 = np.array([4.0, -2.0, 1.0, 0.0, 0.0, np.sin(0.35), np.cos(0.35)])
    p2 = np.array([0.6, 0.3, -0.2, np.sin(0.1), 0.0, 0.0, np.cos(0.1)])
    A = rng.normal(size=(6, 6)); c1 = A @ A.T * 2e-4
    B = rng.normal(size=(6, 6)); c2 = B @ B.T * 3e-4
    pose_cp, cov_cp = orc.compound_pose_with_cov(p1, c1, p2, c2)

    def mats(p):
        T = np.eye(4); T[:3, :3] = synth.quat_to_rot(p[3:]); T[:3, 3] = p[:3]
        return T

    T1, T2, Tc = mats(p1), mats(p2), mats(pose_cp)
    np.testing.assert_allclose(Tc, T1 @ T2, atol=1e-12)
    n = 400000

    def sample(cov):
        xi = rng.multivariate_normal(np.zeros(6), cov, size=n)
        R, J = _so3_exp(xi[:, 3:])
        T = np.tile(np.eye(4), (n, 1, 1))
        T[:, :3, :3] = R
        T[:, :3, 3] = (J @ xi[:, :3, None])[:, :, 0]
        return T

    S = (sample(c1) @ T1) @ (sample(c2) @ T2) @ np.linalg.inv(Tc)
    # log of the residual transform (small): phi from the skew part, rho = J^-1 t
    Rm = S[:, :3, :3]
    cos = np.clip((np.trace(Rm, axis1=1, axis2=2) - 1) / 2, -1, 1)
    th = np.arccos(cos)
    w = np.stack([Rm[:, 2, 1] - Rm[:, 1, 2], Rm[:, 0, 2] - Rm[:, 2, 0], Rm[:, 1, 0] - Rm[:, 0, 1]], axis=1)
    phi = w * (th / (2 * np.sin(np.maximum(th, 1e-12))))[:, None]
    _, J = _so3_exp(phi)
    rho = np.linalg.solve(J, S[:, :3, 3][:, :, None])[:, :, 0]
    xi = np.concatenate([rho, phi], axis=1)
    emp = xi.T @ xi / n
    scale = np.sqrt(np.outer(np.diag(cov_cp), np.diag(cov_cp)))
    assert np.max(np.abs(emp - cov_cp) / scale) < 0.02


def test_transform_cloud_feature_is_the_rigid_transform(orc, synth):
    """transformCloudFeature (visualization.cpp:39-51): float32 R p + t of every point, intensity replaced by the LiDAR index. The
    float32 restatement stays within a few ulp of the float64 rigid transform and is exact for the identity."""
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-80, 80, (500, 3)), rng.uniform(0, 63, (500, 1))], axis=1).astype(np.float32)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    ext = np.concatenate([[0.1, -0.5, 0.02], q])
    out = orc.transform_cloud_feature(pts, ext, 1)
    ref = pts[:, :3].astype(np.float64) @ synth.quat_to_rot(q).T + ext[:3]
    assert out.dtype == np.float32 and out.shape == pts.shape
    np.testing.assert_allclose(out[:, :3], ref, rtol=0, atol=3e-5)           # |p| <= 140 m: 1 ulp = 1.5e-5
    assert np.all(out[:, 3] == 1.0)
    ident = orc.transform_cloud_feature(pts, np.array([0, 0, 0, 0, 0, 0, 1.0]), 0)
    np.testing.assert_array_equal(ident[:, :3], pts[:, :3])
    assert not ident[:, 3].any()


def test_transform_to_end_closed_forms(orc):
    """TransformToEnd (utility.h:79-100): identity pose -> unchanged; no distortion -> T^-1 T p = p; pure translation -> p + (s - 1) t."""
    rng = np.random.default_rng(9)
    pts = np.concatenate([rng.uniform(-50, 50, (300, 3)), (rng.integers(0, 16, (300, 1)) + rng.uniform(0, 0.0999, (300, 1)))], axis=1).astype(np.float32)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    np.testing.assert_array_equal(orc.transform_to_end(pts, ident, True), pts)
    q = np.array([0.02, -0.01, 0.05, 1.0]); q /= np.linalg.norm(q)
    pose = np.concatenate([[0.4, -0.2, 0.05], q])
    np.testing.assert_allclose(orc.transform_to_end(pts, pose, False)[:, :3], pts[:, :3], rtol=0, atol=2e-5)
    tr = np.array([0.4, -0.2, 0.05, 0, 0, 0, 1.0])
    s = ((pts[:, 3] - np.floor(pts[:, 3])) / np.float32(0.1)).astype(np.float64)
    np.testing.assert_allclose(orc.transform_to_end(pts, tr, True)[:, :3], pts[:, :3] + (s[:, None] - 1.0) * tr[:3], rtol=0, atol=2e-5)
