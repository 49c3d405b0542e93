"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/libmloam_oracle.so).

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` import this module, and only as
the checker / reported CPU baseline. Pin status: the reference's own source lines pin extractCloud, the map matches, the factor
classes, Plus and ImageSegmenter (oracle/ref/, ref_* functions below); the Eigen / PCL-FLANN / Ceres arithmetic is restated --
PARITY UNPINNED for that part, see oracle/linalg.hpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmloam_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

GF_METHODS = {"wo_gf": 0, "rnd": 1, "fps": 2, "gd_fix": 3, "gd_float": 4}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp")) or f == "Makefile"]
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_map_create.restype = C.c_void_p
        _lib.orc_map_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _lib.orc_map_destroy.argtypes = [C.c_void_p]
        _lib.orc_map_rebuild_seconds.restype = C.c_double
        _lib.orc_map_rebuild_seconds.argtypes = [C.c_void_p]
        _lib.orc_logdet.restype = C.c_double
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------- oracle/_ref: the reference's own source lines (oracle/ref/)
_ref = None


def ref_lib():
    """libmloam_ref.so = FeatureExtract::extractCloud, LidarMap{PlaneNorm,Edge}Factor compiled from the line ranges of the files under
    /root/reference (oracle/ref/build_ref.py). Returns None when neither the reference tree nor a prebuilt library is available."""
    global _ref
    if _ref is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("mloam_build_ref", os.path.join(_HERE, "ref", "build_ref.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        path = mod.build()
        if not path or not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
    return _ref


def ref_extract(points, scan_start, scan_end):
    """the reference's extractCloud on a ring-major cloud -> its four feature clouds as (n, 4) float arrays"""
    L = ref_lib()
    pts = np.ascontiguousarray(points, np.float32)
    n = pts.shape[0]
    ss = np.ascontiguousarray(scan_start, np.int32); se = np.ascontiguousarray(scan_end, np.int32)
    outs = [np.zeros((max(n, 1), 4), np.float32) for _ in range(4)]
    ptrs = (C.c_void_p * 4)(*[o.ctypes.data_as(C.c_void_p) for o in outs])
    cnt = (C.c_int * 4)()
    L.ref_extract_cloud(_ptr(pts), n, _ptr(ss), _ptr(se), len(ss), ptrs, cnt)
    names = ["sharp", "less_sharp", "flat", "less_flat_ds"]
    return {nm: outs[i][:cnt[i]].copy() for i, nm in enumerate(names)}


def ref_map_factor(kind: str, point, coeff, cov3x3, pose7, want_jacobian=True):
    """LidarMapPlaneNormFactor ('s') / LidarMapEdgeFactor ('c') constructed with (point, coeff, cov_matrix) and evaluated, from the reference's lines"""
    L = ref_lib()
    p = np.ascontiguousarray(point, np.float64); c = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    cov = np.ascontiguousarray(cov3x3, np.float64).reshape(9); pose = np.ascontiguousarray(pose7, np.float64)
    r = np.zeros(1); J = np.zeros(7)
    L.ref_map_factor_evaluate(C.c_char(kind.encode()), _ptr(p), _ptr(c), _ptr(cov), _ptr(pose), _ptr(r), _ptr(J) if want_jacobian else None)
    return r[0], (J if want_jacobian else None)


def mapper_params(min_match_sq_dis=1.0, min_plane_dis=0.2, huber_delta=0.1, map_eig_thre=100.0, with_ua=False,
                  cov_measurement_trace=0.0075, max_outer=2, max_lm_iterations=30, gf_method="wo_gf", gf_ratio=1.0, seed=0,
                  n_neigh=5, check_fov=False, freeze_when_degenerate=False):
    return np.array([min_match_sq_dis, min_plane_dis, huber_delta, map_eig_thre, float(with_ua), cov_measurement_trace,
                     max_outer, max_lm_iterations, GF_METHODS[gf_method], gf_ratio, seed, n_neigh, float(check_fov),
                     float(freeze_when_degenerate)], dtype=np.float64)


def ref_segment_cloud(points4, prm=None):
    """ImageSegmenter::segmentCloud compiled from the reference's lines (its UB included, as g++ -O2 renders it)"""
    L = ref_lib()
    prm = seg_params() if prm is None else np.ascontiguousarray(prm, np.float64)
    p = np.ascontiguousarray(points4, np.float32)
    n = len(p); vs = int(prm[0])
    out = np.zeros((max(n, 1), 4), np.float32); outl = np.zeros((max(n, 1) + 1, 4), np.float32)
    ss = np.zeros(vs, np.int32); se = np.zeros(vs, np.int32)
    no, nl = C.c_int(0), C.c_int(0)
    L.ref_segment_cloud(_ptr(p), n, _ptr(prm), _ptr(out), C.byref(no), _ptr(outl), C.byref(nl), _ptr(ss), _ptr(se))
    return dict(cloud=out[:no.value].copy(), outlier=outl[:nl.value].copy(), scan_start=ss, scan_end=se)


def ref_match(kind: str, map_pts, feats, pose7, n_neigh=5, check_fov=False, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """FeatureExtract::match{Surf,Corner}PointFromMap (feature_extract.hpp:645-883) from the reference's lines, feature by feature"""
    L = ref_lib()
    m4 = np.zeros((len(map_pts), 4), np.float32); m4[:, :3] = np.asarray(map_pts)[:, :3]
    f4 = np.zeros((len(feats), 4), np.float32); f4[:, :min(4, np.asarray(feats).shape[1])] = np.asarray(feats)[:, :4]
    pose = np.ascontiguousarray(pose7, np.float64)
    valid = np.zeros(len(f4), np.uint8); coeffs = np.zeros((len(f4), 6))
    L.ref_match_points(C.c_char(kind.encode()), _ptr(m4), len(m4), _ptr(f4), len(f4), _ptr(pose), int(n_neigh), int(bool(check_fov)),
                       C.c_float(min_match_sq_dis), C.c_float(min_plane_dis), _ptr(valid), _ptr(coeffs))
    return valid, coeffs


def ref_pure_odom(kind: str, point, coeff, sqrt_info, pivot, pose_i, ext, want_jacobian=True):
    """LidarPureOdom{PlaneNorm,Edge}Factor from the reference's lines -> residual, J (3, 7)"""
    L = ref_lib()
    p = np.ascontiguousarray(point, np.float64); c = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    a = [np.ascontiguousarray(x, np.float64) for x in (pivot, pose_i, ext)]
    r = np.zeros(1); J = np.zeros((3, 7))
    L.ref_pure_odom_evaluate(C.c_char(kind.encode()), _ptr(p), _ptr(c), C.c_double(sqrt_info), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(r),
                             _ptr(J) if want_jacobian else None)
    return r[0], (J if want_jacobian else None)


def ref_online_calib(kind: str, point, coeff, sqrt_info, ext):
    L = ref_lib()
    p = np.ascontiguousarray(point, np.float64); c = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    e = np.ascontiguousarray(ext, np.float64)
    r = np.zeros(1); J = np.zeros(7)
    L.ref_online_calib_evaluate(C.c_char(kind.encode()), _ptr(p), _ptr(c), C.c_double(sqrt_info), _ptr(e), _ptr(r), _ptr(J))
    return r[0], J


def ref_optimize_map(poses, exts, feat_rows, estimate_extrinsic, frame_cnt=0, n_cumu_feature=10, num_iterations=10, eig_thre=None, lambda_thre_calib=70.0):
    """Estimator::optimizeMap up to its marginalisation section, compiled from the reference's own lines (estimator.cpp:593-866, with vector2Double /
    double2Vector :1538-1576, evalResidual :1578-1595 and evalDegenracy :1598-1680) over the Ceres-shaped shim; MARGINALIZATION_FACTOR = PRIOR_FACTOR = 0.
    poses: (OPT_WINDOW_SIZE + 1, 7) with the pivot first; exts: (n_laser, 7), IDX_REF = 0; feat_rows: (m, 12) [laser, window frame index, kind 0 surf / 1 corner,
    point xyz, coefficients (4 or 6, zero-padded)]. Returns the solved poses / extrinsics, J^T J of the Jacobian evalResidual evaluates (constant blocks: zero
    columns), that evaluation's cost, the number of residual blocks and the solver summary."""
    L = ref_lib()
    P = np.ascontiguousarray(poses, np.float64).reshape(-1, 7); E = np.ascontiguousarray(exts, np.float64).reshape(-1, 7)
    F = np.ascontiguousarray(feat_rows, np.float64).reshape(-1, 12)
    nb, nl = len(P), len(E)
    thr = np.ascontiguousarray(np.full(nb + nl, 10.0) if eig_thre is None else eig_thre, np.float64)
    Po, Eo = np.zeros_like(P), np.zeros_like(E)
    D = 6 * (nb + nl)
    H = np.zeros((D, D)); cost = C.c_double(0); nblk = C.c_int(0); so = np.zeros(5)
    L.ref_optimize_map(nb - 1, nl, int(estimate_extrinsic), int(n_cumu_feature), int(frame_cnt), int(num_iterations), _ptr(P), _ptr(E), _ptr(F), len(F), _ptr(thr),
                       C.c_double(lambda_thre_calib), _ptr(Po), _ptr(Eo), _ptr(H), C.byref(cost), C.byref(nblk), _ptr(so))
    return dict(poses=Po, exts=Eo, H=H, cost=cost.value, n_blocks=nblk.value,
                solve=dict(lm_iterations=int(so[0]), successful_steps=int(so[1]), initial_cost=so[2], final_cost=so[3], termination=int(so[4])))


def ref_pose_plus(x, delta, V_update=None):
    L = ref_lib()
    x = np.ascontiguousarray(x, np.float64); d = np.ascontiguousarray(delta, np.float64)
    V = None if V_update is None else np.ascontiguousarray(V_update, np.float64)
    out = np.zeros(7)
    L.ref_pose_plus(_ptr(x), _ptr(d), _ptr(V), _ptr(out))
    return out


def _as11(a):
    a = np.asarray(a, np.float32)
    out = np.zeros((a.shape[0], 11), np.float32)
    out[:, :min(11, a.shape[1])] = a[:, :11]
    return np.ascontiguousarray(out)


def ref_good_feature_matching(map_points, kind, feats11, pose7, gf_method, gf_ratio, seed, H=None, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """ActiveFeatureSelection::goodFeatureMatching compiled from the reference's own lines (lidar_mapper.h:229-573), its mt19937 re-seeded"""
    L = ref_lib()
    m, f = _as11(map_points), _as11(feats11)
    pose = np.ascontiguousarray(pose7, np.float64)
    Hm = np.ascontiguousarray(np.eye(6) * 1e-6 if H is None else H, np.float64).copy()
    sel = np.zeros(max(len(f), 1), np.int32); nsel = C.c_int(0)
    L.ref_good_feature_matching(C.c_char(kind.encode()), _ptr(m), len(m), _ptr(f), len(f), _ptr(pose), gf_method.encode(), C.c_double(gf_ratio),
                                C.c_uint(int(seed)), C.c_float(min_match_sq_dis), C.c_float(min_plane_dis), _ptr(Hm), _ptr(sel), C.byref(nsel))
    return dict(sel=sel[:nsel.value].copy(), H=Hm)


def ref_cal_timestamp(xyz, scan_period=0.1):
    """FeatureExtract::findStartEndAngle + calTimestamp (feature_extract.cpp:54-114) from the reference's own lines: the relative time of every point of a raw cloud"""
    L = ref_lib()
    p = np.ascontiguousarray(np.asarray(xyz)[:, :3], np.float32)
    out = np.zeros(len(p), np.float32)
    L.ref_cal_timestamp(_ptr(p), len(p), C.c_float(scan_period), _ptr(out))
    return out


def ref_save_keyframes(poses, distance_keyframes=1.0, orientation_keyframes=1.0):
    """saveKeyframe (lidar_mapper_keyframe.cpp:641-683) from the reference's own lines, called once per pose of the sequence from a clean state -> saved flags"""
    L = ref_lib()
    p = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    saved = np.zeros(len(p), np.uint8)
    L.ref_save_keyframes(_ptr(p), len(p), C.c_float(distance_keyframes), C.c_float(orientation_keyframes), _ptr(saved))
    return saved


def ref_match_cloud(kind: str, map_pts, feats, pose7, n_neigh=5, check_fov=True, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """FeatureExtract::matchSurfFromMap / matchCornerFromMap, the WHOLE-CLOUD forms (feature_extract.hpp:378-643: what buildCalibMap calls, estimator.cpp:1136-1150),
    from the reference's own lines -> (valid per feature, coeffs per feature) in the per-point functions' layout"""
    L = ref_lib()
    m4 = np.zeros((len(map_pts), 4), np.float32); m4[:, :3] = np.asarray(map_pts)[:, :3]
    f4 = np.zeros((len(feats), 4), np.float32); f4[:, :min(4, np.asarray(feats).shape[1])] = np.asarray(feats)[:, :4]
    pose = np.ascontiguousarray(pose7, np.float64)
    idx = np.zeros(2 * len(f4) + 1, np.int32); co = np.zeros((2 * len(f4) + 1, 6)); n = C.c_int(0)
    L.ref_match_cloud(C.c_char(kind.encode()), _ptr(m4), len(m4), _ptr(f4), len(f4), _ptr(pose), int(n_neigh), int(bool(check_fov)), C.c_float(min_match_sq_dis),
                      C.c_float(min_plane_dis), _ptr(idx), _ptr(co), C.byref(n))
    valid = np.zeros(len(f4), np.uint8); coeffs = np.zeros((len(f4), 6))
    valid[idx[:n.value]] = 1; coeffs[idx[:n.value]] = co[:n.value]
    assert len(set(idx[:n.value].tolist())) == n.value
    return valid, coeffs


def ref_odom_good_feature_matching(kind, map_pts, feats, pivot, pose_i, ext, gf_ratio=0.8, seed=0, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """Estimator::goodFeatureMatching + evaluateFeatJacobian (estimator.cpp:1273-1517) compiled from the reference's own lines. Returns sel and rel_pose = the Pose of
    T_pivot^-1 T_i T_ext as that build computes it (4 x 4 products and inverse, rotation matrix -> quaternion: library arithmetic, restated in mini_eigen.hpp)."""
    L = ref_lib()
    m4 = np.zeros((len(map_pts), 4), np.float32); m4[:, :3] = np.asarray(map_pts)[:, :3]
    f4 = np.zeros((len(feats), 4), np.float32); f4[:, :min(4, np.asarray(feats).shape[1])] = np.asarray(feats)[:, :4]
    a = [np.ascontiguousarray(x, np.float64) for x in (pivot, pose_i, ext)]
    sel = np.zeros(max(len(f4), 1), np.int32); n = C.c_int(0); rel = np.zeros(7)
    L.ref_odom_good_feature_matching(C.c_char(kind.encode()), _ptr(m4), len(m4), _ptr(f4), len(f4), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), C.c_float(gf_ratio),
                                     C.c_uint(int(seed)), C.c_float(min_match_sq_dis), C.c_float(min_plane_dis), _ptr(sel), C.byref(n), _ptr(rel))
    return dict(sel=sel[:n.value].copy(), rel_pose=rel)


def ref_eval_full_hessian(map_points, kind, feats11, pose7, H=None, feat_num=0, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """ActiveFeatureSelection::evalFullHessian (lidar_mapper.h:176-227) + common::logDet(mat_H, true) from the reference's own lines"""
    L = ref_lib()
    m, f = _as11(map_points), _as11(feats11)
    pose = np.ascontiguousarray(pose7, np.float64)
    Hm = np.ascontiguousarray(np.eye(6) * 1e-6 if H is None else H, np.float64).copy()
    n = C.c_int(int(feat_num)); ld = C.c_double(0)
    L.ref_eval_full_hessian(C.c_char(kind.encode()), _ptr(m), len(m), _ptr(f), len(f), _ptr(pose), C.c_float(min_match_sq_dis), C.c_float(min_plane_dis),
                            _ptr(Hm), C.byref(n), C.byref(ld))
    return Hm, n.value, ld.value


def ref_track_match(kind, prev4, cur4, pose7, distance_sq_threshold=25.0, nearby_scan=2.5):
    """FeatureExtract::matchCornerFromScan ('c') / matchSurfFromScan ('s') compiled from the reference's own lines (feature_extract.hpp:131-376)"""
    L = ref_lib()
    prev = np.ascontiguousarray(prev4, np.float32); cur = np.ascontiguousarray(cur4, np.float32)
    pose = np.ascontiguousarray(pose7, np.float64)
    valid = np.zeros(len(cur), np.uint8); coeffs = np.zeros((len(cur), 6))
    L.ref_track_match(C.c_char(kind.encode()), _ptr(prev), len(prev), _ptr(cur), len(cur), _ptr(pose), C.c_float(distance_sq_threshold),
                      C.c_float(nearby_scan), _ptr(valid), _ptr(coeffs))
    return valid, coeffs


def ref_scan_factor_eval(kind, point, coeff, pose7, s=1.0):
    """LidarScanPlaneNormFactor ('S') / LidarScanEdgeFactorVector ('E') from the reference's own lines (lidar_scan_factor.hpp:25-62, 236-279)"""
    L = ref_lib()
    point = np.ascontiguousarray(point, np.float64)
    coeff = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    pose = np.ascontiguousarray(pose7, np.float64)
    rows = 1 if kind == "S" else 3
    r = np.zeros(rows); J = np.zeros((rows, 7))
    L.ref_scan_factor_eval(C.c_char(kind.encode()), _ptr(point), _ptr(coeff), C.c_double(s), _ptr(pose), _ptr(r), _ptr(J))
    return r, J


def ref_transform_to_end(points4, pose7, distortion=True, scan_period=0.1):
    """TransformToEnd (utility.h:79-100) from the reference's own lines"""
    L = ref_lib()
    pts = np.ascontiguousarray(points4, np.float32); pose = np.ascontiguousarray(pose7, np.float64)
    out = np.zeros_like(pts)
    L.ref_transform_to_end(_ptr(pts), len(pts), _ptr(pose), int(bool(distortion)), C.c_float(scan_period), _ptr(out))
    return out


def ref_cloud_uct_associate_to_map(pts11, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, trace_threshold):
    """cloudUCTAssociateToMap compiled from the reference's own lines (lidar_mapper_keyframe.cpp:1116-1158)"""
    L = ref_lib()
    p = np.ascontiguousarray(pts11, np.float32)
    pg, cg = np.ascontiguousarray(pose_global, np.float64), np.ascontiguousarray(cov_global, np.float64)
    e, ec = np.ascontiguousarray(ext, np.float64), np.ascontiguousarray(ext_cov, np.float64)
    cm = np.ascontiguousarray(cov_meas, np.float64)
    out = np.zeros_like(p); cnt = C.c_int(0)
    L.ref_cloud_uct_associate_to_map(_ptr(p), p.shape[0], _ptr(pg), _ptr(cg), _ptr(e), _ptr(ec), e.shape[0], _ptr(cm), int(bool(with_ua)),
                                     C.c_double(trace_threshold), _ptr(out), C.byref(cnt))
    return out[:cnt.value].copy()


def ref_eval_hessian(crs_rows, crs_cols, crs_values, n_cols=6):
    """evalHessian compiled from the reference's own lines (lidar_mapper_keyframe.cpp:1160-1169)"""
    rows = np.ascontiguousarray(crs_rows, np.int32); cols = np.ascontiguousarray(crs_cols, np.int32); vals = np.ascontiguousarray(crs_values, np.float64)
    H = np.zeros((6, 6))
    ref_lib().ref_eval_hessian(_ptr(rows), _ptr(cols), _ptr(vals), len(rows) - 1, int(n_cols), _ptr(H))
    return H


def ref_estimator_eval_degeneracy(crs_rows, crs_cols, crs_values, n_cols, opt_window_size, num_of_laser, eig_thre, estimate_extrinsic=True, frame_cnt=0,
                                  n_cumu_feature=10, lambda_thre_calib=70.0):
    """Estimator::evalDegenracy compiled from the reference's own lines (estimator.cpp:1598-1680) on a CRS Jacobian."""
    L = ref_lib()
    rows = np.ascontiguousarray(crs_rows, np.int32); cols = np.ascontiguousarray(crs_cols, np.int32); vals = np.ascontiguousarray(crs_values, np.float64)
    nb = n_cols // 6
    thr = np.ascontiguousarray(eig_thre, np.float64).copy()
    deg = np.zeros(nb, np.int32)
    V = np.zeros((nb, 6, 6))
    dfc = np.zeros(max(num_of_laser, 1))
    L.ref_estimator_eval_degeneracy(_ptr(rows), _ptr(cols), _ptr(vals), len(rows) - 1, int(n_cols), int(opt_window_size), int(num_of_laser), int(bool(estimate_extrinsic)),
                                    int(frame_cnt), int(n_cumu_feature), C.c_double(lambda_thre_calib), _ptr(thr), _ptr(deg), _ptr(V), _ptr(dfc))
    return dict(is_degenerate=deg.astype(bool), V_update=V, eig_thre=thr, d_factor_calib=dfc[:num_of_laser])


def ref_eval_degeneracy(H, eig_thre=100.0):
    """evalDegenracy (lidar_mapper_keyframe.cpp:1172-1204) from the reference's own lines, on a fresh PoseLocalParameterization"""
    L = ref_lib()
    H = np.ascontiguousarray(H, np.float64)
    V = np.zeros((6, 6)); ev = np.zeros(6); deg = C.c_int(0)
    L.ref_eval_degeneracy(_ptr(H), C.c_double(eig_thre), C.byref(deg), _ptr(V), _ptr(ev))
    return dict(is_degenerate=bool(deg.value), V_update=V, eigval=ev)


def ref_downsample_current_scan(surf4, corner4, leaf_surf, leaf_corner, ext_poses, ext_covs, cov_meas, with_ua=True, trace_threshold=0.6):
    """downsampleCurrentScan compiled from the reference's own lines (lidar_mapper_keyframe.cpp:356-421), its three filter objects being
    VoxelGridCovarianceMLOAM<PointI> compiled from the reference's own lines too (voxel_grid_covariance_mloam_impl.hpp:68-457). Returns (surf11, corner11)."""
    L = ref_lib()
    s4 = np.ascontiguousarray(surf4, np.float32); c4 = np.ascontiguousarray(corner4, np.float32)
    ep = np.ascontiguousarray(ext_poses, np.float64).reshape(-1, 7); ec = np.ascontiguousarray(ext_covs, np.float64).reshape(-1, 36)
    cm = np.ascontiguousarray(cov_meas, np.float64)
    so = np.zeros((max(len(s4), 1), 11), np.float32); co = np.zeros((max(len(c4), 1), 11), np.float32)
    ns, nc = C.c_int(0), C.c_int(0)
    L.ref_downsample_current_scan(_ptr(s4), len(s4), _ptr(c4), len(c4), C.c_float(leaf_surf), C.c_float(leaf_corner), _ptr(ep), _ptr(ec), len(ep), _ptr(cm),
                                  int(bool(with_ua)), C.c_double(trace_threshold), _ptr(so), C.byref(ns), _ptr(co), C.byref(nc))
    return so[:ns.value].copy(), co[:nc.value].copy()


def ref_voxel_filter(points, leaf, trace_threshold=-1.0):
    """VoxelGridCovarianceMLOAM<PointT>::filter compiled from the reference's own lines (voxel_grid_covariance_mloam_impl.hpp:68-457): (n, 4)
    PointXYZI records -> the plain branch, (n, 11) PointXYZIWithCov records -> the covariance branch. trace_threshold < 0: the class default."""
    L = ref_lib()
    p = np.ascontiguousarray(points, np.float32)
    assert p.shape[1] in (4, 11)
    out = np.zeros_like(p)
    cnt = C.c_int(0)
    rc = L.ref_voxel_filter(_ptr(p), p.shape[0], p.shape[1], C.c_float(leaf), C.c_float(trace_threshold), _ptr(out), C.byref(cnt))
    assert rc == 0
    return out[:cnt.value].copy()


def _solve_rows(solves, n):
    return [dict(n_blocks=int(o[0]), lm_iterations=int(o[1]), successful_steps=int(o[2]), termination=int(o[3]), initial_cost=o[4], final_cost=o[5])
            for o in solves[:n]]


def ref_scan2map(surf_map, corner_map, surf, corner, pose_init, with_ua=False, cov_meas=None, map_eig_thre=100.0, gf_method="wo_gf", gf_ratio=1.0,
                 seed=0, frame_cnt=1, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """scan2MapOptimization compiled from the reference's own lines (lidar_mapper_keyframe.cpp:423-639) over the Ceres-shaped shim (oracle/ref/ref_shim.cpp):
    matching, selection, block assembly, evalHessian / evalDegenracy, the two outer iterations, the final covariance -- the reference's text; the LM
    iteration inside ceres::Solve is oracle/lm.hpp. Maps / features: (n, >= 3) or (n, 11) arrays. Returns pose, one record per ceres::Solve, cov_mapping."""
    L = ref_lib()
    a = [_as11(x) for x in (surf_map, corner_map, surf, corner)]
    p0 = np.ascontiguousarray(pose_init, np.float64)
    cm = np.ascontiguousarray(np.diag([0.0025] * 3) if cov_meas is None else cov_meas, np.float64)
    pose = np.zeros(7); solves = np.zeros((8, 6)); n = C.c_int(0); cov = np.zeros((6, 6))
    rc = L.ref_scan2map_optimization(_ptr(a[0]), len(a[0]), _ptr(a[1]), len(a[1]), _ptr(a[2]), len(a[2]), _ptr(a[3]), len(a[3]), _ptr(p0), int(bool(with_ua)), _ptr(cm),
                                     C.c_double(map_eig_thre), gf_method.encode(), C.c_double(gf_ratio), C.c_uint(int(seed)), int(frame_cnt), C.c_float(min_match_sq_dis),
                                     C.c_float(min_plane_dis), _ptr(pose), _ptr(solves), 8, C.byref(n), _ptr(cov))
    assert rc == 0
    return dict(pose=pose, solves=_solve_rows(solves, n.value), cov=cov)


def ref_track_cloud(corner_last4, surf_last4, corner_sharp4, surf_flat4, pose_ini, distance_sq_threshold=25.0, nearby_scan=2.5):
    """LidarTracker::trackCloud compiled from the reference's own lines (lidar_tracker.cpp:23-129) over the same Ceres-shaped shim"""
    L = ref_lib()
    a = [np.ascontiguousarray(x, np.float32) for x in (corner_last4, surf_last4, corner_sharp4, surf_flat4)]
    p0 = np.ascontiguousarray(pose_ini, np.float64)
    pose = np.zeros(7); solves = np.zeros((4, 6)); n = C.c_int(0)
    rc = L.ref_track_cloud(_ptr(a[0]), len(a[0]), _ptr(a[1]), len(a[1]), _ptr(a[2]), len(a[2]), _ptr(a[3]), len(a[3]), _ptr(p0), C.c_float(distance_sq_threshold),
                           C.c_float(nearby_scan), _ptr(pose), _ptr(solves), 4, C.byref(n))
    assert rc == 0
    return dict(pose=pose, solves=_solve_rows(solves, n.value))


def ref_compound_pose_with_cov(pose1, cov1, pose2, cov2):
    """the reference's own compoundPoseWithCov lines (associate_uct.hpp:9-86, method 2)"""
    L = ref_lib()
    a = [np.ascontiguousarray(x, np.float64) for x in (pose1, cov1, pose2, cov2)]
    pose_cp, cov_cp = np.zeros(7), np.zeros((6, 6))
    L.ref_compound_pose_with_cov(_ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(pose_cp), _ptr(cov_cp))
    return pose_cp, cov_cp


def ref_eval_point_uncertainty(xyz, T4, cov_pose, cov_meas):
    """the reference's own evalPointUncertainty lines (associate_uct.hpp:164-215, both overloads) on one point; T4: 4x4 pose matrix"""
    L = ref_lib()
    p = np.ascontiguousarray(xyz, np.float32); T = np.ascontiguousarray(T4, np.float64)
    cp = np.ascontiguousarray(cov_pose, np.float64); cm = np.ascontiguousarray(cov_meas, np.float64)
    a, b = np.zeros((3, 3)), np.zeros((3, 3))
    L.ref_eval_point_uncertainty(_ptr(p), _ptr(T), _ptr(cp), _ptr(cm), _ptr(a), _ptr(b))
    return a, b


def extract(points: np.ndarray, scan_start: np.ndarray, scan_end: np.ndarray, tie_rule: int = 0):
    """FeatureExtract::extractCloud. points (n,4) f32 ring-major. tie_rule 0: the reference's comparator (order of equal curvatures =
    libstdc++'s std::sort); 1: (curvature, index) ascending with NaN last -- the documented rule the HIP kernel sorts by."""
    L = lib()
    L.orc_set_tie_rule(int(tie_rule))
    pts = np.ascontiguousarray(points, np.float32)
    n = pts.shape[0]
    ss = np.ascontiguousarray(scan_start, np.int32)
    se = np.ascontiguousarray(scan_end, np.int32)
    curv = np.zeros(n, np.float32)
    label = np.zeros(n, np.int32)
    picked = np.zeros(n, np.int32)
    bufs = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
    cnts = [C.c_int(0) for _ in range(5)]
    lf_ds = np.zeros((max(n, 1), 4), np.float32)
    ties = C.c_long(0)
    L.orc_extract(_ptr(pts), n, _ptr(ss), _ptr(se), len(ss), _ptr(curv), _ptr(label), _ptr(picked),
                  _ptr(bufs[0]), C.byref(cnts[0]), _ptr(bufs[1]), C.byref(cnts[1]), _ptr(bufs[2]), C.byref(cnts[2]),
                  _ptr(bufs[3]), C.byref(cnts[3]), _ptr(lf_ds), C.byref(cnts[4]), C.byref(ties))
    L.orc_set_tie_rule(0)
    return dict(curvature=curv, label=label, picked=picked,
                sharp=bufs[0][:cnts[0].value].copy(), less_sharp=bufs[1][:cnts[1].value].copy(),
                flat=bufs[2][:cnts[2].value].copy(), less_flat_raw=bufs[3][:cnts[3].value].copy(),
                less_flat_ds=lf_ds[:cnts[4].value].copy(), n_ties=ties.value)


def voxel_grid(points: np.ndarray, leaf: float) -> np.ndarray:
    L = lib()
    pts = np.ascontiguousarray(points, np.float32)
    out = np.zeros_like(pts)
    cnt = C.c_int(0)
    L.orc_voxel_grid(_ptr(pts), pts.shape[0], C.c_float(leaf), _ptr(out), C.byref(cnt))
    return out[:cnt.value].copy()


class Map:
    """A map cloud + exact kNN index (the pcl::KdTreeFLANN role)."""

    def __init__(self, pts: np.ndarray):
        self.pts = np.ascontiguousarray(pts, np.float32)
        assert self.pts.ndim == 2 and self.pts.shape[1] >= 3
        self.h = C.c_void_p(lib().orc_map_create(_ptr(self.pts), self.pts.shape[1], self.pts.shape[0]))

    def __del__(self):
        try:
            if self.h:
                lib().orc_map_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def rebuild_seconds(self) -> float:
        return lib().orc_map_rebuild_seconds(self.h)

    def knn(self, queries: np.ndarray, k: int = 5):
        q = np.ascontiguousarray(queries[:, :3], np.float32)
        idx = np.zeros((q.shape[0], k), np.int32)
        d2 = np.zeros((q.shape[0], k), np.float32)
        lib().orc_knn(self.h, _ptr(q), q.shape[0], k, _ptr(idx), _ptr(d2))
        return idx, d2

    def match(self, kind: str, feats: np.ndarray, pose7, n_neigh=5, check_fov=False, min_match_sq_dis=1.0, min_plane_dis=0.2):
        f = np.ascontiguousarray(feats, np.float32)
        n = f.shape[0]
        valid = np.zeros(n, np.uint8)
        coeffs = np.zeros((n, 6), np.float64)
        pose = np.ascontiguousarray(pose7, np.float64)
        lib().orc_match(self.h, C.c_char(kind.encode()), _ptr(f), f.shape[1], n, _ptr(pose), n_neigh, int(check_fov),
                        C.c_float(min_match_sq_dis), C.c_float(min_plane_dis), _ptr(valid), _ptr(coeffs))
        return valid, coeffs


def factor_eval(kind: str, point, coeff, cov_trace: float, pose7):
    point = np.ascontiguousarray(point, np.float64)
    coeff = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    pose = np.ascontiguousarray(pose7, np.float64)
    r = np.zeros(1)
    J = np.zeros(7)
    lib().orc_factor_eval(C.c_char(kind.encode()), _ptr(point), _ptr(coeff), C.c_double(cov_trace), _ptr(pose), _ptr(r), _ptr(J))
    return r[0], J


def pose_chain(wmap_curr_prev, wodom_prev, wodom_cur):
    """The pose the mapper starts the next frame from: transformUpdate + transformAssociateToMap (lidar_mapper_keyframe.cpp:145-160)."""
    L = lib()
    a, b, c = (np.ascontiguousarray(v, np.float64) for v in (wmap_curr_prev, wodom_prev, wodom_cur))
    out = np.zeros(7)
    L.orc_pose_chain(_ptr(a), _ptr(b), _ptr(c), _ptr(out))
    return out


def ref_pose_chain(wmap_curr_prev, wodom_prev, wodom_cur):
    """The same through the reference's own lines (Pose::operator*, Pose::inverse, transformUpdate, transformAssociateToMap)."""
    L = ref_lib()
    a, b, c = (np.ascontiguousarray(v, np.float64) for v in (wmap_curr_prev, wodom_prev, wodom_cur))
    out = np.zeros(7)
    L.ref_pose_chain(_ptr(a), _ptr(b), _ptr(c), _ptr(out))
    return out


def pose_plus(x, delta, V_update=None):
    x = np.ascontiguousarray(x, np.float64)
    d = np.ascontiguousarray(delta, np.float64)
    V = np.ascontiguousarray(np.eye(6) if V_update is None else V_update, np.float64)
    out = np.zeros(7)
    lib().orc_pose_plus(_ptr(x), _ptr(d), _ptr(V), _ptr(out))
    return out


def huber(a: float, s: float):
    rho = np.zeros(3)
    lib().orc_huber(C.c_double(a), C.c_double(s), _ptr(rho))
    return rho


def linearize(kind: str, feats, cov_trace, pose7, valid, coeffs, huber_delta=0.1):
    f = np.ascontiguousarray(feats, np.float32)
    n = f.shape[0]
    ct = None if cov_trace is None else np.ascontiguousarray(cov_trace, np.float64)
    pose = np.ascontiguousarray(pose7, np.float64)
    valid = np.ascontiguousarray(valid, np.uint8)
    coeffs = np.ascontiguousarray(coeffs, np.float64)
    r = np.zeros(n)
    J = np.zeros((n, 6))
    H = np.zeros((6, 6))
    g = np.zeros(6)
    cost = C.c_double(0)
    cnt = C.c_int(0)
    lib().orc_linearize(C.c_char(kind.encode()), _ptr(f), f.shape[1], n, _ptr(ct), _ptr(pose), _ptr(valid), _ptr(coeffs),
                        C.c_double(huber_delta), _ptr(r), _ptr(J), _ptr(H), _ptr(g), C.byref(cost), C.byref(cnt))
    return dict(r=r, J=J, H=H, g=g, cost=cost.value, count=cnt.value)


def window_eval_degeneracy(JtJ, n_pose_blocks, eig_thre, estimate_extrinsic=True, frame_cnt=0, n_cumu_feature=10, lambda_thre_calib=70.0):
    """Estimator::evalDegenracy (estimator.cpp:1598-1680) on the window's J^T J. Returns the per-block flags, V_update, eigenvalues, the UPDATED
    thresholds and d_factor_calib."""
    H = np.ascontiguousarray(JtJ, np.float64)
    D = H.shape[0]
    nb = D // 6
    thr = np.ascontiguousarray(eig_thre, np.float64).copy()
    deg = np.zeros(nb, np.int32)
    V = np.zeros((nb, 6, 6))
    ev = np.zeros((nb, 6))
    dfc = np.zeros(max(nb - n_pose_blocks, 1))
    lib().orc_window_eval_degeneracy(_ptr(H), D, int(n_pose_blocks), _ptr(thr), int(bool(estimate_extrinsic)), C.c_long(int(frame_cnt)), int(n_cumu_feature),
                                     C.c_double(lambda_thre_calib), _ptr(deg), _ptr(V), _ptr(ev), _ptr(dfc))
    return dict(is_degenerate=deg.astype(bool), V_update=V, eigval=ev, eig_thre=thr, d_factor_calib=dfc[:nb - n_pose_blocks])


def eval_degeneracy(H, eig_thre=100.0):
    H = np.ascontiguousarray(H, np.float64)
    ev = np.zeros(6)
    V = np.zeros((6, 6))
    P = np.zeros((6, 6))
    deg = C.c_int(0)
    lib().orc_eval_degeneracy(_ptr(H), C.c_double(eig_thre), _ptr(ev), _ptr(V), _ptr(P), C.byref(deg))
    return dict(eigval=ev, eigvec=V, V_update=P, is_degenerate=bool(deg.value))


def _cov_off(feats):
    return 4 if feats.shape[1] >= 10 else -1


def scan2map(surf_map: Map, corner_map: Map, surf, corner, pose_init, prm):
    s = np.ascontiguousarray(surf, np.float32)
    c = np.ascontiguousarray(corner, np.float32)
    p0 = np.ascontiguousarray(pose_init, np.float64)
    prm = np.ascontiguousarray(prm, np.float64)
    pose = np.zeros(7)
    stats = np.zeros((16, 64))
    n_outer = C.c_int(0)
    Hf = np.zeros((6, 6))
    lib().orc_scan2map(surf_map.h, corner_map.h, _ptr(s), s.shape[1], s.shape[0], _cov_off(s),
                       _ptr(c), c.shape[1], c.shape[0], _cov_off(c), _ptr(p0), _ptr(prm), _ptr(pose), _ptr(stats),
                       C.byref(n_outer), _ptr(Hf))
    outer = []
    for i in range(n_outer.value):
        o = stats[i]
        outer.append(dict(n_surf_sel=int(o[0]), n_corner_sel=int(o[1]), lm_iterations=int(o[2]), successful_steps=int(o[3]),
                          initial_cost=o[4], final_cost=o[5], termination=int(o[6]), is_degenerate=bool(o[7]),
                          eigval=o[8:14].copy(), pose_after=o[14:21].copy(), H0=o[21:57].reshape(6, 6).copy(),
                          evaluations=int(o[57])))
    return dict(pose=pose, outer=outer, H_final=Hf)


def good_feature_matching(map_: Map, kind: str, feats, pose7, prm):
    f = np.ascontiguousarray(feats, np.float32)
    n = f.shape[0]
    pose = np.ascontiguousarray(pose7, np.float64)
    prm = np.ascontiguousarray(prm, np.float64)
    sel = np.zeros(max(n, 1), np.int32)
    nsel = C.c_int(0)
    H = np.zeros((6, 6))
    matched = np.zeros(n, np.uint8)
    jaco = np.zeros((n, 6))
    lib().orc_good_feature_matching(map_.h, C.c_char(kind.encode()), _ptr(f), f.shape[1], n, _cov_off(f), _ptr(pose), _ptr(prm),
                                    _ptr(sel), C.byref(nsel), _ptr(H), _ptr(matched), _ptr(jaco))
    return dict(sel=sel[:nsel.value].copy(), H=H, matched=matched, jaco=jaco)


def odom_good_feature_matching(map_: Map, kind: str, feats, rel_pose, pivot, pose_i, ext, gf_ratio=0.8, seed=0, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """Estimator::goodFeatureMatching (estimator.cpp:1347-1517): the odometry's selection among one LiDAR's features of one frame against the window's local map;
    rel_pose = the Pose of T_pivot^-1 T_i T_ext (what the features are matched at), pivot / pose_i / ext = the three blocks the scored row is evaluated at.
    Returns sel (feature indices in selection order), matched (features the loop matched), jaco (their scored rows)."""
    f = np.ascontiguousarray(feats, np.float32)
    n = f.shape[0]
    a = [np.ascontiguousarray(x, np.float64) for x in (rel_pose, pivot, pose_i, ext)]
    sel = np.zeros(max(n, 1), np.int32); nsel = C.c_int(0)
    matched = np.zeros(n, np.uint8); jaco = np.zeros((n, 6))
    lib().orc_odom_good_feature_matching(map_.h, C.c_char(kind.encode()), _ptr(f), f.shape[1], n, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]),
                                         C.c_float(gf_ratio), C.c_uint(int(seed)), C.c_float(min_match_sq_dis), C.c_float(min_plane_dis), _ptr(sel), C.byref(nsel),
                                         _ptr(matched), _ptr(jaco))
    return dict(sel=sel[:nsel.value].copy(), matched=matched, jaco=jaco)


def eval_full_hessian(map_: Map, kind: str, feats, pose7, H=None, feat_num=0, min_match_sq_dis=1.0, min_plane_dis=0.2):
    """ActiveFeatureSelection::evalFullHessian (lidar_mapper.h:176-227): match ALL features, mat_H += J^T J of the matched ones (rows weighted
    by the point's uncertainty, NOT loss-corrected), feat_num += matches. H defaults to the caller's 1e-6 * I seed (cpp:461)."""
    f = np.ascontiguousarray(feats, np.float32)
    H = np.ascontiguousarray(np.eye(6) * 1e-6 if H is None else H, np.float64).copy()
    pose = np.ascontiguousarray(pose7, np.float64)
    n = C.c_int(int(feat_num))
    lib().orc_eval_full_hessian(map_.h, C.c_char(kind.encode()), _ptr(f), f.shape[1], f.shape[0], _cov_off(f), _ptr(pose),
                                C.c_float(min_match_sq_dis), C.c_float(min_plane_dis), _ptr(H), C.byref(n))
    return H, n.value


def gf_ratio_policy(gf_method: str, gf_ratio_ini: float, gf_deg_factor: float, map_deg_thre: float, gf_ratio_cur: float) -> float:
    """The every-10th-frame block of scan2MapOptimization (lidar_mapper_keyframe.cpp:474-492), statement by statement: returns the new
    gf_ratio_cur (unchanged when no branch assigns it -- an unknown method, or gd_float with a NaN factor)."""
    if gf_method == "wo_gf":
        gf_ratio_cur = 1.0
    elif gf_method in ("rnd", "fps", "gd_fix"):
        gf_ratio_cur = gf_ratio_ini
    elif gf_method == "gd_float":
        if gf_deg_factor > map_deg_thre:
            gf_ratio_cur = gf_ratio_ini
        elif gf_deg_factor <= map_deg_thre:
            gf_ratio_cur = 0.8
    return gf_ratio_cur


def gn_iterations(surf_map: Map, corner_map: Map, surf, corner, pose_init, prm, n_iters=5, n_threads=1):
    s = np.ascontiguousarray(surf, np.float32)
    c = np.ascontiguousarray(corner, np.float32)
    pose = np.ascontiguousarray(pose_init, np.float64).copy()
    prm = np.ascontiguousarray(prm, np.float64)
    stats = np.zeros((n_iters, 48))
    secs = C.c_double(0)
    lib().orc_gn_iterations(surf_map.h, corner_map.h, _ptr(s), s.shape[1], s.shape[0], _cov_off(s),
                            _ptr(c), c.shape[1], c.shape[0], _cov_off(c), _ptr(pose), _ptr(prm), n_iters, n_threads,
                            _ptr(stats), C.byref(secs))
    iters = []
    for o in stats:
        H = np.zeros((6, 6))
        q = 17
        for r in range(6):
            for cc in range(r, 6):
                H[r, cc] = H[cc, r] = o[q]
                q += 1
        iters.append(dict(n_surf=int(o[0]), n_corner=int(o[1]), cost=o[2], is_degenerate=bool(o[3]),
                          pose_after=o[4:11].copy(), g=o[11:17].copy(), H=H))
    return dict(pose=pose, iters=iters, seconds=secs.value)


def pure_odom_eval(kind: str, point, coeff, pivot, pose_i, ext, sqrt_info=1.0):
    point = np.ascontiguousarray(point, np.float64)
    coeff = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    pv, pi, pe = (np.ascontiguousarray(a, np.float64) for a in (pivot, pose_i, ext))
    r = np.zeros(1)
    J = np.zeros((3, 7))
    lib().orc_pure_odom_eval(C.c_char(kind.encode()), _ptr(point), _ptr(coeff), C.c_double(sqrt_info), _ptr(pv), _ptr(pi), _ptr(pe), _ptr(r), _ptr(J))
    return r[0], J


def pure_odom_eval_batch(types, points, coeffs6, sqrt_info, frame_idx, ext_idx, pivot, frames, exts):
    """LidarPureOdom*Factor::Evaluate for a whole window (types 0 plane / 1 edge) -> residuals (n,), J (n, 3, 7)."""
    t = np.ascontiguousarray(types, np.int32); p = np.ascontiguousarray(points, np.float64); c = np.ascontiguousarray(coeffs6, np.float64)
    si = np.ascontiguousarray(sqrt_info, np.float64); fi = np.ascontiguousarray(frame_idx, np.int32); ei = np.ascontiguousarray(ext_idx, np.int32)
    pv = np.ascontiguousarray(pivot, np.float64); fr = np.ascontiguousarray(frames, np.float64); ex = np.ascontiguousarray(exts, np.float64)
    r = np.zeros(len(t)); J = np.zeros((len(t), 3, 7))
    lib().orc_pure_odom_eval_batch(len(t), _ptr(t), _ptr(p), _ptr(c), _ptr(si), _ptr(fi), _ptr(ei), _ptr(pv), _ptr(fr), _ptr(ex), _ptr(r), _ptr(J))
    return r, J


def pure_odom_normal_eq(types, points, coeffs6, sqrt_info, frame_idx, ext_idx, pivot, frames, exts, huber_delta=1.0):
    """J^T J (D x D), J^T r, cost and residual count of the coupled window problem (estimator.cpp:687-848, 1577-1595); D = 6 (1 + F + E)."""
    t = np.ascontiguousarray(types, np.int32); p = np.ascontiguousarray(points, np.float64); c = np.ascontiguousarray(coeffs6, np.float64)
    si = None if sqrt_info is None else np.ascontiguousarray(sqrt_info, np.float64)
    fi = np.ascontiguousarray(frame_idx, np.int32); ei = np.ascontiguousarray(ext_idx, np.int32)
    pv = np.ascontiguousarray(pivot, np.float64); fr = np.ascontiguousarray(frames, np.float64).reshape(-1, 7); ex = np.ascontiguousarray(exts, np.float64).reshape(-1, 7)
    D = 6 * (1 + len(fr) + len(ex))
    H = np.zeros((D, D)); g = np.zeros(D); cost = C.c_double(0); cnt = C.c_int(0)
    lib().orc_pure_odom_normal_eq(len(t), _ptr(t), _ptr(p), _ptr(c), _ptr(si), _ptr(fi), _ptr(ei), _ptr(pv), _ptr(fr), len(fr), _ptr(ex), len(ex),
                                  C.c_double(huber_delta), _ptr(H), _ptr(g), C.byref(cost), C.byref(cnt))
    return dict(H=H, g=g, cost=cost.value, count=cnt.value)


def seg_params(vertical_scans=16, horizon_scans=1800, min_cluster_size=30, segment_valid_point_num=5, segment_valid_line_num=3,
               segment_theta=1.047, roi_range=1.0, segment_flag=True):
    """config_realvehicle_hercules.yaml:7-13, 102 defaults"""
    return np.array([vertical_scans, horizon_scans, min_cluster_size, segment_valid_point_num, segment_valid_line_num, segment_theta, roi_range,
                     float(segment_flag)], np.float64)


def segment_cloud(points4, prm=None):
    """ImageSegmenter::segmentCloud on an unordered cloud (n, 4) [x y z intensity] -> ring-major cloud, ScanInfo arrays, outliers, images"""
    prm = seg_params() if prm is None else np.ascontiguousarray(prm, np.float64)
    p = np.ascontiguousarray(points4, np.float32)
    n = len(p); vs, hs = int(prm[0]), int(prm[1])
    out = np.zeros((max(n, 1), 4), np.float32); outl = np.zeros((max(n, 1) + 1, 4), np.float32)
    ss = np.zeros(vs, np.int32); se = np.zeros(vs, np.int32)
    rng = np.zeros((vs, hs), np.float32); lab = np.zeros((vs, hs), np.int32); pix = np.zeros(max(n, 1), np.int32)
    no, nl = C.c_int(0), C.c_int(0)
    lib().orc_segment_cloud(_ptr(p), n, _ptr(prm), _ptr(out), C.byref(no), _ptr(outl), C.byref(nl), _ptr(ss), _ptr(se), _ptr(rng), _ptr(lab), _ptr(pix))
    return dict(cloud=out[:no.value].copy(), outlier=outl[:nl.value].copy(), scan_start=ss, scan_end=se, range_mat=rng, label_mat=lab, pixel_of_point=pix[:n].copy())


def eig3f(A):
    A = np.ascontiguousarray(A, np.float32)
    val = np.zeros(3, np.float32)
    vec = np.zeros((3, 3), np.float32)
    rc = lib().orc_eig3f(_ptr(A), _ptr(val), _ptr(vec))
    return val, vec, rc


def qr_solve(A, b):
    A = np.ascontiguousarray(A, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    x = np.zeros(3, np.float32)
    lib().orc_qr_solve(_ptr(A), _ptr(b), A.shape[0], _ptr(x))
    return x


def logdet(A):
    A = np.ascontiguousarray(A, np.float64)
    return lib().orc_logdet(_ptr(A), A.shape[0])


def eval_point_uncertainty(xyz, pose7, cov_pose, cov_meas):
    p = np.ascontiguousarray(xyz, np.float32)
    pose = np.ascontiguousarray(pose7, np.float64)
    cp = np.ascontiguousarray(cov_pose, np.float64)
    cm = np.ascontiguousarray(cov_meas, np.float64)
    out = np.zeros((p.shape[0], 3, 3))
    lib().orc_eval_point_uncertainty(_ptr(p), p.shape[0], p.shape[1], _ptr(pose), _ptr(cp), _ptr(cm), _ptr(out))
    return out


def voxel_grid_cov(pts11, leaf, trace_threshold):
    p = np.ascontiguousarray(pts11, np.float32)
    assert p.shape[1] == 11
    out = np.zeros_like(p)
    cnt = C.c_int(0)
    lib().orc_voxel_grid_cov(_ptr(p), p.shape[0], C.c_float(leaf), C.c_float(trace_threshold), _ptr(out), C.byref(cnt))
    return out[:cnt.value].copy()


def voxel_grid_mloam_plain(points, leaf, member_order=1):
    """VoxelGridCovarianceMLOAM<PointI> (no covariance fields): xyz mean, intensity of the voxel's LAST member. member_order 0 = the order
    std::sort leaves (the reference, libstdc++-defined for mixed-intensity voxels), 1 = point-index order (the rule the HIP path is pinned on)."""
    p = np.ascontiguousarray(points[:, :4], np.float32)
    out = np.zeros_like(p)
    cnt = C.c_int(0)
    lib().orc_voxel_grid_mloam_plain(_ptr(p), p.shape[0], C.c_float(leaf), int(member_order), _ptr(out), C.byref(cnt))
    return out[:cnt.value].copy()


def std_sort_permutation(keys):
    """the permutation libstdc++'s std::sort leaves for a comparator that sees the key only (one thread, the library's own call)"""
    k = np.ascontiguousarray(keys, np.int32)
    perm = np.zeros(len(k), np.int32)
    if len(k):
        lib().orc_std_sort_permutation(_ptr(k), len(k), _ptr(perm))
    return perm


def compound_pose_with_cov(pose1, cov1, pose2, cov2):
    a = [np.ascontiguousarray(x, np.float64) for x in (pose1, cov1, pose2, cov2)]
    pose_cp, cov_cp = np.zeros(7), np.zeros((6, 6))
    lib().orc_compound_pose_with_cov(_ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(pose_cp), _ptr(cov_cp))
    return pose_cp, cov_cp


def cloud_uct_associate_to_map(pts11, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, trace_threshold):
    p = np.ascontiguousarray(pts11, np.float32)
    assert p.shape[1] == 11
    pg, cg = np.ascontiguousarray(pose_global, np.float64), np.ascontiguousarray(cov_global, np.float64)
    e, ec = np.ascontiguousarray(ext, np.float64), np.ascontiguousarray(ext_cov, np.float64)
    cm = np.ascontiguousarray(cov_meas, np.float64)
    out = np.zeros_like(p)
    cnt = C.c_int(0)
    lib().orc_cloud_uct_associate_to_map(_ptr(p), p.shape[0], _ptr(pg), _ptr(cg), _ptr(e), _ptr(ec), e.shape[0], _ptr(cm),
                                         int(bool(with_ua)), C.c_double(trace_threshold), _ptr(out), C.byref(cnt))
    return out[:cnt.value].copy()


def transform_cloud_feature(points4, ext_pose, lidar_idx):
    """transformCloudFeature (estimator/src/utility/visualization.cpp:39-51): pcl::transformPointCloud with the float 4x4 of the
    extrinsic, then intensity <- LiDAR index. float32 throughout, row sums left to right, translation last."""
    p = np.ascontiguousarray(points4, np.float32)
    t = np.asarray(ext_pose[:3], np.float64)
    x, y, z, w = [float(v) for v in ext_pose[3:7]]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]).astype(np.float32)
    tf = t.astype(np.float32)
    out = np.empty((len(p), 4), np.float32)
    for r in range(3):
        out[:, r] = ((R[r, 0] * p[:, 0] + R[r, 1] * p[:, 1]) + R[r, 2] * p[:, 2]) + tf[r]
    out[:, 3] = np.float32(lidar_idx)
    return out


def transform_point_cloud(points4, pose7):
    """pcl::transformPointCloud(cloud, cloud, pose.T_.cast<float>()) (estimator.cpp:1185-1192): float32 R p + t, other fields kept."""
    out = transform_cloud_feature(points4, pose7, 0)
    out[:, 3] = np.ascontiguousarray(points4, np.float32)[:, 3]
    return out


def transform_to_end(points4, pose7, distortion=True, scan_period=0.1):
    """TransformToEnd (utility.h:79-100) over rows [x y z intensity]; intensity = ring id + relative time inside the sweep."""
    pts = np.ascontiguousarray(points4, np.float32)
    assert pts.shape[1] == 4
    pose = np.ascontiguousarray(pose7, np.float64)
    out = np.zeros_like(pts)
    lib().orc_transform_to_end(_ptr(pts), len(pts), _ptr(pose), int(bool(distortion)), C.c_float(scan_period), _ptr(out))
    return out


def track_params(distance_sq_threshold=25.0, nearby_scan=2.5, scan_period=0.1, huber_delta=0.1, max_outer=2, max_lm_iterations=4):
    return np.array([distance_sq_threshold, nearby_scan, scan_period, huber_delta, max_outer, max_lm_iterations], np.float64)


def track_match(kind, prev4, cur4, pose7, tprm=None):
    """matchCornerFromScan ('c') / matchSurfFromScan ('s'): prev4, cur4 rows [x y z ring]."""
    prev = np.ascontiguousarray(prev4, np.float32); cur = np.ascontiguousarray(cur4, np.float32)
    assert prev.shape[1] == 4 and cur.shape[1] == 4
    pose = np.ascontiguousarray(pose7, np.float64)
    tprm = track_params() if tprm is None else np.ascontiguousarray(tprm, np.float64)
    valid = np.zeros(len(cur), np.uint8); coeffs = np.zeros((len(cur), 6))
    lib().orc_track_match(C.c_char(kind.encode()), _ptr(prev), len(prev), _ptr(cur), len(cur), _ptr(pose), _ptr(tprm), _ptr(valid), _ptr(coeffs))
    return valid, coeffs


def scan_factor_eval(kind, point, coeff, pose7, s=1.0):
    point = np.ascontiguousarray(point, np.float64)
    coeff = np.ascontiguousarray(np.concatenate([np.asarray(coeff, np.float64), np.zeros(6)])[:6])
    pose = np.ascontiguousarray(pose7, np.float64)
    rows = 1 if kind == "S" else 3
    r = np.zeros(rows); J = np.zeros((rows, 7))
    lib().orc_scan_factor_eval(C.c_char(kind.encode()), _ptr(point), _ptr(coeff), C.c_double(s), _ptr(pose), _ptr(r), _ptr(J))
    return r, J


def track_cloud(corner_last4, surf_last4, corner_sharp4, surf_flat4, pose_ini, tprm=None):
    a = [np.ascontiguousarray(x, np.float32) for x in (corner_last4, surf_last4, corner_sharp4, surf_flat4)]
    p0 = np.ascontiguousarray(pose_ini, np.float64)
    tprm = track_params() if tprm is None else np.ascontiguousarray(tprm, np.float64)
    pose = np.zeros(7); stats = np.zeros((8, 16)); n_outer = C.c_int(0)
    lib().orc_track_cloud(_ptr(a[0]), len(a[0]), _ptr(a[1]), len(a[1]), _ptr(a[2]), len(a[2]), _ptr(a[3]), len(a[3]), _ptr(p0), _ptr(tprm),
                          _ptr(pose), _ptr(stats), C.byref(n_outer))
    outer = [dict(n_corner=int(o[0]), n_surf=int(o[1]), solved=bool(o[2]), lm_iterations=int(o[3]), initial_cost=o[4], final_cost=o[5],
                  termination=int(o[6]), pose_after=o[7:14].copy()) for o in stats[:n_outer.value]]
    return dict(pose=pose, outer=outer)
