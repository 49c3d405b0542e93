"""m-loam_amd -- MI355X-native (gfx950) scan-to-map hot path of M-LOAM.

The product is ``lib/libmloam_hip.so`` (hand-written HIP kernels behind the C-ABI of ``include/mloam_hip.h``) plus the
C++ facade under ``host/`` that mirrors the reference's ``FeatureExtract`` / factor / ``PoseLocalParameterization``
interfaces. This module is the thin ctypes harness the tests and ``bench.py`` use to call through that C-ABI; it
contains no compute and no CPU fallback: if the HIP library is missing or no GPU is visible it raises.

Import with ``importlib.import_module("m-loam_amd")`` (the directory name carries a hyphen).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLOAM_HIP_LIB") or os.path.join(_HERE, "lib", "libmloam_hip.so")   # override: instrumented debug builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mloam_hip.h")

SURF, CORNER = 0, 1
ALL_KINDS = -1
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_CHECK_FOV, FLAG_WITH_UA, FLAG_NO_LOSS = 1, 2, 4
GF_METHODS = {"wo_gf": 0, "rnd": 1, "fps": 2, "gd_fix": 3, "gd_float": 4}
K_KNN, K_FIT, K_LINEARIZE, K_SOLVE, K_GRID_BUILD, K_EXTRACT, K_ALLREDUCE, K_KNN_PRE, K_KNN_FIRST = range(9)
K_ALL = 0x1FF


class MlhError(RuntimeError):
    pass


class SolverOpts(C.Structure):
    _fields_ = [("min_match_sq_dis", C.c_float), ("min_plane_dis", C.c_float), ("huber_delta", C.c_double),
                ("map_eig_thre", C.c_double), ("cov_measurement_trace", C.c_double), ("flags", C.c_uint32),
                ("max_outer", C.c_int), ("max_lm_iterations", C.c_int), ("gf_method", C.c_int), ("gf_ratio", C.c_double),
                ("gf_seed", C.c_uint64)]


class BlockOpts(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("k_neigh", C.c_int * 8), ("eig_thre", C.c_double * 8), ("freeze", C.c_int * 8)]


class SegmentParams(C.Structure):
    _fields_ = [("vertical_scans", C.c_int32), ("horizon_scans", C.c_int32), ("min_cluster_size", C.c_int32), ("segment_valid_point_num", C.c_int32),
                ("segment_valid_line_num", C.c_int32), ("segment_theta", C.c_float), ("roi_range", C.c_double), ("segment_flag", C.c_int32)]


class TrackOpts(C.Structure):
    _fields_ = [("distance_sq_threshold", C.c_float), ("nearby_scan", C.c_float), ("huber_delta", C.c_double),
                ("max_outer", C.c_int32), ("max_lm_iterations", C.c_int32)]


class IterStat(C.Structure):
    _fields_ = [("n_surf", C.c_int32), ("n_corner", C.c_int32), ("is_degenerate", C.c_int32), ("lm_iterations", C.c_int32),
                ("successful_steps", C.c_int32), ("termination", C.c_int32), ("cost", C.c_double), ("final_cost", C.c_double),
                ("eigval", C.c_double * 6), ("H", C.c_double * 36), ("g", C.c_double * 6), ("pose_after", C.c_double * 7)]

    def as_dict(self):
        return dict(n_surf=self.n_surf, n_corner=self.n_corner, is_degenerate=bool(self.is_degenerate),
                    lm_iterations=self.lm_iterations, successful_steps=self.successful_steps, termination=self.termination,
                    cost=self.cost, final_cost=self.final_cost, eigval=np.array(self.eigval), H=np.array(self.H).reshape(6, 6),
                    g=np.array(self.g), pose_after=np.array(self.pose_after))


class DeviceInfo(C.Structure):
    """mlh_device_info (include/mloam_hip.h)."""
    _fields_ = [("cu_count", C.c_int32), ("cu_solver", C.c_int32), ("loop_blocks_per_cu", C.c_int32 * 3), ("loop_max_tiles", C.c_int32 * 3),
                ("scan_uploads_from_ahead", C.c_int32), ("loop_launches", C.c_uint64), ("loop_timeouts", C.c_uint64), ("loop_fallbacks", C.c_uint64)]

    def as_dict(self):
        return dict(cu_count=self.cu_count, cu_solver=self.cu_solver, loop_blocks_per_cu=list(self.loop_blocks_per_cu),
                    loop_max_tiles=list(self.loop_max_tiles), loop_launches=int(self.loop_launches), loop_timeouts=int(self.loop_timeouts),
                    loop_fallbacks=int(self.loop_fallbacks), scan_uploads_from_ahead=int(self.scan_uploads_from_ahead))


_lib = None


def load_library():
    """Loads libmloam_hip.so; raises MlhError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MlhError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(LIB_PATH)
    vp, ci, cf, cd = C.c_void_p, C.c_int, C.c_float, C.c_double
    lib.mlh_create.argtypes = [C.POINTER(vp), ci]
    lib.mlh_destroy.argtypes = [vp]
    lib.mlh_destroy.restype = None
    lib.mlh_last_error.argtypes = [vp]
    lib.mlh_last_error.restype = C.c_char_p
    lib.mlh_version.restype = C.c_char_p
    lib.mlh_stream.argtypes = [vp]
    lib.mlh_stream.restype = vp
    lib.mlh_synchronize.argtypes = [vp]
    lib.mlh_get_info.argtypes = [vp, C.POINTER(DeviceInfo)]
    lib.mlh_profile_enable.argtypes = [vp, ci]
    lib.mlh_comm_finalize.argtypes = [vp]
    lib.mlh_profile_sample.argtypes = [vp, ci]
    lib.mlh_profile_reset.argtypes = [vp]
    lib.mlh_profile_get.argtypes = [vp, ci, C.POINTER(cd), C.POINTER(C.c_longlong)]
    lib.mlh_scan_upload.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, ci]
    lib.mlh_scan_upload_ahead.argtypes = [vp, vp, ci, ci]
    lib.mlh_segment_params_default.argtypes = [C.POINTER(SegmentParams)]
    lib.mlh_segment_params_default.restype = None
    lib.mlh_segment_cloud.argtypes = [vp, vp, ci, ci, ci, ci, C.POINTER(SegmentParams), vp, C.POINTER(C.c_int32), vp, vp, vp, ci, C.POINTER(C.c_int32)]
    lib.mlh_extract_run.argtypes = [vp]
    lib.mlh_extract_fetch.argtypes = [vp, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_int32)]
    lib.mlh_extract_voxel_run.argtypes = [vp, cf]
    lib.mlh_extract_fetch_voxel.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    lib.mlh_point_uncertainty.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp, cd, vp, vp]
    dp = C.POINTER(C.c_double)
    lib.mlh_cloud_uct_associate_to_map.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, ci, cd, vp, C.POINTER(C.c_int32), ci]
    lib.mlh_compound_pose_with_cov.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mlh_downsample_current_scan.argtypes = [vp, ci, vp, ci, ci, ci, ci, cf, vp, vp, ci, vp, ci, cd, vp, C.POINTER(C.c_int32)]
    lib.mlh_track_opts_default.argtypes = [vp]
    lib.mlh_track_opts_default.restype = None
    lib.mlh_track_set_prev.argtypes = [vp, ci, vp, ci, ci, ci, ci, cf]
    lib.mlh_track_set_cur.argtypes = [vp, ci, vp, ci, ci, ci, ci]
    lib.mlh_track_set_from_scan.argtypes = [vp, ci, cf]
    lib.mlh_downsample_current_scan_pair.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, cf, cf, vp, vp, ci, vp, ci, C.c_double, vp, vp]
    lib.mlh_downsample_scan2map.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, cf, cf, vp, vp, ci, vp, ci, C.c_double, vp, vp, vp, vp]
    lib.mlh_voxel_grid.argtypes = [vp, vp, ci, ci, ci, cf, vp, vp, ci]
    lib.mlh_transform_point_cloud.argtypes = [vp, vp, ci, ci, vp, ci]
    lib.mlh_transform_to_end.argtypes = [vp, vp, ci, ci, ci, vp, ci, cf, ci]
    lib.mlh_scan_undistort.argtypes = [vp, vp, cf]
    lib.mlh_fuse_reset.argtypes = [vp]
    lib.mlh_fuse_add_scan.argtypes = [vp, ci, vp]
    lib.mlh_fuse_add_scan_from.argtypes = [vp, vp, ci, vp]
    lib.mlh_fuse_add_rings.argtypes = [vp, ci, ci, ci, vp]
    lib.mlh_fused_cloud.argtypes = [vp, ci, vp, vp]
    lib.mlh_track_match.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.mlh_track_cloud.argtypes = [vp, vp, vp, vp]
    lib.mlh_pure_odom_set.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    lib.mlh_pure_odom_evaluate.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp]
    lib.mlh_pure_odom_normal_eq.argtypes = [vp, vp, vp, ci, vp, ci, cd, vp, vp, C.POINTER(cd), C.POINTER(C.c_int32)]
    lib.mlh_pure_odom_gn_solve.argtypes = [vp, vp, vp, ci, vp, ci, cd, ci, C.c_uint32, vp, C.POINTER(cd), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.mlh_voxel_filter.argtypes = [vp, vp, ci, ci, ci, ci, ci, cf, cf, vp, C.POINTER(C.c_int32), ci]
    lib.mlh_map_set.argtypes = [vp, ci, vp, ci, ci, cf, ci]
    lib.mlh_map_set_pair.argtypes = [vp, vp, ci, vp, ci, ci, cf, ci]
    lib.mlh_map_set_pair_overlapped.argtypes = [vp, vp, ci, vp, ci, ci, cf, ci]
    lib.mlh_map_rebuild.argtypes = [vp, ci]
    lib.mlh_map_info.argtypes = [vp, ci, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(cd), C.POINTER(C.c_int32)]
    lib.mlh_set_voxel_member_order.argtypes = [vp, ci]
    lib.mlh_set_extract_tie_order.argtypes = [vp, ci]
    lib.mlh_set_gn_schedule.argtypes = [vp, ci, ci, ci]
    lib.mlh_scan2map_begin.argtypes = [vp, vp, C.POINTER(SolverOpts), ci]
    lib.mlh_scan2map_begin_chained.argtypes = [vp, vp, vp, C.POINTER(SolverOpts), ci]
    lib.mlh_scan2map_end.argtypes = [vp, vp, vp]
    lib.mlh_std_sort_permutation.argtypes = [vp, vp, ci, ci, vp, ci]
    lib.mlh_debug_bad_launch.argtypes = [vp]
    lib.mlh_pure_odom_begin.argtypes = [vp]
    lib.mlh_pure_odom_add_matches.argtypes = [vp, ci, vp, ci, C.c_uint32, cf, cf, ci, ci]
    lib.mlh_pure_odom_add_matches_gf.argtypes = [vp, ci, vp, vp, vp, vp, ci, C.c_uint32, cf, cf, ci, ci, cf, C.c_uint64, vp, C.POINTER(C.c_int32)]
    lib.mlh_knn.argtypes = [vp, ci, vp, ci, ci, vp, vp]
    lib.mlh_features_set.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci]
    lib.mlh_features_set_block.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci]
    lib.mlh_gn_solve_blocks.argtypes = [vp, vp, ci, C.POINTER(SolverOpts), C.POINTER(BlockOpts), vp]
    lib.mlh_match_linearize.argtypes = [vp, ci, vp, ci, C.c_uint32, cf, cf, cd, cd, vp, vp, vp, vp, vp, vp, C.POINTER(cd), C.POINTER(C.c_int32)]
    lib.mlh_match_coeffs.argtypes = [vp, ci, vp, vp, C.POINTER(C.c_int32)]
    lib.mlh_linearize.argtypes = [vp, ci, vp, C.c_uint32, cd, cd, vp, vp, vp, vp, C.POINTER(cd), C.POINTER(C.c_int32)]
    lib.mlh_good_feature_matching.argtypes = [vp, ci, vp, ci, cd, C.c_uint64, cf, cf, vp, C.POINTER(C.c_int32), vp, vp]
    lib.mlh_solver_opts_default.argtypes = [C.POINTER(SolverOpts)]
    lib.mlh_solver_opts_default.restype = None
    lib.mlh_gn_solve.argtypes = [vp, vp, ci, C.POINTER(SolverOpts), vp]
    lib.mlh_gn_solve_begin.argtypes = [vp, vp, ci, C.POINTER(SolverOpts)]
    lib.mlh_features_copy.argtypes = [vp, vp, ci]
    lib.mlh_gn_solve_begin_chained.argtypes = [vp, vp, vp, ci, C.POINTER(SolverOpts)]
    lib.mlh_gn_solve_end.argtypes = [vp, vp]
    lib.mlh_scan2map.argtypes = [vp, vp, C.POINTER(SolverOpts), vp]
    lib.mlh_shard_set.argtypes = [vp, vp, vp]
    lib.mlh_shard_set_features.argtypes = [vp, ci, ci]
    lib.mlh_comm_unique_id.argtypes = [vp]
    lib.mlh_comm_init.argtypes = [vp, ci, ci, vp]
    lib.mlh_p2p_mailbox.argtypes = [vp, vp]
    lib.mlh_p2p_comm_init.argtypes = [vp, ci, ci, vp]
    lib.mlh_allreduce_f64.argtypes = [vp, vp, ci]
    lib.mlh_pose_plus.argtypes = [vp, vp, vp, vp]
    lib.mlh_eval_degeneracy.argtypes = [vp, cd, vp, vp]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "mlh_create", "mlh_destroy", "mlh_last_error", "mlh_version", "mlh_stream", "mlh_synchronize", "mlh_get_info",
    "mlh_comm_finalize", "mlh_profile_enable", "mlh_profile_sample", "mlh_profile_reset", "mlh_profile_get",
    "mlh_segment_params_default", "mlh_segment_cloud", "mlh_scan_upload", "mlh_scan_upload_ahead", "mlh_extract_run", "mlh_extract_fetch", "mlh_extract_voxel_run", "mlh_extract_fetch_voxel",
    "mlh_point_uncertainty", "mlh_downsample_current_scan", "mlh_voxel_filter", "mlh_pure_odom_set", "mlh_pure_odom_evaluate", "mlh_pure_odom_normal_eq", "mlh_track_opts_default", "mlh_track_set_prev", "mlh_track_set_cur",
    "mlh_track_set_from_scan", "mlh_downsample_current_scan_pair", "mlh_downsample_scan2map", "mlh_voxel_grid", "mlh_transform_point_cloud", "mlh_transform_to_end", "mlh_scan_undistort", "mlh_fuse_reset", "mlh_fuse_add_scan", "mlh_fuse_add_scan_from", "mlh_fuse_add_rings", "mlh_fused_cloud", "mlh_track_match", "mlh_track_cloud", "mlh_cloud_uct_associate_to_map", "mlh_compound_pose_with_cov",
    "mlh_map_set", "mlh_map_set_pair", "mlh_map_set_pair_overlapped", "mlh_map_rebuild", "mlh_map_info", "mlh_set_voxel_member_order", "mlh_debug_bad_launch", "mlh_set_extract_tie_order", "mlh_set_gn_schedule", "mlh_std_sort_permutation", "mlh_pure_odom_begin", "mlh_pure_odom_add_matches", "mlh_pure_odom_add_matches_gf", "mlh_pure_odom_gn_solve", "mlh_knn", "mlh_features_set", "mlh_features_set_block", "mlh_gn_solve_blocks",
    "mlh_match_linearize", "mlh_match_coeffs", "mlh_linearize", "mlh_good_feature_matching", "mlh_solver_opts_default", "mlh_gn_solve", "mlh_gn_solve_begin", "mlh_gn_solve_begin_chained", "mlh_gn_solve_end", "mlh_features_copy", "mlh_scan2map", "mlh_scan2map_begin", "mlh_scan2map_begin_chained", "mlh_scan2map_end",
    "mlh_shard_set", "mlh_shard_set_features", "mlh_comm_unique_id", "mlh_comm_init", "mlh_p2p_mailbox", "mlh_p2p_comm_init", "mlh_allreduce_f64",
    "mlh_pose_plus", "mlh_eval_degeneracy",
]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceCloud:
    """A device-resident cloud owned by the library (mlh_fused_cloud): pointer, record count, record stride in bytes."""
    def __init__(self, ptr, n, stride=16):
        self.ptr, self.n, self.stride = ptr, n, stride


def _src(points):
    """(pointer, stride_bytes, n, mem, keepalive) for a numpy array (host), a torch CUDA tensor or a DeviceCloud (device)."""
    if isinstance(points, DeviceCloud):
        return C.c_void_p(points.ptr), points.stride, points.n, MEM_DEVICE, points
    if isinstance(points, np.ndarray):
        a = np.ascontiguousarray(points, np.float32)
        return a.ctypes.data_as(C.c_void_p), a.shape[1] * 4, a.shape[0], MEM_HOST, a
    # torch tensor on the GPU
    t = points.contiguous()
    assert t.is_cuda and t.dtype.itemsize == 4
    return C.c_void_p(t.data_ptr()), t.shape[1] * 4, t.shape[0], MEM_DEVICE, t


def default_track_opts(**kw) -> TrackOpts:
    o = TrackOpts()
    load_library().mlh_track_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def default_opts(**kw) -> SolverOpts:
    o = SolverOpts()
    load_library().mlh_solver_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Context:
    """One mlh_ctx (one HIP stream, device-resident buffers)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.mlh_create(C.byref(h), device)
        if rc != 0:
            raise MlhError(f"mlh_create failed ({rc}): no usable HIP device {device}")
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.mlh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise MlhError(f"mlh error {rc}: {self.lib.mlh_last_error(self.h).decode()}")

    def stream(self):
        return self.lib.mlh_stream(self.h)

    def synchronize(self):
        self._ck(self.lib.mlh_synchronize(self.h))

    def info(self) -> dict:
        """mlh_get_info: compute units, the in-kernel loops' residency gates and what became of them (launches / timeouts / fallbacks)."""
        di = DeviceInfo()
        self._ck(self.lib.mlh_get_info(self.h, C.byref(di)))
        return di.as_dict()

    # ---- profiling
    def profile_enable(self, kernel_mask=K_ALL):
        """kernel_mask: bit k brackets launches of kernel id k with HIP events (0/False = off, True = all)."""
        if kernel_mask is True:
            kernel_mask = K_ALL
        self._ck(self.lib.mlh_profile_enable(self.h, int(kernel_mask)))

    def profile_sample(self, every_n=1):
        self._ck(self.lib.mlh_profile_sample(self.h, int(every_n)))

    def profile_reset(self):
        self._ck(self.lib.mlh_profile_reset(self.h))

    def profile_get(self, kernel_id):
        ms, n = C.c_double(0), C.c_longlong(0)
        self._ck(self.lib.mlh_profile_get(self.h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- extraction
    def scan_upload(self, points, scan_start, scan_end):
        ptr, stride, n, mem, keep = _src(points)
        if mem == MEM_HOST:
            ss = np.ascontiguousarray(scan_start, np.int32)
            se = np.ascontiguousarray(scan_end, np.int32)
            self._ck(self.lib.mlh_scan_upload(self.h, ptr, stride, 12 if stride >= 16 else -1, n, _p(ss), _p(se), len(ss), mem))
        else:
            ss, se = scan_start.contiguous(), scan_end.contiguous()
            self._ck(self.lib.mlh_scan_upload(self.h, ptr, stride, 12 if stride >= 16 else -1, n, C.c_void_p(ss.data_ptr()), C.c_void_p(se.data_ptr()), ss.numel(), mem))
        self._scan_n = n

    def scan_upload_ahead(self, points):
        """mlh_scan_upload_ahead: the NEXT scan's points (a HOST array, kept alive and unchanged by the caller until the scan_upload that names it) go to the device
        now, beside the kernels of the current scan."""
        ptr, stride, n, mem, keep = _src(points)
        if mem != MEM_HOST:
            raise ValueError("scan_upload_ahead takes a host array")
        self._ck(self.lib.mlh_scan_upload_ahead(self.h, ptr, stride, n))

    def segment_cloud(self, points4, fetch=True, outlier_capacity=None, **kw):
        """ImageSegmenter::segmentCloud: unordered cloud (n, 4) [x y z intensity] (numpy or torch CUDA) -> the context's scan (ring-major, on the
        device) and, with fetch, the ring-major cloud / ScanInfo arrays / outlier cloud. kw: fields of SegmentParams."""
        prm = SegmentParams()
        self.lib.mlh_segment_params_default(C.byref(prm))
        for k, v in kw.items():
            setattr(prm, k, v)
        ptr, stride, n, mem, keep = _src(points4)
        vs = prm.vertical_scans
        out = np.zeros((max(n, 1), 4), np.float32) if fetch else None
        cap = min(n, vs * ((prm.horizon_scans + 4) // 5)) + 1 if outlier_capacity is None else int(outlier_capacity)
        outl = np.zeros((max(cap, 1), 4), np.float32) if fetch else None
        ss = np.zeros(vs, np.int32); se = np.zeros(vs, np.int32)
        no, nl = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.mlh_segment_cloud(self.h, ptr, stride, 12 if stride >= 16 else -1, n, mem, C.byref(prm), _p(out) if fetch else None, C.byref(no),
                                            _p(ss), _p(se), _p(outl) if fetch else None, cap, C.byref(nl)))
        self._scan_n = no.value
        return dict(cloud=out[:no.value].copy() if fetch else None, outlier=outl[:min(nl.value, cap)].copy() if fetch else None, n_outlier=nl.value, scan_start=ss, scan_end=se, n=no.value)

    def extract_run(self):
        self._ck(self.lib.mlh_extract_run(self.h))

    def extract_fetch(self):
        n = self._scan_n
        label = np.zeros(n, np.int32)
        curv = np.zeros(n, np.float32)
        picked = np.zeros(n, np.int32)
        lists = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
        ptrs = (C.c_void_p * 4)(*[l.ctypes.data_as(C.c_void_p) for l in lists])
        cnt = (C.c_int32 * 4)()
        self._ck(self.lib.mlh_extract_fetch(self.h, _p(label), _p(curv), _p(picked), ptrs, cnt))
        names = ["sharp", "less_sharp", "flat", "less_flat_raw"]
        out = dict(label=label, curvature=curv, picked=picked)
        for i, nm in enumerate(names):
            out[nm] = lists[i][:cnt[i]].copy()
        return out

    def extract(self, points, scan_start, scan_end, voxel_leaf=None):
        self.scan_upload(points, scan_start, scan_end)
        self.extract_run()
        out = self.extract_fetch()
        if voxel_leaf:
            out["less_flat_ds"] = self.extract_voxel(voxel_leaf)
        return out

    def extract_voxel_run(self, leaf=0.2):
        """The per-ring VoxelGrid of the less-flat points, result kept on the device (for fuse_add_scan / track_set_from_scan)."""
        self._ck(self.lib.mlh_extract_voxel_run(self.h, leaf))

    def extract_voxel(self, leaf=0.2):
        """surf_points_less_flat after the per-ring VoxelGrid (feature_extract.cpp:266-271)."""
        self._ck(self.lib.mlh_extract_voxel_run(self.h, leaf))
        buf = np.zeros((max(self._scan_n, 1), 4), np.float32)
        n = C.c_int32(0)
        self._ck(self.lib.mlh_extract_fetch_voxel(self.h, _p(buf), C.byref(n)))
        return buf[:n.value].copy()

    def point_uncertainty(self, points, ext_poses, ext_covs, cov_measurement, trace_threshold=0.0):
        """points (n, >=4) [x y z lidar-id ...] -> (cov_vec (n, 6) f32, keep (n,) bool)."""
        ptr, stride, n, mem, keep = _src(points)
        ep = np.ascontiguousarray(ext_poses, np.float64).reshape(-1, 7)
        ec = np.ascontiguousarray(ext_covs, np.float64).reshape(-1, 36)
        cm = np.ascontiguousarray(cov_measurement, np.float64).reshape(9)
        cov = np.zeros((n, 6), np.float32)
        kp = np.zeros(n, np.int32)
        self._ck(self.lib.mlh_point_uncertainty(self.h, ptr, stride, n, 12, mem, _p(ep), _p(ec), ep.shape[0], _p(cm), trace_threshold, _p(cov), _p(kp)))
        return cov, kp.astype(bool)

    # ---- scan-to-scan odometry (LidarTracker)
    def track_set_prev(self, kind, points4, distance_sq_threshold=25.0):
        ptr, stride, n, mem, keep = _src(points4)
        self._ck(self.lib.mlh_track_set_prev(self.h, kind, ptr, stride, n, 12, mem, distance_sq_threshold))

    def track_set_cur(self, kind, points4):
        ptr, stride, n, mem, keep = _src(points4)
        self._ck(self.lib.mlh_track_set_cur(self.h, kind, ptr, stride, n, 12, mem))
        self._m_track = getattr(self, "_m_track", {})
        self._m_track[kind] = n

    def track_set_from_scan(self, which, distance_sq_threshold=25.0):
        """which = 0: current frame <- this context's scan (sharp / flat); 1: previous frame <- (less sharp / thinned less flat)."""
        self._ck(self.lib.mlh_track_set_from_scan(self.h, which, distance_sq_threshold))

    def voxel_grid(self, points4, leaf):
        """pcl::VoxelGrid<PointXYZI> over rows [x y z intensity] -> centroids (all four fields averaged), ascending voxel index."""
        ptr, stride, n, mem, keep = _src(points4)
        assert stride == 16 and mem == MEM_HOST
        out = np.zeros((n, 4), np.float32)
        cnt = C.c_int32(0)
        self._ck(self.lib.mlh_voxel_grid(self.h, ptr, 16, n, 12, float(leaf), _p(out), C.byref(cnt), MEM_HOST))
        return out[:cnt.value].copy()

    def transform_point_cloud(self, points4, pose):
        """pcl::transformPointCloud with the float 4x4 of `pose` (host array in, transformed copy out; intensity kept)."""
        a = np.array(points4, np.float32, order="C", copy=True)
        ps = np.ascontiguousarray(pose, np.float64).reshape(7)
        self._ck(self.lib.mlh_transform_point_cloud(self.h, _p(a), a.shape[1] * 4, a.shape[0], _p(ps), MEM_HOST))
        return a

    def transform_to_end(self, points4, pose, distortion=True, scan_period=0.1):
        """TransformToEnd over rows [x y z intensity] (host array in, transformed copy out)."""
        a = np.array(points4, np.float32, order="C", copy=True)
        assert a.shape[1] == 4
        ps = np.ascontiguousarray(pose, np.float64).reshape(7)
        self._ck(self.lib.mlh_transform_to_end(self.h, _p(a), 16, a.shape[0], 12, _p(ps), int(bool(distortion)), float(scan_period), MEM_HOST))
        return a

    def scan_undistort(self, pose_undist, scan_period=0.1):
        """Estimator::undistortMeasurements on the scan this context holds (in place, on the device)."""
        ps = np.ascontiguousarray(pose_undist, np.float64).reshape(7)
        self._ck(self.lib.mlh_scan_undistort(self.h, _p(ps), float(scan_period)))

    def fuse_reset(self):
        self._ck(self.lib.mlh_fuse_reset(self.h))

    def fuse_add_scan(self, lidar_idx, ext_pose):
        """transformCloudFeature for the scan this context holds: append its mapping features, in the body frame, to the fused clouds."""
        e = np.ascontiguousarray(ext_pose, np.float64).reshape(7)
        self._ck(self.lib.mlh_fuse_add_scan(self.h, int(lidar_idx), _p(e)))

    def fuse_add_scan_from(self, src, lidar_idx, ext_pose):
        """The same for the scan ANOTHER context of this device holds (extracted + voxel-thinned there): device to device, ordered by events."""
        e = np.ascontiguousarray(ext_pose, np.float64).reshape(7)
        self._ck(self.lib.mlh_fuse_add_scan_from(self.h, src.h, int(lidar_idx), _p(e)))

    def fuse_add_rings(self, ring_begin, ring_end, lidar_idx, ext_pose):
        """The same for rings [ring_begin, ring_end) of a scan that holds several LiDARs back to back."""
        e = np.ascontiguousarray(ext_pose, np.float64).reshape(7)
        self._ck(self.lib.mlh_fuse_add_rings(self.h, int(ring_begin), int(ring_end), int(lidar_idx), _p(e)))

    def fused_cloud(self, kind) -> "DeviceCloud":
        ptr, n = C.c_void_p(), C.c_int32(0)
        self._ck(self.lib.mlh_fused_cloud(self.h, kind, C.byref(ptr), C.byref(n)))
        return DeviceCloud(ptr.value, n.value)

    def track_match(self, kind, pose, opts=None):
        opts = opts or default_track_opts()
        pose = np.ascontiguousarray(pose, np.float64)
        m = self._m_track[kind]
        valid = np.zeros(m, np.uint8); coeffs = np.zeros((m, 6))
        self._ck(self.lib.mlh_track_match(self.h, kind, _p(pose), C.byref(opts), _p(valid), _p(coeffs)))
        return valid, coeffs

    def track_cloud(self, pose_ini, opts=None, want_stats=True):
        opts = opts or default_track_opts()
        pose = np.ascontiguousarray(pose_ini, np.float64).copy()
        if not want_stats:
            self._ck(self.lib.mlh_track_cloud(self.h, _p(pose), C.byref(opts), None))
            return pose, None
        stats = (IterStat * opts.max_outer)()
        self._ck(self.lib.mlh_track_cloud(self.h, _p(pose), C.byref(opts), C.cast(stats, C.c_void_p)))
        return pose, [s.as_dict() for s in stats]

    def pure_odom_set(self, types, points, coeffs, frame_idx, ext_idx, sqrt_info=None):
        t = np.ascontiguousarray(types, np.int32); fi = np.ascontiguousarray(frame_idx, np.int32); ei = np.ascontiguousarray(ext_idx, np.int32)
        p = np.ascontiguousarray(points, np.float64); c = np.ascontiguousarray(coeffs, np.float64)
        assert p.shape == (len(t), 3) and c.shape == (len(t), 6)
        si = None if sqrt_info is None else np.ascontiguousarray(sqrt_info, np.float64)
        self._ck(self.lib.mlh_pure_odom_set(self.h, len(t), _p(t), _p(p), _p(c), _p(si) if si is not None else None, _p(fi), _p(ei)))
        self._n_odom = len(t)

    def pure_odom_begin(self):
        """start a device-resident LidarPureOdom factor table (filled by pure_odom_add_matches)"""
        self._ck(self.lib.mlh_pure_odom_begin(self.h))
        self._n_odom = 0

    def pure_odom_add_matches(self, kind, rel_pose, frame_idx, ext_idx, k_neigh=5, flags=0, min_match_sq_dis=1.0, min_plane_dis=0.2):
        """match the staged features of `kind` at rel_pose = T_pivot^-1 T_frame T_ext and append the valid ones as factors, all on the device"""
        p = np.ascontiguousarray(rel_pose, np.float64)
        self._ck(self.lib.mlh_pure_odom_add_matches(self.h, kind, _p(p), int(k_neigh), int(flags), float(min_match_sq_dis), float(min_plane_dis),
                                                    int(frame_idx), int(ext_idx)))

    def pure_odom_add_matches_gf(self, kind, rel_pose, pivot, pose_i, ext, frame_idx, ext_idx, gf_ratio=0.8, seed=0, k_neigh=5, flags=0, min_match_sq_dis=1.0,
                                 min_plane_dis=0.2):
        """pure_odom_add_matches behind the odometry's good-feature selection (Estimator::goodFeatureMatching, estimator.cpp:1347-1517) -> the selected feature
        indices in selection order; only they are appended as factors"""
        a = [np.ascontiguousarray(x, np.float64) for x in (rel_pose, pivot, pose_i, ext)]
        sel = np.zeros(max(self._m[kind], 1), np.int32)
        n = C.c_int32(0)
        self._ck(self.lib.mlh_pure_odom_add_matches_gf(self.h, kind, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), int(k_neigh), int(flags), float(min_match_sq_dis),
                                                       float(min_plane_dis), int(frame_idx), int(ext_idx), float(gf_ratio), int(seed), _p(sel), C.byref(n)))
        return sel[:n.value].copy()

    def pure_odom_evaluate(self, pivot, frames, exts, want_jacobians=True):
        pv = np.ascontiguousarray(pivot, np.float64); fr = np.ascontiguousarray(frames, np.float64).reshape(-1, 7)
        ex = np.ascontiguousarray(exts, np.float64).reshape(-1, 7)
        r = np.zeros(self._n_odom); J = np.zeros((self._n_odom, 3, 7)) if want_jacobians else None
        self._ck(self.lib.mlh_pure_odom_evaluate(self.h, _p(pv), _p(fr), len(fr), _p(ex), len(ex), _p(r), _p(J) if J is not None else None))
        return r, J

    def pure_odom_normal_eq(self, pivot, frames, exts, huber_delta=1.0):
        """J^T J, J^T r, cost, count of the coupled window problem over [pivot | frames | extrinsics] (6 local parameters each)."""
        pv = np.ascontiguousarray(pivot, np.float64); fr = np.ascontiguousarray(frames, np.float64).reshape(-1, 7)
        ex = np.ascontiguousarray(exts, np.float64).reshape(-1, 7)
        D = 6 * (1 + len(fr) + len(ex))
        H = np.zeros((D, D)); g = np.zeros(D); cost = C.c_double(0); cnt = C.c_int32(0)
        self._ck(self.lib.mlh_pure_odom_normal_eq(self.h, _p(pv), _p(fr), len(fr), _p(ex), len(ex), float(huber_delta), _p(H), _p(g), C.byref(cost), C.byref(cnt)))
        return dict(H=H, g=g, cost=cost.value, count=cnt.value)

    def pure_odom_gn_solve(self, pivot, frames, exts, n_iters=5, huber_delta=1.0, const_blocks=None, V_update=None):
        """the coupled window problem [pivot | frames | extrinsics] solved on the device (mlh_pure_odom_gn_solve): n_iters Gauss-Newton iterations on the staged
        factor table; const_blocks: indices into [pivot, frames..., exts...] held constant (default: the pivot and extrinsic 0, as Estimator::optimizeMap does)."""
        pv = np.ascontiguousarray(pivot, np.float64)
        fr = np.ascontiguousarray(frames, np.float64).reshape(-1, 7).copy(); ex = np.ascontiguousarray(exts, np.float64).reshape(-1, 7).copy()
        nb = 1 + len(fr) + len(ex)
        const_blocks = [0, 1 + len(fr)] if const_blocks is None else list(const_blocks)
        mask = 0
        for b in const_blocks:
            mask |= 1 << int(b)
        V = None if V_update is None else np.ascontiguousarray(V_update, np.float64).reshape(nb, 36)
        cost, n, st = C.c_double(0), C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.mlh_pure_odom_gn_solve(self.h, _p(pv), _p(fr), len(fr), _p(ex), len(ex), float(huber_delta), int(n_iters), mask, _p(V), C.byref(cost), C.byref(n), C.byref(st)))
        return dict(frames=fr, exts=ex, cost=cost.value, count=n.value, status=st.value)

    def downsample_current_scan(self, kind, points4, leaf, ext_poses, ext_covs, cov_measurement, with_ua=True, trace_threshold=0.6, fetch=True):
        """downsampleCurrentScan for one kind; the result becomes the kind's feature set and, with fetch, is also returned (m, 11)
        (without: the number of features; nothing leaves the device)."""
        ptr, stride, n, mem, keep = _src(points4)
        ep = np.ascontiguousarray(ext_poses, np.float64).reshape(-1, 7)
        ec = np.ascontiguousarray(ext_covs, np.float64).reshape(-1, 36)
        cm = np.ascontiguousarray(cov_measurement, np.float64).reshape(9)
        out = np.zeros((n, 11), np.float32) if fetch else None
        cnt = C.c_int32(0)
        self._ck(self.lib.mlh_downsample_current_scan(self.h, kind, ptr, stride, n, 12, mem, leaf, _p(ep), _p(ec), len(ep), _p(cm), int(bool(with_ua)),
                                                      float(trace_threshold), _p(out) if fetch else None, C.byref(cnt)))
        self._m = getattr(self, "_m", {})
        self._m[kind] = cnt.value
        return out[:cnt.value].copy() if fetch else cnt.value

    def downsample_current_scan_pair(self, surf4, corner4, leaf_surf, leaf_corner, ext_poses, ext_covs, cov_measurement, with_ua=True, trace_threshold=0.6):
        """downsampleCurrentScan for both kinds in one call (one thinning pipeline for the two fused clouds) -> (n_surf_features, n_corner_features)."""
        ps, ss, ns, ms, ks = _src(surf4)
        pc, sc, nc, mc, kc = _src(corner4)
        assert ss == sc and ms == mc
        ep = np.ascontiguousarray(ext_poses, np.float64).reshape(-1, 7)
        ec = np.ascontiguousarray(ext_covs, np.float64).reshape(-1, 36)
        cm = np.ascontiguousarray(cov_measurement, np.float64).reshape(9)
        a, b = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.mlh_downsample_current_scan_pair(self.h, ps, ns, pc, nc, ss, 12, ms, float(leaf_surf), float(leaf_corner), _p(ep), _p(ec), len(ep), _p(cm),
                                                           int(bool(with_ua)), float(trace_threshold), C.byref(a), C.byref(b)))
        self._m = getattr(self, "_m", {})
        self._m[SURF], self._m[CORNER] = a.value, b.value
        return a.value, b.value

    def downsample_scan2map(self, surf4, corner4, leaf_surf, leaf_corner, ext_poses, ext_covs, cov_measurement, pose, opts=None, with_ua=True, trace_threshold=0.6):
        """downsampleCurrentScan (both kinds) + scan2MapOptimization with no host read between them (mlh_downsample_scan2map) -> (pose, (n_surf_features, n_corner_features))."""
        opts = opts or default_opts()
        ps, ss, ns, ms, ks = _src(surf4)
        pc, sc, nc, mc, kc = _src(corner4)
        assert ss == sc and ms == mc
        ep = np.ascontiguousarray(ext_poses, np.float64).reshape(-1, 7)
        ec = np.ascontiguousarray(ext_covs, np.float64).reshape(-1, 36)
        cm = np.ascontiguousarray(cov_measurement, np.float64).reshape(9)
        x = np.ascontiguousarray(pose, np.float64).copy()
        a, b = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.mlh_downsample_scan2map(self.h, ps, ns, pc, nc, ss, 12, ms, float(leaf_surf), float(leaf_corner), _p(ep), _p(ec), len(ep), _p(cm),
                                                  int(bool(with_ua)), float(trace_threshold), _p(x), C.byref(opts), C.byref(a), C.byref(b)))
        self._m = getattr(self, "_m", {})
        self._m[SURF], self._m[CORNER] = a.value, b.value
        return x, (a.value, b.value)

    def cloud_uct_associate_to_map(self, points11, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, trace_threshold):
        """cloudUCTAssociateToMap on (n, 11) records [x y z i cov6 trace] -> kept, transformed records in input order."""
        a = np.ascontiguousarray(points11, np.float32)
        assert a.shape[1] == 11
        pg, cg = np.ascontiguousarray(pose_global, np.float64), np.ascontiguousarray(cov_global, np.float64)
        e, ec = np.ascontiguousarray(ext, np.float64), np.ascontiguousarray(ext_cov, np.float64)
        cm = np.ascontiguousarray(cov_meas, np.float64)
        out = np.zeros_like(a)
        cnt = C.c_int32(0)
        self._ck(self.lib.mlh_cloud_uct_associate_to_map(self.h, _p(a), 44, a.shape[0], 12, 16, 40, _p(pg), _p(cg), _p(e), _p(ec), e.shape[0],
                                                         _p(cm), int(bool(with_ua)), float(trace_threshold), _p(out), C.byref(cnt), MEM_HOST))
        return out[:cnt.value].copy()

    def cloud_uct_associate_to_map_device(self, d_points11, d_out, pose_global, cov_global, ext, ext_cov, cov_meas, with_ua, trace_threshold):
        """Device-resident variant: d_points11 / d_out are torch CUDA float32 tensors (n, 11); returns the number of records written."""
        ptr, stride, n, mem, keep = _src(d_points11)
        optr, ostride, on, omem, okeep = _src(d_out)
        assert mem == MEM_DEVICE and omem == MEM_DEVICE and stride == 44 and ostride == 44 and on >= n
        pg, cg = np.ascontiguousarray(pose_global, np.float64), np.ascontiguousarray(cov_global, np.float64)
        e, ec = np.ascontiguousarray(ext, np.float64), np.ascontiguousarray(ext_cov, np.float64)
        cm = np.ascontiguousarray(cov_meas, np.float64)
        cnt = C.c_int32(0)
        self._ck(self.lib.mlh_cloud_uct_associate_to_map(self.h, ptr, 44, n, 12, 16, 40, _p(pg), _p(cg), _p(e), _p(ec), e.shape[0], _p(cm),
                                                         int(bool(with_ua)), float(trace_threshold), optr, C.byref(cnt), MEM_DEVICE))
        return cnt.value

    def voxel_filter_device(self, d_points, d_out, leaf, trace_threshold=0.0):
        """Device-resident VoxelGridCovarianceMLOAM: torch CUDA float32 (n, 4) or (n, 11) in, same layout out; returns the count."""
        ptr, stride, n, mem, keep = _src(d_points)
        optr, ostride, on, omem, okeep = _src(d_out)
        assert mem == MEM_DEVICE and omem == MEM_DEVICE and stride == ostride and on >= n
        ncol = stride // 4
        cov_off, tr_off = (16, 40) if ncol >= 11 else (-1, -1)
        cnt = C.c_int32(0)
        self._ck(self.lib.mlh_voxel_filter(self.h, ptr, stride, n, 12 if ncol >= 4 else -1, cov_off, tr_off, leaf, trace_threshold, optr, C.byref(cnt), MEM_DEVICE))
        return cnt.value

    def voxel_filter(self, points, leaf, trace_threshold=0.0):
        """VoxelGridCovarianceMLOAM: points (n, 4) [x y z intensity] -> plain branch; (n, 11) [x y z i cov6 trace] -> covariance branch."""
        a = np.ascontiguousarray(points, np.float32)
        n, ncol = a.shape
        out = np.zeros_like(a)
        cnt = C.c_int32(0)
        cov_off, tr_off = (16, 40) if ncol >= 11 else (-1, -1)
        self._ck(self.lib.mlh_voxel_filter(self.h, _p(a), ncol * 4, n, 12 if ncol >= 4 else -1, cov_off, tr_off, leaf, trace_threshold, _p(out), C.byref(cnt), MEM_HOST))
        return out[:cnt.value].copy()

    # ---- map / features
    def map_set(self, kind, points, min_match_sq_dis=1.0):
        ptr, stride, n, mem, keep = _src(points)
        self._ck(self.lib.mlh_map_set(self.h, kind, ptr, stride, n, min_match_sq_dis, mem))

    def map_set_pair(self, surf_points, corner_points, min_match_sq_dis=1.0):
        """both setInputCloud calls of a frame in one set of launches (same memory space and record stride for the two clouds)"""
        ps, ss, ns, ms, ks = _src(surf_points)
        pc, sc, nc, mc, kc = _src(corner_points)
        assert ss == sc and ms == mc
        self._ck(self.lib.mlh_map_set_pair(self.h, ps, ns, pc, nc, ss, min_match_sq_dis, ms))

    def map_set_pair_overlapped(self, surf_points, corner_points, min_match_sq_dis=1.0):
        """stage the NEXT frame's maps while a submitted solve still runs (double-buffered map sets, second stream); see mlh_map_set_pair_overlapped"""
        ps, ss, ns, mems, _k1 = _src(surf_points)
        pc, sc, nc, memc, _k2 = _src(corner_points)
        assert ss == sc and mems == memc
        self._ck(self.lib.mlh_map_set_pair_overlapped(self.h, ps, ns, pc, nc, ss, min_match_sq_dis, mems))

    def map_rebuild(self, kind):
        self._ck(self.lib.mlh_map_rebuild(self.h, kind))

    def set_extract_tie_order(self, reference=True):
        """extractCloud, equal curvatures inside a sector: True = the order the reference's std::sort leaves (default), False = (curvature, index)"""
        self._ck(self.lib.mlh_set_extract_tie_order(self.h, 1 if reference else 0))

    def set_gn_schedule(self, deferred_finish=True, knn_warm_start=True, final_in_successor=True):
        """layout of a Gauss-Newton solve's iterations on the device (mlh_set_gn_schedule): results do not depend on it"""
        self._ck(self.lib.mlh_set_gn_schedule(self.h, 1 if deferred_finish else 0, 1 if knn_warm_start else 0, 1 if final_in_successor else 0))

    def debug_bad_launch(self):
        """one launch with an impossible configuration: raises under MLH_CHECK_LAUNCH=1 (naming the kernel), returns otherwise"""
        self._ck(self.lib.mlh_debug_bad_launch(self.h))

    def set_voxel_member_order(self, mode):
        """voxel filters of this context, members of a voxel: True / 1 = in libstdc++'s std::sort order (the reference's), produced on the
        device (the default); 2 or "host" = the same through a host pass with the platform's own std::sort; False / 0 = by point index
        (a voxel that mixes LiDAR ids may keep another id)"""
        mode = 2 if mode == "host" else int(mode)
        self._ck(self.lib.mlh_set_voxel_member_order(self.h, mode))

    def std_sort_permutation(self, keys, n0=None, mode=1):
        """the permutation std::sort (comparator on the key only) leaves for keys[:n0] and keys[n0:]; mode 1 device, 2 host"""
        keys = np.ascontiguousarray(keys, np.int32)
        n = len(keys)
        perm = np.zeros(n, np.int32)
        self._ck(self.lib.mlh_std_sort_permutation(self.h, _p(keys), n if n0 is None else int(n0), n, _p(perm), int(mode)))
        return perm

    def map_info(self, kind):
        n, occ, lanes, pop = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_double(0)
        self._ck(self.lib.mlh_map_info(self.h, kind, C.byref(n), C.byref(occ), C.byref(pop), C.byref(lanes)))
        return dict(n=n.value, occupied_cells=occ.value, mean_cell_population=round(pop.value, 2), knn_lanes=lanes.value)

    def knn(self, kind, queries, k=5):
        q = np.ascontiguousarray(queries[:, :3], np.float32)
        idx = np.zeros((q.shape[0], k), np.int32)
        d2 = np.zeros((q.shape[0], k), np.float32)
        self._ck(self.lib.mlh_knn(self.h, kind, _p(q), q.shape[0], k, _p(idx), _p(d2)))
        return idx, d2

    def features_set(self, kind, points):
        """points: (m, 4) [x y z intensity] or (m, 11) [x y z intensity cov6 trace] (compact PointXYZIWithCov)."""
        ptr, stride, n, mem, keep = _src(points)
        ncol = stride // 4
        self._ck(self.lib.mlh_features_set(self.h, kind, ptr, stride, n, 12 if ncol >= 4 else -1, 16 if ncol >= 10 else -1, mem))
        self._m = getattr(self, "_m", {})
        self._m[kind] = n

    def features_set_blocks(self, kind, clouds):
        """clouds: list of (m_b, >=3) arrays, one per pose block (block 0 = body pose, block n = extrinsic of LiDAR n)."""
        for b, pts in enumerate(clouds):
            ptr, stride, n, mem, keep = _src(pts)
            self._ck(self.lib.mlh_features_set_block(self.h, kind, b, ptr, stride, n, 16 if stride // 4 >= 10 else -1, mem))

    def gn_solve_blocks(self, poses, n_iters, k_neigh, eig_thre, freeze, opts: SolverOpts | None = None, want_stats=True):
        opts = opts or default_opts()
        poses = np.ascontiguousarray(poses, np.float64).copy().reshape(-1, 7)
        nb = poses.shape[0]
        bo = BlockOpts()
        bo.n_blocks = nb
        for b in range(nb):
            bo.k_neigh[b] = int(k_neigh[b]); bo.eig_thre[b] = float(eig_thre[b]); bo.freeze[b] = int(freeze[b])
        stats = (IterStat * (n_iters * nb))() if want_stats else None
        self._ck(self.lib.mlh_gn_solve_blocks(self.h, _p(poses), n_iters, C.byref(opts), C.byref(bo), C.cast(stats, C.c_void_p) if want_stats else None))
        st = [[stats[it * nb + b].as_dict() for b in range(nb)] for it in range(n_iters)] if want_stats else None
        return poses, st

    # ---- multi-GPU
    def comm_finalize(self):
        self._ck(self.lib.mlh_comm_finalize(self.h))

    def shard_set(self, lo_plane=None, hi_plane=None):
        lo = None if lo_plane is None else np.ascontiguousarray(lo_plane, np.float32)
        hi = None if hi_plane is None else np.ascontiguousarray(hi_plane, np.float32)
        self._ck(self.lib.mlh_shard_set(self.h, _p(lo), _p(hi)))

    def shard_set_features(self, n_ranks, rank):
        """replicated map, features dealt round-robin: slot f is owned iff f % n_ranks == rank"""
        self._ck(self.lib.mlh_shard_set_features(self.h, int(n_ranks), int(rank)))

    def comm_init(self, n_ranks, rank, unique_id: bytes):
        _torch_rccl_first()
        buf = C.create_string_buffer(unique_id, 128)
        self._ck(self.lib.mlh_comm_init(self.h, n_ranks, rank, C.cast(buf, C.c_void_p)))

    def p2p_mailbox(self) -> bytes:
        """this rank's mailbox of the mailbox communicator: its 64-byte hipIpc handle, to be all-gathered by the caller (mlh_p2p_mailbox)"""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.mlh_p2p_mailbox(self.h, C.cast(buf, C.c_void_p)))
        return buf.raw

    def p2p_comm_init(self, n_ranks, rank, handles):
        """handles: the n_ranks 64-byte handles in rank order (mlh_p2p_comm_init)"""
        blob = b"".join(handles)
        assert len(blob) == 64 * n_ranks
        buf = C.create_string_buffer(blob, len(blob))
        self._ck(self.lib.mlh_p2p_comm_init(self.h, n_ranks, rank, C.cast(buf, C.c_void_p)))

    def allreduce_f64(self, arr):
        a = np.ascontiguousarray(arr, np.float64).copy()
        self._ck(self.lib.mlh_allreduce_f64(self.h, _p(a), a.size))
        return a

    # ---- host-driven evaluation
    def match_linearize(self, kind, pose, flags=0, min_match_sq_dis=1.0, min_plane_dis=0.2, huber_delta=0.1,
                        cov_measurement_trace=0.0075, dense=True, k_neigh=5):
        m = self._m[kind]
        pose = np.ascontiguousarray(pose, np.float64)
        valid = np.zeros(m, np.uint8)
        coeffs = np.zeros((m, 6), np.float64)
        r = np.zeros(m) if dense else None
        J = np.zeros((m, 6)) if dense else None
        H = np.zeros((6, 6))
        g = np.zeros(6)
        cost, cnt = C.c_double(0), C.c_int32(0)
        self._ck(self.lib.mlh_match_linearize(self.h, kind, _p(pose), int(k_neigh), flags, min_match_sq_dis, min_plane_dis, huber_delta,
                                              cov_measurement_trace, _p(valid), _p(coeffs), _p(r), _p(J), _p(H), _p(g),
                                              C.byref(cost), C.byref(cnt)))
        return dict(valid=valid, coeffs=coeffs, r=r, J=J, H=H, g=g, cost=cost.value, count=cnt.value)

    def linearize(self, kind, pose, flags=0, huber_delta=0.1, cov_measurement_trace=0.0075, dense=True):
        m = self._m[kind]
        pose = np.ascontiguousarray(pose, np.float64)
        r = np.zeros(m) if dense else None
        J = np.zeros((m, 6)) if dense else None
        H = np.zeros((6, 6))
        g = np.zeros(6)
        cost, cnt = C.c_double(0), C.c_int32(0)
        self._ck(self.lib.mlh_linearize(self.h, kind, _p(pose), flags, huber_delta, cov_measurement_trace, _p(r), _p(J), _p(H), _p(g),
                                        C.byref(cost), C.byref(cnt)))
        return dict(r=r, J=J, H=H, g=g, cost=cost.value, count=cnt.value)

    def good_feature_matching(self, kind, pose, gf_method="gd_fix", gf_ratio=0.2, seed=0, min_match_sq_dis=1.0, min_plane_dis=0.2):
        m = self._m[kind]
        pose = np.ascontiguousarray(pose, np.float64)
        sel = np.zeros(max(m, 1), np.int32)
        n_sel = C.c_int32(0)
        H = np.eye(6) * 1e-6
        matched = np.zeros(m, np.uint8)
        self._ck(self.lib.mlh_good_feature_matching(self.h, kind, _p(pose), GF_METHODS[gf_method], gf_ratio, seed, min_match_sq_dis,
                                                    min_plane_dis, _p(sel), C.byref(n_sel), _p(H), _p(matched)))
        return dict(sel=sel[:n_sel.value].copy(), H=H, matched=matched)

    # ---- device-resident solvers
    def gn_solve(self, pose, n_iters, opts: SolverOpts | None = None, want_stats=True):
        opts = opts or default_opts()
        pose = np.ascontiguousarray(pose, np.float64).copy()
        stats = (IterStat * n_iters)() if want_stats else None
        self._ck(self.lib.mlh_gn_solve(self.h, _p(pose), n_iters, C.byref(opts), C.cast(stats, C.c_void_p) if want_stats else None))
        return pose, ([s.as_dict() for s in stats] if want_stats else None)

    def features_copy_from(self, src: "Context", kind):
        """this context's feature set of `kind` <- the one staged in `src` (same GPU), device to device (mlh_features_copy)"""
        self._ck(self.lib.mlh_features_copy(self.h, src.h, kind))

    def gn_solve_begin(self, pose, n_iters, opts: SolverOpts | None = None):
        """submit the solve and return at once (mlh_gn_solve_begin); collect the pose with gn_solve_end. One in flight per context."""
        self._opts_keep = opts or default_opts()
        pose = np.ascontiguousarray(pose, np.float64)
        self._ck(self.lib.mlh_gn_solve_begin(self.h, _p(pose), n_iters, C.byref(self._opts_keep)))

    def gn_solve_begin_chained(self, wodom_prev, wodom_cur, n_iters, opts: SolverOpts | None = None):
        """submit the next frame's solve; its start pose is made on the device from the previous solve's result (transformUpdate +
        transformAssociateToMap, lidar_mapper_keyframe.cpp:145-160) -- mlh_gn_solve_begin_chained."""
        self._opts_keep2 = opts or default_opts()
        a = np.ascontiguousarray(wodom_prev, np.float64)
        b = np.ascontiguousarray(wodom_cur, np.float64)
        self._ck(self.lib.mlh_gn_solve_begin_chained(self.h, _p(a), _p(b), n_iters, C.byref(self._opts_keep2)))

    def gn_solve_end(self):
        pose = np.zeros(7)
        self._ck(self.lib.mlh_gn_solve_end(self.h, _p(pose)))
        return pose

    def scan2map_begin(self, pose, opts: SolverOpts | None = None, lm_lookahead=0):
        """submit scan2MapOptimization for the staged frame and return (mlh_scan2map_begin); collect with scan2map_end"""
        opts = opts or default_opts()
        p = np.ascontiguousarray(pose, np.float64)
        self._ck(self.lib.mlh_scan2map_begin(self.h, _p(p), C.byref(opts), int(lm_lookahead)))

    def scan2map_begin_chained(self, wodom_prev, wodom_cur, opts: SolverOpts | None = None, lm_lookahead=0):
        opts = opts or default_opts()
        a, b = np.ascontiguousarray(wodom_prev, np.float64), np.ascontiguousarray(wodom_cur, np.float64)
        self._ck(self.lib.mlh_scan2map_begin_chained(self.h, _p(a), _p(b), C.byref(opts), int(lm_lookahead)))

    def scan2map_end(self):
        """-> (pose, status): 0 = converged inside the look-ahead, 2 = re-solved synchronously inside the call, 1 = the caller must re-solve (pose = start pose)"""
        p = np.zeros(7, np.float64)
        st = C.c_int32(0)
        self._ck(self.lib.mlh_scan2map_end(self.h, _p(p), C.byref(st)))
        return p, int(st.value)

    def scan2map(self, pose, opts: SolverOpts | None = None, want_stats=True):
        opts = opts or default_opts()
        pose = np.ascontiguousarray(pose, np.float64).copy()
        if not want_stats:
            self._ck(self.lib.mlh_scan2map(self.h, _p(pose), C.byref(opts), None))
            return pose, None
        stats = (IterStat * opts.max_outer)()
        self._ck(self.lib.mlh_scan2map(self.h, _p(pose), C.byref(opts), C.cast(stats, C.c_void_p)))
        return pose, [s.as_dict() for s in stats]


def compound_pose_with_cov(pose1, cov1, pose2, cov2):
    a = [np.ascontiguousarray(x, np.float64) for x in (pose1, cov1, pose2, cov2)]
    pose_cp, cov_cp = np.zeros(7), np.zeros((6, 6))
    rc = load_library().mlh_compound_pose_with_cov(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(pose_cp), _p(cov_cp))
    if rc:
        raise MlhError(f"mlh_compound_pose_with_cov -> {rc}")
    return pose_cp, cov_cp


def _torch_rccl_first():
    """A Python host that also uses torch.distributed ends up with torch's bundled librccl mapped; import torch BEFORE the library
    binds RCCL so that both use that one copy (the library binds an already-mapped librccl, see csrc/comm.hip)."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def comm_unique_id() -> bytes:
    _torch_rccl_first()
    buf = C.create_string_buffer(128)
    rc = load_library().mlh_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc:
        raise MlhError(f"mlh_comm_unique_id failed ({rc}): librccl not loadable?")
    return buf.raw


def pose_plus(x, delta, V_update=None):
    x = np.ascontiguousarray(x, np.float64)
    d = np.ascontiguousarray(delta, np.float64)
    V = None if V_update is None else np.ascontiguousarray(V_update, np.float64)
    out = np.zeros(7)
    rc = load_library().mlh_pose_plus(_p(x), _p(d), _p(V), _p(out))
    if rc:
        raise MlhError(f"mlh_pose_plus failed ({rc})")
    return out


def eval_degeneracy(H, eig_thre=100.0):
    H = np.ascontiguousarray(H, np.float64)
    ev = np.zeros(6)
    V = np.zeros((6, 6))
    deg = load_library().mlh_eval_degeneracy(_p(H), eig_thre, _p(ev), _p(V))
    return dict(eigval=ev, V_update=V, is_degenerate=bool(deg))
