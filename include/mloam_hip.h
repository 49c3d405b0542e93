/*
 * mloam_hip.h -- C-ABI of the MI355X-native (gfx950) scan-to-map hot path of M-LOAM.
 *
 * One shared library (libmloam_hip.so, hand-written HIP kernels) behind plain C entry points: opaque context,
 * plain pointers and sizes, int status (0 = ok, negative = error; text via mlh_last_error). No exceptions cross
 * the boundary. A context owns one HIP stream and all device buffers; distinct contexts may be used from
 * different threads concurrently (FeatureExtract::extractCloud is called from NUM_OF_LASER OpenMP threads,
 * estimator/src/estimator/estimator.cpp:249-263 -> one context per thread), a single context is not re-entrant.
 *
 * Every entry point names the reference interface it replaces (paths relative to the M-LOAM source tree).
 * Pose / parameter blocks are double[7] = [tx ty tz qx qy qz qw] exactly as the reference's Ceres parameter
 * blocks (estimator/src/factor/lidar_map_factor.hpp:46-47).
 *
 * Pointer arguments marked HOST are host memory, DEV are device memory on the context's device. Points are read
 * through (base, stride_bytes): x,y,z are 3 consecutive floats at the start of each record, so pcl::PointXYZI
 * (32 B), pcl::PointXYZIWithCov (48 B, mloam_pcl/include/mloam_pcl/point_with_cov.hpp:45-53) and packed float4
 * clouds can all be passed without repacking on the caller side.
 */
#ifndef MLOAM_HIP_H
#define MLOAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mlh_ctx mlh_ctx;

enum { MLH_SURF = 0, MLH_CORNER = 1 };              /* feature / map kind ('s' / 'c' in PointPlaneFeature::type_) */
#define MLH_ALL_KINDS (-1)                          /* mlh_map_rebuild: both maps in one set of launches */
enum { MLH_MEM_HOST = 0, MLH_MEM_DEVICE = 1 };

/* status codes */
enum {
    MLH_OK = 0,
    MLH_ERR_INVALID = -1,     /* bad argument */
    MLH_ERR_HIP = -2,         /* a HIP runtime call failed */
    MLH_ERR_STATE = -3,       /* call order violated (e.g. match before map_set) */
    MLH_ERR_NOMEM = -4,
    MLH_ERR_UNSUPPORTED = -5,
    MLH_ERR_INCOMPLETE = -6   /* mlh_scan2map_end with status_out == NULL on a frame whose pose_out is NOT a result (status 1 / 3 below) */
};

/* flags for mlh_match_linearize / solver options */
enum {
    MLH_FLAG_CHECK_FOV = 1u << 0,   /* CHECK_FOV argument of match*FromMap (feature_extract.hpp:696-715) */
    MLH_FLAG_WITH_UA = 1u << 1,     /* with_ua_flag: weight from the feature's own covariance (lidar_mapper_keyframe.cpp:541-544) */
    MLH_FLAG_NO_LOSS = 1u << 2      /* reduce un-corrected rows (ActiveFeatureSelection's H, lidar_mapper.h:162-164) */
};

/* ---------------------------------------------------------------- context */
int mlh_create(mlh_ctx **out, int device_id);
void mlh_destroy(mlh_ctx *ctx);
const char *mlh_last_error(const mlh_ctx *ctx);
const char *mlh_version(void);
/* the context's HIP stream (hipStream_t), for callers that interleave their own device work */
void *mlh_stream(mlh_ctx *ctx);
int mlh_synchronize(mlh_ctx *ctx);
/* What mlh_create found on the device, and what the in-kernel Levenberg-Marquardt loops (the one-launch forms of scan2MapOptimization's ceres::Solve,
 * lidar_mapper_keyframe.cpp:586-596, and of trackCloud's, lidar_tracker.cpp:106-113) did with it. Those launches synchronise their workgroups among themselves and
 * therefore need all of them resident at once: loop_max_tiles is the number of 256-feature tiles the context will put behind one such barrier --
 * min(hipOccupancyMaxActiveBlocksPerMultiprocessor, 6) x the compute units the solver's stream may use, less a margin of an eighth of those compute units (at
 * least 8; kernels of other streams and contexts are transient and only delay an arrival), never more than the 160 tiles beyond which every workgroup re-summing every record stops
 * paying (the fused thinning + solve call: 512). A frame with more tiles, a partitioned or smaller device, a CU-masked solver
 * stream (environment at mlh_create: MLH_SOLVER_CU_MASK=<hex word>[,<hex word>...], 32 compute units per word) take the launch-per-iteration forms: the same
 * arithmetic, the same pose bits, no residency requirement. Should a barrier nevertheless not complete within MLH_LOOP_TIMEOUT_US (default 20 000), the frame is
 * solved again through those forms (loop_fallbacks; the caller gets the same pose, later) and loop_max_tiles is halved for the context. MLH_LOOP_MAX_TILES=<n>
 * lowers the limits by hand (0: never use the one-launch loops). */
typedef struct mlh_device_info {
    int32_t cu_count;                /* compute units of the device */
    int32_t cu_solver;               /* ... the solver's stream may use */
    int32_t loop_blocks_per_cu[3];   /* occupancy query: scan2map's loop kernel, its device-counted variant (mlh_downsample_scan2map), the tracker's */
    int32_t loop_max_tiles[3];       /* the gates in force now, same order */
    int32_t scan_uploads_from_ahead; /* mlh_scan_upload calls that packed from what mlh_scan_upload_ahead had sent (wraps) */
    uint64_t loop_launches;          /* frames / tracker calls that went through the one-launch loops */
    uint64_t loop_timeouts;          /* ... whose barrier was given up on */
    uint64_t loop_fallbacks;         /* ... and that were solved again by the launch-per-iteration form inside the same call */
} mlh_device_info;
int mlh_get_info(mlh_ctx *ctx, mlh_device_info *out);

/* ---------------------------------------------------------------- per-kernel timing (HIP events on the context's stream)
 * Single-kernel ids (KNN, FIT, LINEARIZE) are timed with the dispatch's own start/stop timestamps (hipExtLaunchKernelGGL with
 * start/stop events: the same clock rocprofv3 reports); multi-kernel ids (SOLVE, GRID_BUILD, EXTRACT) are bracketed with
 * hipEventRecord on the context's stream. Durations / launch counts are read back with mlh_profile_get. Kernel ids: */
enum {
    MLH_K_KNN = 0,           /* correspondence kernel (exact 5-NN, surf + corner features in one launch) -- the roofline kernel */
    MLH_K_FIT = 1,           /* fit + gates + residual/Jacobian + J^T J reduction (+ fused GN solve in its last workgroup)     */
    MLH_K_LINEARIZE = 2,     /* re-linearisation on stored correspondences (LM iterations)                                     */
    MLH_K_SOLVE = 3,         /* stand-alone partial-sum reduction + degeneracy + 6x6 solve / LM step kernels                   */
    MLH_K_GRID_BUILD = 4,    /* local-map index build (all kernels of one build)                                               */
    MLH_K_EXTRACT = 5,       /* extractCloud (all kernels of one extraction)                                                   */
    MLH_K_ALLREDUCE = 6,     /* the RCCL all-reduce of the packed normal equations (N > 1)                                     */
    MLH_K_KNN_PRE = 7,       /* the correspondence kernel of iterations >= 1 of a deferred-finish Gauss-Newton solve: the same search (bounded by the previous
                                iteration's neighbours) behind the prologue that completes the previous iteration (sum of the tiles' records, 6x6 solve, Plus)          */
    MLH_K_KNN_FIRST = 8,     /* the correspondence kernel of iteration 0 of a CHAINED solve that completes its predecessor first (final_in_successor): the predecessor's last
                                iteration summed, solved and published, this frame's start pose chained from it, then the cold search                                        */
    MLH_K_COUNT = 9
};
/* kernel_mask: bit k enables the brackets of kernel id k (0 = profiling off, -1 = all) */
int mlh_profile_enable(mlh_ctx *ctx, int kernel_mask);
/* bracket only every n-th launch of a single-kernel id (an event pair costs host + queue time of its own); default 1 */
int mlh_profile_sample(mlh_ctx *ctx, int every_n);
int mlh_profile_reset(mlh_ctx *ctx);
int mlh_profile_get(mlh_ctx *ctx, int kernel_id, double *total_ms, long long *launches);

/* ---------------------------------------------------------------- (f3) ImageSegmenter::segmentCloud
 * replaces ImageSegmenter::segmentCloud(laser_cloud_in, laser_cloud_out, laser_cloud_outlier, scan_info)
 *   estimator/src/imageSegmenter/image_segmenter.hpp:139-393 (projectCloud :88-136, setParameter image_segmenter.cpp:18-63), called by
 *   Estimator::inputCloud (estimator.cpp:228, 258, 298, 317).
 * In: the raw, UNORDERED cloud of one LiDAR (HOST or DEVICE records: x y z at offset 0, f32 intensity at intensity_offset_bytes or -1).
 * Out: the ring-major cloud [x y z intensity + row] and ScanInfo::scan_start_ind_ / scan_end_ind_ (+5 / -6 insets), staged as the context's
 * scan exactly as mlh_scan_upload leaves it -- mlh_extract_run follows directly, the cloud never has to be assembled on the host -- and,
 * when the pointers are given, copied back: cloud_out (up to n x 4 floats), scan_start / scan_end (vertical_scans each), outlier_out
 * (laser_cloud_outlier: rows of 4 floats; the caller states the rows it has room for in outlier_capacity, writes are clamped to it and
 * *n_outlier always reports the rows the cloud HAS, so a short buffer is detected, not overrun. An upper bound that never truncates:
 * min(n, vertical_scans * ((horizon_scans + 4) / 5)) + 1 -- one row per infeasible-cluster pixel in a column that is a multiple of 5,
 * image_segmenter.hpp:366-378, plus the first output point, hpp:391). Projection, ground pairs and the final gather run on the GPU; the cluster search and
 * the outlier erasure are defined by their sequential order and run on the host inside this call (m-loam_amd/csrc/segment.hip says why).
 * The reference's undefined spots -- alpha used before it is set, erase with shifted positions, the 64-ring ground loop's row 64 -- behave as
 * INTEGRATION.md documents. vertical_scans 16, 32 or 64. */
typedef struct mlh_segment_params {
    int32_t vertical_scans;            /* N_SCANS */
    int32_t horizon_scans;             /* HORIZON_SCAN (horizon_scan) */
    int32_t min_cluster_size;          /* MIN_CLUSTER_SIZE */
    int32_t segment_valid_point_num;   /* SEGMENT_VALID_POINT_NUM */
    int32_t segment_valid_line_num;    /* SEGMENT_VALID_LINE_NUM */
    float segment_theta;               /* SEGMENT_THETA */
    double roi_range;                  /* ROI_RANGE */
    int32_t segment_flag;              /* ScanInfo::segment_flag_ (segment_cloud): 0 = project and reorder only, nothing is erased */
} mlh_segment_params;
void mlh_segment_params_default(mlh_segment_params *p);     /* config_realvehicle_hercules.yaml:7-13, 102 */
int mlh_segment_cloud(mlh_ctx *ctx, const void *points, int stride_bytes, int intensity_offset_bytes, int n, int mem, const mlh_segment_params *prm,
                      float *cloud_out, int32_t *n_out, int32_t *scan_start, int32_t *scan_end, float *outlier_out, int32_t outlier_capacity, int32_t *n_outlier);

/* ---------------------------------------------------------------- (a1-a3) FeatureExtract::extractCloud
 * replaces FeatureExtract::extractCloud(const PointICloud&, const ScanInfo&, cloudFeature&)
 *   estimator/src/featureExtract/feature_extract.cpp:118-297, declared feature_extract.hpp:74.
 * The cloud is ring-major; scan_start[r]/scan_end[r] are ScanInfo::scan_start_ind_/scan_end_ind_
 * (already inset by +5/-6, image_segmenter.hpp:385-387).
 *
 * mlh_scan_upload   stages one cloud (+ring table) in HBM             [HOST or DEV source]; intensity_offset_bytes (-1: none) is the
 *                   byte offset of the f32 intensity field the per-ring VoxelGrid averages with the coordinates (the tracker reads
 *                   the ring id from it, image_segmenter.hpp:128)
 * mlh_extract_run   curvature (cpp:133-142), per-sector sort + greedy labelling (cpp:152-265), index lists
 * mlh_extract_fetch copies results back: label[n] in {2,1,0,-1} (cloud_label), curvature[n], and the four index
 *                   lists in the reference's emission order: 0 corner_points_sharp, 1 corner_points_less_sharp,
 *                   2 surf_points_flat, 3 surf_points_less_flat BEFORE the per-ring VoxelGrid (positions with
 *                   label<=0, cpp:258-264). Any output pointer may be NULL.
 */
int mlh_scan_upload(mlh_ctx *ctx, const void *points, int stride_bytes, int intensity_offset_bytes, int n, const int *scan_start,
                    const int *scan_end, int n_rings, int mem);
/* mlh_scan_upload_ahead: the points of the scan a LATER mlh_scan_upload(.., MLH_MEM_HOST) will stage, sent to the device NOW on a copy stream of the context's own,
 * while the context's stream still works on the current scan (extraction, thinning, a solve): the copy engine beside the kernels, the upload off the frame's chain
 * (a replayed bag, a driver with the next sweep already in memory; with a live sensor there is nothing to send ahead). The mlh_scan_upload that names the SAME
 * points / stride_bytes / n packs from what arrived instead of copying; any other host upload in between drops it; a second call replaces the first (one scan
 * ahead). `points` must stay valid and unchanged until that mlh_scan_upload has returned. No reference counterpart: the reference's clouds never leave the host
 * (estimator.cpp:248-263 hands pcl clouds to extractCloud). */
int mlh_scan_upload_ahead(mlh_ctx *ctx, const void *points, int stride_bytes, int n);
int mlh_extract_run(mlh_ctx *ctx);
/* The order of EQUAL curvatures inside a sector. The reference sorts a sector's point indices with std::sort and compObject, which compares
 * cloudCurvature only (feature_extract.cpp:152-162), so among equal values the greedy walks meet the points in whatever order libstdc++'s
 * introsort leaves them.
 *   1 (default)  that order: a sector whose sorted curvatures show two equal values (or a NaN) is re-ordered ON THE DEVICE by the library's
 *                algorithm on the same initial arrangement (m-loam_amd/csrc/stdsort_dev.hpp: the same comparison sequence, hence the same
 *                permutation); sectors with distinct curvatures -- every sector of ordinary float data -- have one sorted order and pay nothing;
 *   0            (curvature, index) ascending, NaN curvatures last by index: a total order that does not depend on a standard library. */
int mlh_set_extract_tie_order(mlh_ctx *ctx, int mode);
int mlh_extract_fetch(mlh_ctx *ctx, int32_t *label, float *curvature, int32_t *picked, int32_t *idx_out[4], int32_t n_out[4]);
/* (a3) the per-ring pcl::VoxelGrid(leaf = 0.2 m) extractCloud applies to the less-flat points of every ring
 * (feature_extract.cpp:266-271): run after mlh_extract_run; fetch returns "surf_points_less_flat" as the reference emits it
 * (ring asc, voxel index asc; x y z intensity centroids). PCL sums a voxel's members in the order an unstable std::sort (comparator on
 * the voxel index only) left them; with the context's default member order (mlh_set_voxel_member_order, below) the sums run along that same
 * permutation -- one device std::sort per ring -- and the centroids are the reference's bit for bit; with mode 0 the members are summed in
 * scan order (equal up to f32 rounding of the sum, ~0.1 ms per scan pair cheaper). */
int mlh_extract_voxel_run(mlh_ctx *ctx, float leaf);
int mlh_extract_fetch_voxel(mlh_ctx *ctx, float *xyzi_out, int32_t *n_out);

/* (a18) per-point uncertainty of downsampleCurrentScan (lidar_mapper_keyframe.cpp:375-418): for every point (intensity =
 * LiDAR index n) Sigma_p = evalPointUncertainty(pose_ext[n]^-1 * p, pose_ext[n])  (associate_uct.hpp:196-215) with
 * cov_input = diag(ext_covs[n] (6x6), cov_measurement (3x3)). Outputs the f32 cov_vec[6] of PointXYZIWithCov per point and
 * keep[i] = 0 where trace > trace_threshold (TRACE_THRESHOLD_MAPPING; <= 0 keeps everything); the caller appends the kept
 * points in order, as the reference's loop does. */
int mlh_point_uncertainty(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                          const double *ext_poses, const double *ext_covs, int n_lidar, const double cov_measurement[9],
                          double trace_threshold, float *cov_vec_out, int32_t *keep_out);

/* The staged feature set of `kind` (mlh_features_set / mlh_downsample_current_scan) copied from context `src` to context `dst` on the same GPU, device to
 * device. The reference runs extraction / odometry and mapping as separate nodes joined by ROS messages (estimator.cpp publishes the feature clouds, lidar_mapper_
 * keyframe.cpp:162-190 queues them); two contexts driven by two host threads are the same arrangement on one GPU -- the estimator-side context extracts and thins
 * frame k + 1 while the mapper-side context solves frame k -- and this call is the message. Meant to be called from the thread that drives dst; of src it only needs the feature set to stay as it is until the call returns
 * (then src may restage at once). */
int mlh_features_copy(mlh_ctx *dst, mlh_ctx *src, int kind);
/* (f2) downsampleCurrentScan for one feature kind (lidar_mapper_keyframe.cpp:356-421), device-resident: VoxelGridCovarianceMLOAM<PointI>
 * at `leaf` (plain branch), then per thinned point (intensity = LiDAR index n) Sigma = evalPointUncertainty(pose_ext[n]^-1 * p,
 * pose_ext[n]) and the trace gate (with_ua only; Sigma = 0 and no gate otherwise). The kept points BECOME the kind's feature set
 * (as if handed to mlh_features_set with their covariance), so extraction output can flow into the solver without touching the
 * host; features_out (HOST, n x 11 floats [x y z i cov6 trace], may be NULL) additionally returns them. */
int mlh_downsample_current_scan(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                                float leaf, const double *ext_poses, const double *ext_covs, int n_lidar, const double cov_measurement[9],
                                int with_ua, double trace_threshold, float *features_out, int32_t *n_features);
/* The same for both feature kinds in one call (the mapper thins the surf and the corner cloud back to back, cpp:359-368). When both
 * clouds are the context's fused clouds (mlh_fused_cloud) they run through ONE thinning pipeline -- half the dependent launches, one
 * host round trip; other inputs take the two single calls. Records of both clouds share stride / intensity offset / memory kind;
 * afterwards both feature sets are staged exactly as two single calls leave them. */
int mlh_downsample_current_scan_pair(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                                     int intensity_offset_bytes, int mem, float leaf_surf, float leaf_corner, const double *ext_poses,
                                     const double *ext_covs, int n_lidar, const double cov_measurement[9], int with_ua, double trace_threshold,
                                     int32_t *n_surf_features, int32_t *n_corner_features);

/* ---------------------------------------------------------------- (f1) covariance-aware voxel thinning
 * replaces pcl::VoxelGridCovarianceMLOAM<PointT>::filter (mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp:68-457),
 * the filter that produces the voxel-thinned local map (lidar_mapper_keyframe.cpp:343-347) and thins the scan features
 * (cpp:359-368). cov_offset_bytes >= 0 selects the covariance branch (PointXYZIWithCov, :296-333): members with |trace| >=
 * trace_threshold are dropped, w = trace_threshold - trace, mu = sum w p / sum w, cov = sum w^2 cov_i / (sum w)^2, intensity of the
 * heaviest member, trace recomputed; otherwise the plain branch (PointXYZI, :392-420): xyz mean, intensity of the last member.
 * Output (HOST, capacity n records) uses the input's record layout and is ordered by voxel index like the reference's.
 * MEMBER ORDER. The reference groups a voxel's members with std::sort and a comparator that sees the voxel index only (impl.hpp:227), i.e.
 * inside a voxel they come in whatever order libstdc++'s introsort leaves. That order decides "the last member" of the plain branch --
 * for a fused multi-LiDAR cloud the LiDAR id downsampleCurrentScan propagates the uncertainty through (lidar_mapper_keyframe.cpp:377) --,
 * the first-heaviest member of the covariance branch on equal weights, and the association of every f32 sum. Every voxel filter of the
 * context (mlh_voxel_filter, mlh_voxel_grid, mlh_downsample_current_scan, ..._pair; and extractCloud's per-ring grid, mlh_extract_voxel_run, where
 * modes 1 and 2 both mean the device path) follows mlh_set_voxel_member_order(ctx, mode):
 *   1 (default)  the reference's order, produced ON THE DEVICE: libstdc++'s std::sort (introsort: median-of-three pivot at the same
 *                positions, the same unguarded Hoare partition, the same depth budget and heap-sort fallback, the same final insertion
 *                pass) restated data-parallel -- one launch per recursion depth, a range's partition from rank tables -- so the
 *                permutation equals std::sort's element for element and the filters' results equal the reference's bit for bit;
 *   2            the same order through a host pass: the points' slots come back (pinned), the platform's OWN std::sort runs on the
 *                same sequence, the member lists go back (+1.2 ms for a frame's clouds). For a build against a standard library
 *                whose std::sort is not the algorithm mode 1 restates (GCC's libstdc++, unchanged in this respect since 4.x);
 *                mlh_std_sort_permutation lets an integrator compare the two on any key sequence;
 *   0            members in ascending point index (what a stable sort would give), device only and a few launches cheaper. Voxel set,
 *                output order and counts stay identical to the reference's and sums agree to f32 rounding, but the surviving id of a
 *                voxel that MIXES ids can differ (on a two-LiDAR frame: ~31 % of the 0.4 m surf voxels, ~5 % of the 0.2 m corner
 *                voxels; with uncertainty weighting that moved the frame's pose by 5 mm in scripts/framebench.py). A single-LiDAR
 *                cloud has no mixed voxels. */
int mlh_set_voxel_member_order(mlh_ctx *ctx, int mode);
/* Debug aid, no reference counterpart. With MLH_CHECK_LAUNCH=1 in the environment every kernel launch of the library is followed by hipGetLastError(): a bad
 * launch configuration is reported by the entry point that made it, as MLH_ERR_HIP with "launch of <kernel>: <error>" in mlh_last_error, instead of surfacing at
 * a later synchronisation under another call's name. mlh_debug_bad_launch makes one launch with an impossible configuration (4096 threads per workgroup) so that
 * the mechanism itself can be tested: MLH_ERR_HIP naming debug_noop_kernel under MLH_CHECK_LAUNCH=1, MLH_OK otherwise. */
int mlh_debug_bad_launch(mlh_ctx *ctx);
/* perm_out[0..n) (HOST) <- the permutation of 0..n-1 that two std::sort calls -- over [0, n0) and [n0, n), comparator on keys[] (HOST,
 * non-negative) only -- leave: mode 1 = the device restatement the voxel filters use, mode 2 = the platform's std::sort on the host. */
int mlh_std_sort_permutation(mlh_ctx *ctx, const int32_t *keys, int n0, int n, int32_t *perm_out, int mode);
/* mlh_voxel_filter: `mem` describes BOTH buffers: MLH_MEM_DEVICE takes device records and leaves the result in device memory (`out`), so a map
 * assembled with mlh_cloud_uct_associate_to_map can be thinned and handed to mlh_map_set without leaving HBM. */
int mlh_voxel_filter(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int cov_offset_bytes,
                     int trace_offset_bytes, float leaf, float trace_threshold, void *out, int32_t *n_out, int mem);

/* ---------------------------------------------------------------- (a19) odometry-side three-block factors
 * replaces the per-feature LidarPureOdomPlaneNormFactor / LidarPureOdomEdgeFactor objects Estimator::optimizeMap hands to Ceres
 * (estimator.cpp:700-780) and their Evaluate (lidar_pure_odom_factor.hpp:38-102, 209-282): one residual + three 1x7 Jacobians
 * (pivot pose, window pose i, extrinsic n), point moved with T_pivot^-1 T_i T_ext.
 * mlh_pure_odom_set stages the factor table once per optimisation: type[i] 0 = plane ('s'), 1 = edge ('c'); points n x 3;
 * coeffs n x 6 (plane: [n, d, -, -], edge: the two line points); sqrt_info may be NULL (= 1.0, as the reference constructs them);
 * frame_idx / ext_idx select rows of the pose arrays given to mlh_pure_odom_evaluate.
 * mlh_pure_odom_evaluate: residuals n; jacobians n x 21 row-major [pivot 1x7 | frame 1x7 | extrinsic 1x7] or NULL. The columns
 * are the reference's expressions term by term (two of them are not exact derivatives; see m-loam_amd/csrc/odom.hip). */
int mlh_pure_odom_set(mlh_ctx *ctx, int n, const int32_t *type, const double *points, const double *coeffs, const double *sqrt_info,
                      const int32_t *frame_idx, const int32_t *ext_idx);
int mlh_pure_odom_evaluate(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                           double *residuals, double *jacobians);
/* The same factor table built WITHOUT the host: mlh_pure_odom_begin starts an empty table; every mlh_pure_odom_add_matches matches the
 * staged feature set of `kind` (mlh_features_set / mlh_downsample_current_scan, one LiDAR's features in that LiDAR's frame) against the
 * resident map -- the window's local map in the pivot frame, estimator.cpp:1160-1268 -- at rel_pose = T_pivot^-1 T_frame T_ext
 * (FeatureExtract::matchCornerFromMap / matchSurfFromMap as Estimator::optimizeMap calls them, estimator.cpp:700-780; k_neigh 5 or 10,
 * MLH_FLAG_CHECK_FOV honoured) and appends one factor per valid correspondence: type = kind, point = the feature, coefficients = the
 * fitted plane / line, s = 1.0, blocks (frame_idx, ext_idx). Nothing is copied back; the number of factors is returned by
 * mlh_pure_odom_normal_eq (n_residuals). Such a table serves mlh_pure_odom_normal_eq only (its regions are padded to whole tiles);
 * mlh_pure_odom_evaluate needs a host-staged one. */
int mlh_pure_odom_begin(mlh_ctx *ctx);
int mlh_pure_odom_add_matches(mlh_ctx *ctx, int kind, const double rel_pose[7], int k_neigh, uint32_t flags, float min_match_sq_dis,
                              float min_plane_dis, int frame_idx, int ext_idx);
/* The same with the ODOMETRY's good-feature selection in front of the append -- Estimator::goodFeatureMatching with Estimator::evaluateFeatJacobian
 * (estimator.cpp:1273-1517), which buildLocalMap runs on every (frame, LiDAR) group with gf_ratio = ODOM_GF_RATIO (estimator.cpp:1241-1263; 0.8 in every shipped
 * configuration): all features of the group are matched on the GPU at rel_pose (= the Pose of T_pivot^-1 T_i T_ext, as the caller builds it), the row the
 * selection scores -- a surf feature's LidarPureOdomPlaneNormFactor(point, coeffs, 1.0) frame block at (pivot, pose_i, ext); (1 0 0 0 0 0) for a corner feature,
 * cpp:1339-1342 -- is evaluated for every matched feature in one launch, and the sequential draw loop runs on the host with the reference's draw sequence
 * (std::mt19937(seed) + uniform_int_distribution; MAX_RANDOM_QUEUE_TIME = 10, estimator.h:63; ten failed draws restart the round instead of ending the selection,
 * as the reference's loop does; its 7 ms wall-clock cut-off is not applied -- the loop ends when nothing is left to draw). gf_ratio == 1.0: every matched
 * feature, no loop. gf_ratio is a float, as ODOM_GF_RATIO is: the number of features asked for is size_t(n * (double)gf_ratio).
 * Only the selected correspondences are appended as factors. sel_out (nullable, capacity = the staged feature count) / n_sel (nullable): the selected feature
 * indices in selection order. One GPU (the loop needs the whole group's features). */
int mlh_pure_odom_add_matches_gf(mlh_ctx *ctx, int kind, const double rel_pose[7], const double pivot[7], const double pose_i[7], const double ext[7], int k_neigh,
                                 uint32_t flags, float min_match_sq_dis, float min_plane_dis, int frame_idx, int ext_idx, float gf_ratio, uint64_t seed,
                                 int32_t *sel_out, int32_t *n_sel);
/* The normal equations of the COUPLED window problem those factors form (BASELINE config 4; Estimator::optimizeMap, estimator.cpp:687-848):
 * local parameters in para_ids order [pivot | frames 0..n_frames) | extrinsics 0..n_ext)], 6 each (PoseLocalParameterization::ComputeJacobian
 * = [I6; 0]), D = 6 (1 + n_frames + n_ext). What Estimator::evalResidual gets from problem.Evaluate (estimator.cpp:1577-1595) -- rows
 * loss-corrected with HuberLoss(huber_delta) (1.0 at estimator.cpp:602; <= 0: no loss), every listed block variable -- reduced on the device:
 * JtJ D x D row-major (symmetric, both triangles filled), Jtr D, cost = sum rho(r^2)/2, n_residuals. evalDegenracy (estimator.cpp:1598-1680)
 * reads its diagonal 6x6 blocks (mlh_eval_degeneracy on each); with several GPUs (factors split over the ranks) the record is summed with
 * mlh_allreduce_f64 -- D (D + 1) / 2 + D + 2 = 326 doubles for the 24-dimensional hercules window. Deterministic (no atomics). */
int mlh_pure_odom_normal_eq(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                            double huber_delta, double *JtJ, double *Jtr, double *cost, int32_t *n_residuals);
/* The coupled window problem solved on the device: n_iters Gauss-Newton iterations on the staged (or device-built) factor table -- per iteration the
 * normal equations above, then ONE workgroup gathers the rows / columns of the blocks that are not held constant, factorises them (Cholesky in LDS; + 1e-6 I
 * and a second attempt when not positive definite) and applies PoseLocalParameterization::Plus block by block; three launches per iteration, the poses stay in
 * HBM until the call returns. replaces the ceres::Solve of Estimator::optimizeMap on the LidarPureOdom / LidarOnlineCalib factors (estimator.cpp:593-680,
 * 852-861) for the part of that problem built here (no marginalisation prior -- SURVEY section 2 #13 -- and Gauss-Newton steps, not Ceres' trust region).
 *   const_block_mask  bit b set = block b of [pivot | frames 0.. | extrinsics 0..] is held constant (estimator.cpp:636 para_pose_[0], :642 the reference LiDAR's
 *                     extrinsic; 1u | 1u << (1 + n_frames) for the reference's choice)
 *   V_update          NULL (identity) or 36 doubles per block, row-major: what Estimator::evalDegenracy (estimator.cpp:1598-1680; facade evalDegenracy) left in
 *                     every block's PoseLocalParameterization -- a projector for a degenerate pose block, zero for an extrinsic that is not to be updated
 *   frames / exts     in: the linearisation point, out: the result.  cost / n_residuals: of the LAST linearisation (before the final update).
 *   status            0 solved, 1 some iteration needed the + 1e-6 I, 2 some iteration's system was not positive definite even then (that update was skipped)
 * Under a communicator (mlh_comm_init / mlh_p2p_comm_init) the call returns MLH_ERR_UNSUPPORTED: it factorises this rank's sums, and ranks holding different
 * parts of the factor table would apply different updates. Sharded callers all-reduce mlh_pure_odom_normal_eq's H / g (mlh_allreduce_f64) and solve on the host. */
int mlh_pure_odom_gn_solve(mlh_ctx *ctx, const double pivot[7], double *frames, int n_frames, double *exts, int n_ext, double huber_delta, int n_iters,
                           uint32_t const_block_mask, const double *V_update, double *cost, int32_t *n_residuals, int32_t *status);

/* (f1) cloudUCTAssociateToMap (lidar_mapper_keyframe.cpp:1116-1158): moves one keyframe's feature cloud into the map frame while
 * building the local map (extractSurroundingKeyFrames, cpp:254-354). Per point (intensity = LiDAR index n):
 *   with_ua: point_sel = pose_ext[n]^-1 * p; Sigma = evalPointUncertainty(point_sel, pose_global (+) pose_ext[n]) where the
 *            compound pose and its 6x6 covariance come from compoundPoseWithCov method 2 (associate_uct.hpp:90-147);
 *            points with trace(Sigma) > trace_threshold are dropped;
 *   always:  the record is copied, xyz <- pose_global * p (f64 math, f32 store), cov_vec <- Sigma (zeros without with_ua),
 *            cov_trace <- trace.
 * Output keeps the input order and record layout; `mem` describes both buffers (see mlh_voxel_filter).
 * Covariance layout: 6x6 row-major over [translation, rotation], as Pose::cov_. */
int mlh_cloud_uct_associate_to_map(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int cov_offset_bytes,
                                   int trace_offset_bytes, const double pose_global[7], const double cov_global[36], const double *ext_poses,
                                   const double *ext_covs, int n_lidar, const double cov_measurement[9], int with_ua, double trace_threshold,
                                   void *out, int32_t *n_out, int mem);
/* compoundPoseWithCov(pose_1, pose_2, pose_cp, method = 2) (associate_uct.hpp:90-147), host arithmetic, f64 */
int mlh_compound_pose_with_cov(const double pose_1[7], const double cov_1[36], const double pose_2[7], const double cov_2[36],
                               double pose_cp[7], double cov_cp[36]);

/* ---------------------------------------------------------------- (a5) local map index
 * replaces pcl::KdTreeFLANN<PointT>::setInputCloud(cloud) as used at
 *   estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:433-434 (and estimator.cpp:1095-1109, 1230-1233).
 * Builds a dense cell grid (cell edge just above sqrt(min_match_sq_dis), so the 27-cell neighbourhood is exact for
 * the acceptance test "5th neighbour sq-dist < min_match_sq_dis", feature_extract.hpp:667/814) over the cloud
 * and keeps the points cell-sorted in HBM as float4 {x, y, z, original index}.
 * mlh_map_rebuild re-runs the index build on the resident cloud(s) (the reference rebuilds every frame); kind may be
 * MLH_ALL_KINDS to rebuild both maps in one set of fused launches.
 */
int mlh_map_set(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, float min_match_sq_dis, int mem);
/* the two setInputCloud calls of a mapper frame (lidar_mapper_keyframe.cpp:433-434) in one set of launches and one host hand-shake.
 * The grid box of a kind is laid out with a margin when it is first computed; while later clouds keep fitting it (the local map is a
 * sliding keyframe window) a call costs pack + index build only, and returns as soon as the pack pass has reported that the clouds fit (the index build is still
 * running on the context's stream then; every later call is ordered behind it) -- a cloud that has outgrown the box is detected on the
 * device and takes the bounds pass again. Results never depend on which way a call went (the grid is an acceleration structure only). */
int mlh_map_set_pair(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                     float min_match_sq_dis, int mem);
/* The same, for the NEXT frame while a solve submitted with mlh_gn_solve_begin is still running: the maps are double-buffered, this call stages and indexes
 * into the set the solve does not read, on a second stream, returns when that index is complete and makes the set current -- launches enqueued afterwards (the
 * next mlh_gn_solve_begin) read it; the solve in flight keeps the old one (one solve in flight at this point: collect the older one first). Called a second
 * time beside the SAME uncollected solve, its target is the set that solve reads: the call then waits for the main stream to run dry before it rewrites the set
 * (correct, no longer overlapped).
 * The GPU then builds the next index (a chain of small launches) in the shadow of the current frame's iterations. What the caller must know: in the reference the
 * local map of frame k + 1 depends on frame k's optimised pose in one place -- extractSurroundingKeyFrames (lidar_mapper_keyframe.cpp:254-354) picks the keyframes
 * within a radius of the PREDICTED pose of frame k + 1, which carries frame k's map-to-odometry correction (cpp:145-160). A caller that stages this early selects
 * with the correction of frame k - 1 and re-checks the selection when frame k's pose arrives (a radius test over the keyframe positions on the host); the set
 * changes only when a keyframe sits within the change of the correction (sub-centimetre) of the radius, and then the frame is staged and solved again
 * synchronously. INTEGRATION.md section 2 spells the loop out.
 * Without a solve in flight and with device-resident clouds the same call overlaps the index build with whatever the main stream is doing that reads no map --
 * a frame's own front end: call it after the frame's upload / extraction / fusion have been enqueued and before the first call that waits for them
 * (mlh_fused_cloud); the host then waits for the build instead of for the front end, and scan2MapOptimization finds the index ready (the local map is made of
 * earlier keyframes: it does not depend on the scan being processed). Without a solve in flight and with host-resident clouds: mlh_map_set_pair. */
int mlh_map_set_pair_overlapped(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                                float min_match_sq_dis, int mem);
int mlh_map_rebuild(mlh_ctx *ctx, int kind);
/* diagnostics of the resident index (no reference counterpart: pcl::KdTreeFLANN exposes nothing of the kind): points, non-empty grid
 * cells, the population of the cell an average map point lives in (sum of squared cell populations / n), and the lanes per query the
 * correspondence kernel will use for this kind with the staged feature sets (8 or 16; chosen from the launch size and that density --
 * a tuning decision, results do not depend on it). Any output pointer may be null. */
int mlh_map_info(mlh_ctx *ctx, int kind, int32_t *n_points, int32_t *occupied_cells, double *mean_cell_population, int32_t *knn_lanes);
/* k-NN against the resident map (pcl::KdTreeFLANN::nearestKSearch role; feature_extract.hpp:666, 813):
 * queries are xyz triples (HOST), outputs idx[nq*k] (original map indices, ascending distance, ties by index) and
 * sqdist[nq*k]; slots beyond the number of points found inside the 27-cell neighbourhood are -1 / +inf. k = 5.
 * EXACT ONLY INSIDE THE ACCEPTANCE RADIUS: the search covers the 27 cells around the query (cell edge 1.001 * sqrt(min_match_sq_dis)),
 * so indices and distances equal a kd-tree's for every neighbour with sqdist < min_match_sq_dis -- all the mapper looks at
 * (sq_dis[k-1] < MIN_MATCH_SQ_DIS, hpp:667/814). A returned neighbour with sqdist >= min_match_sq_dis may be farther than a map
 * point just outside the 27 cells; a caller that needs exact neighbours beyond that radius must stage the map with a larger
 * min_match_sq_dis. */
int mlh_knn(mlh_ctx *ctx, int kind, const float *queries_xyz, int nq, int k, int32_t *idx, float *sqdist);

/* ---------------------------------------------------------------- scan features (the cloud_data / laser_cloud argument)
 * stages a feature cloud (PointIWithCov role). cov_offset_bytes: byte offset of cov_vec[6] (cxx cxy cxz cyy cyz czz, f32)
 * inside a record, or -1 when the records carry no covariance (then trace = 0 -> weight 1, lidar_map_factor.hpp:41).
 * intensity_offset_bytes: byte offset of the f32 intensity (= LiDAR index, visualization.cpp:48), or -1.
 */
int mlh_features_set(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes,
                     int cov_offset_bytes, int mem);

/* Pose blocks (a19, BASELINE config 4: online extrinsic calibration). The estimator's calibration problem
 * (estimator.cpp:656-780, buildCalibMap :1067-1157) gives every LiDAR its own 7-parameter block: the reference LiDAR's features
 * constrain the body pose (LidarPureOdom* blocks with the pivot and the reference extrinsic held constant), LiDAR n's features
 * constrain its extrinsic alone (LidarOnlineCalib{PlaneNorm,Edge}Factor, lidar_online_calib_factor.hpp:24-165 -- the same
 * residual/Jacobian form as the map factors with sqrt_info = 1). Stage the feature cloud of block 0, 1, ... in ascending order;
 * every block starts on a 256-feature boundary in HBM so that a workgroup never straddles two pose blocks. */
int mlh_features_set_block(mlh_ctx *ctx, int kind, int block, const void *points, int stride_bytes, int n, int cov_offset_bytes, int mem);

/* ---------------------------------------------------------------- (a4, a6-a11, a15) match + linearise + reduce
 * replaces, for every staged feature of `kind`, at pose `pose`:
 *   pointAssociateToMap                         estimator/src/utility/utility.h:103-117
 *   FeatureExtract::match{Surf,Corner}PointFromMap / match{Surf,Corner}FromMap   feature_extract.hpp:379-883
 *   LidarMapPlaneNormFactor / LidarMapEdgeFactor ctor + Evaluate                  lidar_map_factor.hpp:28-71, 132-174
 *   ActiveFeatureSelection::evaluateFeatJacobianMatching                           lidar_mapper.h:130-174
 *   the J^T J / J^T r accumulation of evalHessian / Ceres' evaluator              lidar_mapper_keyframe.cpp:575-581, 1160-1169
 * Dense outputs (HOST, nullable), in feature order:
 *   valid[m]  1 when the feature found a correspondence (the bool return of match*PointFromMap)
 *   coeffs[m*6]  PointPlaneFeature::coeffs_ : surf = (n_hat, d, 0, 0); corner = (X1, X2)
 *   r[m], J[m*6] the factor's residual and first 6 Jacobian columns (weighted by sqrt_info, NOT loss-corrected)
 * Reduced outputs (HOST, nullable): JtJ[36] row-major, Jtr[6], cost (sum 0.5*rho), n_valid -- rows are Huber-corrected
 * as Ceres' ResidualBlock::Evaluate does (huber_delta <= 0 or MLH_FLAG_NO_LOSS: trivial loss).
 */
int mlh_match_linearize(mlh_ctx *ctx, int kind, const double pose[7], int k_neigh, uint32_t flags,
                        float min_match_sq_dis, float min_plane_dis, double huber_delta, double cov_measurement_trace,
                        uint8_t *valid, double *coeffs, double *r, double *J,
                        double *JtJ, double *Jtr, double *cost, int32_t *n_valid);

/* Re-linearise the correspondences found by the last mlh_match_linearize of `kind` at a new pose
 * (what ceres::Solve does on every LM iteration: CostFunction::Evaluate on fixed factors). Same outputs. */
int mlh_linearize(mlh_ctx *ctx, int kind, const double pose[7], uint32_t flags, double huber_delta, double cov_measurement_trace,
                  double *r, double *J, double *JtJ, double *Jtr, double *cost, int32_t *n_valid);

/* the correspondences of `kind` currently live on the device (after mlh_match_linearize, or -- only the selected ones -- after
 * mlh_good_feature_matching): valid[m], coeffs[m x 6] (PointPlaneFeature::coeffs_: plane n, d / line X1, X2; zeros where invalid) */
int mlh_match_coeffs(mlh_ctx *ctx, int kind, uint8_t *valid, double *coeffs, int32_t *n_valid);

/* ---------------------------------------------------------------- (a13) good-feature selection
 * replaces ActiveFeatureSelection::goodFeatureMatching (estimator/src/lidarMapper/lidar_mapper.h:229-573).
 * ALL features of `kind` are matched and their weighted, un-corrected 1x6 Jacobians evaluated on the GPU in one pass
 * (what match*PointFromMap + evaluateFeatJacobianMatching produce one feature at a time, lidar_mapper.h:130-174, 483-521); the
 * inherently sequential draw loops -- rnd, stochastic-greedy logdet (gd_fix, gd_float) -- then run on the host
 * over those rows with the reference's draw sequence (std::mt19937 + uniform_int_distribution, common/random_generator.hpp:53;
 * the MAX_FEATURE_SELECT_TIME wall-clock cut-off is not applied); fps draws one number (its starting point) and its
 * farthest-point arg-max loop runs on the device for up to 16384 features (same f32 arithmetic, lowest index among equal
 * distances; a round re-measures only the buckets of points the new pick can change -- the same picks, 1 us per round; inside
 * mlh_scan2map the two kinds' loops run side by side), the host replaying selection list and information matrix along the visiting order. sub_mat_H must come in as the reference initialises it
 * (1e-6 * I, cpp:505/520) and returns H + sum j^T j of the selected rows. On return only the selected features stay valid
 * on the device, so mlh_linearize / the LM of mlh_scan2map see exactly the residual blocks the reference would add.
 * fps asked for more features than match (gf_ratio * n above the number of matching features): the reference's loop has no "every point visited" exit
 * (lidar_mapper.h:391-399), so once everything has been visited it matches feature 1 again on every further round and -- when that feature matches -- appends it,
 * with its J^T J, until the count is reached. Reproduced: sel_out then ends in repeats of index 1, sub_mat_H holds its outer product as many times, and on the
 * device the feature weighs as that many residual blocks. (When feature 1 does not match, the reference spins into its 20 ms cut-off and returns what it has: the
 * same list, without the wait.) */
enum { MLH_GF_WO = 0, MLH_GF_RND = 1, MLH_GF_FPS = 2, MLH_GF_GD_FIX = 3, MLH_GF_GD_FLOAT = 4 };
int mlh_good_feature_matching(mlh_ctx *ctx, int kind, const double pose[7], int gf_method, double gf_ratio, uint64_t seed,
                              float min_match_sq_dis, float min_plane_dis, int32_t *sel_idx, int32_t *n_sel,
                              double sub_mat_H[36], uint8_t *matched);

/* ---------------------------------------------------------------- (a16, a17) device-resident solvers
 * Common options. */
typedef struct mlh_solver_opts {
    float min_match_sq_dis;          /* MIN_MATCH_SQ_DIS, parameters.cpp:232 */
    float min_plane_dis;             /* MIN_PLANE_DIS,    parameters.cpp:233 */
    double huber_delta;              /* ceres::HuberLoss(0.1), lidar_mapper_keyframe.cpp:443 */
    double map_eig_thre;             /* MAP_EIG_THRE, evalDegenracy, lidar_mapper_keyframe.cpp:1178-1189 */
    double cov_measurement_trace;    /* trace(COV_MEASUREMENT) used when with_ua is off (cpp:543-544) */
    uint32_t flags;                  /* MLH_FLAG_WITH_UA | MLH_FLAG_CHECK_FOV. CHECK_FOV applies to mlh_gn_solve* (estimator.cpp:1142, 1149 pass it); the mlh_scan2map*
                                      * entry points ignore it, as scan2MapOptimization does: its matching goes through goodFeatureMatching, which hard-codes
                                      * n_neigh = 5 and CHECK_FOV = false (lidar_mapper.h:256-283) */
    int max_outer;                   /* max_iter = 2, cpp:439 */
    int max_lm_iterations;           /* options.max_num_iterations = 30, cpp:590 */
    int gf_method;                   /* FLAGS_gf_method: MLH_GF_* (lidar_mapper.h:89); mlh_scan2map only */
    double gf_ratio;                 /* gf_ratio_cur (cpp:474-492) */
    uint64_t gf_seed;                /* the reference seeds its mt19937 from std::random_device; fixed here */
} mlh_solver_opts;
void mlh_solver_opts_default(mlh_solver_opts *o);

/* per-iteration record written by the solvers (HOST array provided by the caller) */
typedef struct mlh_iter_stat {
    int32_t n_surf, n_corner;        /* matched features of each kind */
    int32_t is_degenerate;
    int32_t lm_iterations;           /* scan2map only */
    int32_t successful_steps;        /* scan2map only */
    int32_t termination;             /* scan2map only: 0 max-iters 1 gradient 2 parameter 3 function 4 failure */
    double cost;                     /* cost at the linearisation point (GN) / initial cost (scan2map) */
    double final_cost;               /* scan2map only */
    double eigval[6];
    double H[36];                    /* J^T J at the start of the iteration (evalHessian) */
    double g[6];
    double pose_after[7];
} mlh_iter_stat;

/* n_iters Gauss-Newton iterations, entirely on the device (no host round trip inside): per iteration re-match both
 * kinds at the current pose, linearise with Huber correction, reduce J^T J / J^T r, evalDegenracy
 * (lidar_mapper_keyframe.cpp:1172-1204), solve H d = -g, pose <- PoseLocalParameterization::Plus(pose, V_update d)
 * (pose_local_parameterization.cpp:26-45). This is BASELINE.json's "GN iteration". stats may be NULL. */
int mlh_gn_solve(mlh_ctx *ctx, double pose_inout[7], int n_iters, const mlh_solver_opts *opts, mlh_iter_stat *stats);
/* How the iterations of mlh_gn_solve / mlh_gn_solve_begin* are laid out on the device. Neither switch changes a result (same arithmetic in the same order; the
 * tests compare the variants bit for bit); they exist for A/B measurements and as the reference forms in the tests. Environment at mlh_create: MLH_GN_DEFER,
 * MLH_KNN_WARM (0 / 1).
 *   deferred_finish 1 (default): on one GPU, without per-iteration statistics and for frame-sized launches (at most 160 fit tiles = 40 960 feature slots; larger
 *       launches keep the classic form, where the redundant sums would cost more than the serial tail they replace), every iteration but the last leaves the J^T J / J^T r records of its tiles in HBM and
 *       the NEXT iteration's correspondence launch starts by summing them, solving and applying Plus in every workgroup redundantly (the kernel boundary is the only
 *       synchronisation); 0: the fit kernel's last-arriving workgroup finishes every iteration (evalHessian + evalDegenracy + solve + Plus,
 *       lidar_mapper_keyframe.cpp:575-596, 1160-1204) while the other compute units wait.
 *   final_in_successor 1 (default; needs deferred_finish): a solve submitted with mlh_gn_solve_begin* leaves its LAST iteration as records too. The next
 *       mlh_gn_solve_begin_chained completes it in its first correspondence launch -- sum, solve, Plus, the pose published to that solve's host record and stored as the
 *       state's pose from there, then this frame's chained start pose (lidar_mapper_keyframe.cpp:145-160) computed in the same prologue: neither the serial finish nor
 *       the chain launch stand between two frames. mlh_gn_solve_end, or any other solver call, completes a solve no successor took with a one-workgroup launch.
 *       0: the last iteration finishes and publishes in its own fit launch. Environment: MLH_GN_FINAL_DEFER.
 *   knn_warm_start 1 (default): iterations >= 1 bound the 5-NN search of a feature by the distances from its new position to the five neighbours the previous
 *       iteration found (an upper bound of the fifth-neighbour distance: the search stays exact, feature_extract.hpp:666/813); 0: every iteration searches cold. */
int mlh_set_gn_schedule(mlh_ctx *ctx, int deferred_finish, int knn_warm_start, int final_in_successor);
/* The same solve submitted and collected separately (one GPU, or several ranks joined by the mailbox communicator -- not under RCCL; no statistics): _begin enqueues the n_iters iterations and returns at once, _end waits for the
 * pose. Between the two the caller may stage the NEXT frame's maps (mlh_map_set_pair): those launches queue up behind the solve on the context's stream, so the
 * GPU does not idle through the host's turn-around at the frame boundary (bench.py submits its frames this way). At most two solves in flight per context
 * (frame k + 1 may be submitted before frame k's pose is collected); mlh_gn_solve_end returns them in submission order. */
int mlh_gn_solve_begin(mlh_ctx *ctx, const double pose_in[7], int n_iters, const mlh_solver_opts *opts);
/* Frame k + 1 submitted while frame k is still being solved: its start pose is the reference's -- transformUpdate() with frame k's RESULT and odometry pose, then
 * transformAssociateToMap() with frame k + 1's odometry pose (lidar_mapper_keyframe.cpp:145-160: pose_wmap_wodom = pose_wmap_curr * pose_wodom_curr.inverse();
 * pose_wmap_curr = pose_wmap_wodom * pose_wodom_curr; Pose::operator* and Pose::inverse as pose.cpp:99-113) -- computed ON THE DEVICE from the pose the previous
 * solve leaves there, by a one-lane launch in front of the iterations. wodom_prev / wodom_cur: [t, q] of the odometry poses of frame k and frame k + 1. The data
 * dependency between consecutive frames stays where the data is; the host never needs frame k's pose to submit frame k + 1. */
int mlh_gn_solve_begin_chained(mlh_ctx *ctx, const double wodom_prev[7], const double wodom_cur[7], int n_iters, const mlh_solver_opts *opts);
int mlh_gn_solve_end(mlh_ctx *ctx, double pose_out[7]);

/* Per-block options of mlh_gn_solve_blocks. k_neigh: N_NEIGH of the block's correspondences (5 for the reference LiDAR, 10 for
 * the others in buildCalibMap, estimator.cpp:1135); eig_thre / freeze: degeneracy handling -- freeze = 0 projects the weak
 * directions out (evalDegenracy, estimator.cpp:1608-1636), freeze = 1 leaves the block untouched when lambda_min < eig_thre
 * (the extrinsic branch, estimator.cpp:1662-1676, without its frame-count state). */
typedef struct mlh_block_opts {
    int n_blocks;
    int k_neigh[8];
    double eig_thre[8];
    int freeze[8];
} mlh_block_opts;
/* n_iters GN iterations on n_blocks independent pose blocks (poses_inout: n_blocks x 7), one correspondence + one fit launch
 * per iteration for all blocks and both feature kinds. stats: n_iters x n_blocks records (iteration-major), may be NULL;
 * termination = 1 in a record marks a frozen block. Multi-LiDAR features must have been staged with mlh_features_set_block (a block of one kind may be empty,
 * n = 0: a LiDAR without corner features in this frame; any subset of the blocks solved by itself returns the bits those blocks have among all of them). */
int mlh_gn_solve_blocks(mlh_ctx *ctx, double *poses_inout, int n_iters, const mlh_solver_opts *opts, const mlh_block_opts *block_opts,
                        mlh_iter_stat *stats);

/* scan2MapOptimization(): max_outer x { goodFeatureMatching (corner, then surf; wo_gf = all matched features),
 * evalHessian + evalDegenracy, Levenberg-Marquardt (Ceres trust-region semantics, <= max_lm_iterations) on the selected,
 * fixed correspondences }. Device-resident for wo_gf; the other gf methods add one host round trip per outer iteration for
 * the selection loop. replaces lidar_mapper_keyframe.cpp:423-639. stats: max_outer records, or NULL (then evalDegenracy takes
 * the eigen-decomposition only when H - thre*I is not positive definite, i.e. when something IS degenerate). */
int mlh_scan2map(mlh_ctx *ctx, double pose_inout[7], const mlh_solver_opts *opts, mlh_iter_stat *stats);
/* scan2MapOptimization submitted and collected separately (lidar_mapper_keyframe.cpp:423-639 as the mapper's per-frame call, :145-160 for the chained start pose):
 * mlh_scan2map_begin enqueues the whole solve and returns. lm_lookahead = 0 (automatic): per outer iteration the match launch and ONE launch that runs the
 * Levenberg-Marquardt loop to its end on the device -- nothing to overflow -- wherever that launch applies (one GPU, at most mlh_get_info's loop_max_tiles fit
 * tiles: 160 = 40 960 feature slots on a whole MI355X, fewer on a part of one);
 * otherwise, and with lm_lookahead > 0: per outer iteration the match launch and that many LM launches (automatic: the previous collected frame's longest LM loop + 2,
 * 10 before any; at most max_lm_iterations), launches behind the LM loop's termination find `done` on the device and leave. The caller stages the NEXT frame's maps
 * (mlh_map_set_pair_overlapped) and submits the next frame (mlh_scan2map_begin_chained: start pose = transformUpdate + transformAssociateToMap on the pose the
 * previous solve leaves on the device) before it collects this one's pose with mlh_scan2map_end. Up to two solves in flight, in submission order, shared with
 * mlh_gn_solve_begin / _end (each collected by its own _end). One GPU or the mailbox communicator; gf_method MLH_GF_WO (a selection runs host loops between launches).
 * status_out (nullable):
 *   0  the LM loops terminated inside the look-ahead: pose_out is bit for bit what mlh_scan2map returns on the same inputs;
 *   2  a loop needed more LM iterations than were enqueued -- or a one-launch loop gave its in-kernel barrier up (its workgroups were not all resident at once:
 *      mlh_get_info counts it and lowers the context's gate); nothing had been restaged and no younger solve was chained behind, so the frame was solved again
 *      synchronously from its start pose inside this call (after a barrier given up on: through the launch-per-iteration form): pose_out is mlh_scan2map's;
 *   1  the same, but the inputs of the frame are no longer staged (or a younger solve continues from this one's unfinished pose): pose_out is the frame's START
 *      pose; the caller solves the frame with mlh_scan2map on its inputs and resubmits what was chained behind it;
 *   3  this frame was chained behind a frame that ended with status 1 (or 3, or with an error): it began from that frame's unfinished pose, so pose_out -- whatever
 *      its own loops did, finished or not -- is not the mapper's; resubmit it after the predecessor has been solved.
 * With status_out == NULL the statuses 1 and 3 are returned as MLH_ERR_INCOMPLETE (pose_out is still filled in): a pose that is not a result never comes back
 * under a success code the caller cannot tell apart. */
int mlh_scan2map_begin(mlh_ctx *ctx, const double pose_in[7], const mlh_solver_opts *opts, int lm_lookahead);
int mlh_scan2map_begin_chained(mlh_ctx *ctx, const double wodom_prev[7], const double wodom_cur[7], const mlh_solver_opts *opts, int lm_lookahead);
int mlh_scan2map_end(mlh_ctx *ctx, double pose_out[7], int32_t *status_out);
/* downsampleCurrentScan (both kinds) and scan2MapOptimization back to back WITHOUT the host reading what the thinning kept (lidar_mapper_keyframe.cpp:356-421 ->
 * :423-639, as process() calls them one after the other, cpp:1000-1112): the solve's launches are sized for an upper bound (the input clouds) and take the feature
 * counts from device memory; the counts come back with the pose. Same results as mlh_downsample_current_scan_pair followed by mlh_scan2map(stats = NULL), bit for
 * bit -- and exactly those two calls wherever the fused form does not apply (clouds that are not the context's fused clouds, a good-feature selection, several
 * ranks, a local map below scan2MapOptimization's minimum, more than 131 072 input points). pose_inout is written on success only. */
int mlh_downsample_scan2map(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                            int intensity_offset_bytes, int mem, float leaf_surf, float leaf_corner, const double *ext_poses, const double *ext_covs,
                            int n_lidar, const double cov_measurement[9], int with_ua, double trace_threshold, double pose_inout[7],
                            const mlh_solver_opts *opts, int32_t *n_surf_features, int32_t *n_corner_features);

/* ---------------------------------------------------------------- (f4) scan-to-scan odometry: LidarTracker::trackCloud
 * replaces lidar_tracker.cpp:23-129 and the functions it drives: matchCornerFromScan / matchSurfFromScan
 * (feature_extract.hpp:132-376; kd-tree 1-NN within DISTANCE_SQ_THRESHOLD, then the walks over the neighbouring scan lines),
 * TransformToStart without distortion (utility.h:55-77), LidarScanPlaneNormFactor / LidarScanEdgeFactorVector
 * (lidar_scan_factor.hpp:24-64, 236-279) under HuberLoss(0.1), ceres::Solve with max_num_iterations = 4, two rounds.
 * kind MLH_CORNER: previous = "corner_points_less_sharp", current = "corner_points_sharp";
 * kind MLH_SURF:   previous = "surf_points_less_flat",    current = "surf_points_flat"   (lidar_tracker.cpp:30-38).
 * Previous-frame clouds must be ordered by ring id = int(intensity) (as extractCloud emits them); 0 <= id < 255. */
typedef struct mlh_track_opts {
    float distance_sq_threshold;     /* DISTANCE_SQ_THRESHOLD, config distance_sq_threshold (25) */
    float nearby_scan;               /* NEARBY_SCAN, config nearby_scan (2.5) */
    double huber_delta;              /* ceres::HuberLoss(0.1), lidar_tracker.cpp:45 */
    int32_t max_outer;               /* 2, cpp:42 */
    int32_t max_lm_iterations;       /* options.max_num_iterations = 4, cpp:113 */
} mlh_track_opts;
void mlh_track_opts_default(mlh_track_opts *o);
/* stage the previous frame's cloud of one kind and build its index (pcl::KdTreeFLANN::setInputCloud, cpp:33-34) */
int mlh_track_set_prev(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                       float distance_sq_threshold);
/* stage the current frame's features of one kind */
int mlh_track_set_cur(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int m, int intensity_offset_bytes, int mem);
/* device-to-device hand-over from the extractor of the SAME context (after mlh_extract_run, and mlh_extract_voxel_run for which = 1):
 * which = 0: the scan's sharp corners / flat surfs become the current frame; which = 1: its less-sharp corners / voxel-thinned
 * less-flat surfs become the previous frame (call it after mlh_track_cloud, for the next frame). The scan must have been uploaded
 * with its intensity (ring id) field. */
int mlh_track_set_from_scan(mlh_ctx *ctx, int which, float distance_sq_threshold);
/* pcl::VoxelGrid<PointXYZI>::filter (PCL 1.8.0 filters/impl/voxel_grid.hpp) over a whole cloud: one centroid per occupied voxel, x, y, z
 * AND intensity averaged, ascending voxel index; a grid of more than 2^31 cells returns the input unchanged ("leaf size is too small").
 * Estimator::buildLocalMap / buildCalibMap thin the window's clouds with it (estimator/src/estimator/estimator.cpp:1124-1130,
 * 1194-1203). out: same record layout as the input (fields other than x, y, z, intensity zeroed). Host or device buffers. */
int mlh_voxel_grid(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, float leaf, void *out,
                   int32_t *n_out, int mem);
/* pcl::transformPointCloud(cloud, cloud, pose.T_.cast<float>()) in place (estimator.cpp:1185-1192: the window clouds moved into
 * the pivot frame): x, y, z <- R p + t in single precision, the other fields kept. pose = [t(3), q(xyzw)]. */
int mlh_transform_point_cloud(mlh_ctx *ctx, void *points, int stride_bytes, int n, const double pose[7], int mem);
/* TransformToEnd (estimator/src/utility/utility.h:79-100; TransformToStart :55-77) over n records, in place: p_end = T^-1 T(s) p with
 * T(s) = (Identity.slerp(s, q), s t), s = (intensity - int(intensity)) / scan_period when b_distortion (the intensity field carries
 * ring id + time inside the sweep), else 1. f64 arithmetic, the intermediate and final points rounded to f32 as the reference's
 * float points do. pose = [t(3), q(xyzw)]. Host (copied in and out) or device buffers. */
int mlh_transform_to_end(mlh_ctx *ctx, void *points, int stride_bytes, int n, int intensity_offset_bytes, const double pose[7],
                         int b_distortion, float scan_period, int mem);
/* Estimator::undistortMeasurements (estimator.cpp:376-410, DISTORTION = 1) for the scan the context holds: TransformToEnd(...,
 * true, scan_period) on all its points (laser_cloud; the less-sharp corner list indexes them) and on its voxel-thinned less-flat
 * cloud, in place on the device. Order as in Estimator::process (estimator.cpp:532-586): the tracker consumes this scan's sharp /
 * flat features and mlh_track_set_from_scan(1) copies the (still distorted) less-sharp / less-flat clouds to the previous frame
 * first; then this call; then mlh_fuse_add_scan hands the undistorted clouds to the mapper. */
int mlh_scan_undistort(mlh_ctx *ctx, const double pose_undist[7], float scan_period);

/* The mapper's input clouds without leaving HBM. transformCloudFeature (estimator/src/utility/visualization.cpp:39-51) moves every
 * LiDAR's features into the body frame and overwrites intensity with the LiDAR index before they are published to the mapper;
 * mlh_fuse_add_scan does that for the scan the context holds (after mlh_extract_run + mlh_extract_voxel_run): its voxel-thinned
 * less-flat points are appended to the fused SURF cloud, its less-sharp points to the fused CORNER cloud (float32: (r0 x + r1 y)
 * + r2 z + t per row, R rounded once from the double quaternion). mlh_fused_cloud hands out the device pointer (float4 records
 * {x,y,z,lidar}: stride 16, intensity offset 12) for mlh_downsample_current_scan(..., MLH_MEM_DEVICE); it stays valid until the
 * next mlh_fuse_add_*. ext_pose = [t(3), q(xyzw)] of the LiDAR in the body frame. mlh_fuse_add_rings takes rings [ring_begin,
 * ring_end) of the scan only: several LiDARs uploaded as ONE scan (their rings back to back, one launch set for all of them)
 * are appended one ring range at a time, each with its own extrinsic. The appends never wait for the host: the record counts live
 * on the device and mlh_fused_cloud fetches them once. */
int mlh_fuse_reset(mlh_ctx *ctx);
int mlh_fuse_add_scan(mlh_ctx *ctx, int lidar_idx, const double ext_pose[7]);
int mlh_fuse_add_rings(mlh_ctx *ctx, int ring_begin, int ring_end, int lidar_idx, const double ext_pose[7]);
/* mlh_fuse_add_scan with the scan ANOTHER context (src, same device) holds -- estimator.cpp:248-263 runs every LiDAR's segmentCloud -> extractCloud on a thread of its
 * own; with a context per thread (the host-side cluster searches of the LiDARs side by side) this gathers their mapping features in ONE context's fused clouds
 * without a host hop. Ordered on the device in both directions: ctx's stream waits for what src's stream has been given so far, and whatever rewrites src's scan
 * next (mlh_scan_upload, mlh_segment_cloud, mlh_extract_run, mlh_extract_voxel_run, mlh_scan_undistort) waits for this append. src must be idle while this
 * call runs (its last call has returned, none is running: the calling thread touches src's bookkeeping). */
int mlh_fuse_add_scan_from(mlh_ctx *ctx, mlh_ctx *src, int lidar_idx, const double ext_pose[7]);
int mlh_fused_cloud(mlh_ctx *ctx, int kind, const void **device_points, int32_t *n);
/* match*FromScan at `pose`: valid[m] and coeffs[m x 6] ('c': closest point, second point; 's': w, negative_OA_dot_norm, 0, 0); either may be NULL */
int mlh_track_match(mlh_ctx *ctx, int kind, const double pose[7], const mlh_track_opts *opts, uint8_t *valid, double *coeffs);
/* trackCloud: pose_inout = pose_ini -> pose_prev_cur; stats: max_outer records (n_surf / n_corner = residual blocks) or NULL */
int mlh_track_cloud(mlh_ctx *ctx, double pose_inout[7], const mlh_track_opts *opts, mlh_iter_stat *stats);

/* ---------------------------------------------------------------- (e) multi-GPU: map shards + one all-reduce per iteration
 * One process (context) per GPU. The local map is partitioned spatially: rank g stages only the map points of its
 * region plus a halo >= sqrt(min_match_sq_dis) (mlh_map_set on that subset), and OWNS the features whose map-frame
 * position p = pointAssociateToMap(feature) satisfies  lo.xyz . p + lo.w >= 0  and  hi.xyz . p + hi.w < 0
 * (either plane may be NULL = unbounded; evaluated in f32 as ((a*x + b*y) + c*z) + d, so neighbouring ranks that share a
 * plane take complementary decisions and every feature has exactly one owner). Features a rank does not own contribute
 * nothing on that rank. Because every map point within the acceptance radius of an owned feature is local, the 5-NN /
 * gate decisions are identical to the unsharded ones.
 * mlh_comm_init joins the ranks (RCCL, dlopen'ed): the device-resident solvers then sum the packed normal equations
 * (32 doubles: 21 J^T J + 6 J^T r + cost + counts) with ONE ncclAllReduce per evaluation on the context's stream, and
 * every rank applies the identical 6x6 solve / Plus redundantly (no pose broadcast). */
int mlh_shard_set(mlh_ctx *ctx, const float *lo_plane4, const float *hi_plane4);
/* The balanced alternative of SURVEY 8(e): every rank stages the WHOLE map (4 M points are 64 MB of 288 GB) and owns the features whose
 * slot index f satisfies f % n_ranks == rank -- no halo, equal shares whatever the scene, the same one all-reduce per evaluation. The index
 * build is then replicated instead of divided, which is the price. n_ranks = 1 switches it off; it combines with mlh_shard_set (both tests
 * must hold) but is meant to be used instead of it. */
int mlh_shard_set_features(mlh_ctx *ctx, int n_ranks, int rank);
int mlh_comm_unique_id(void *out_128_bytes);
int mlh_comm_init(mlh_ctx *ctx, int n_ranks, int rank, const void *unique_id_128_bytes);
/* The same join WITHOUT a collective library -- the mailbox communicator: the path's one collective is a few hundred bytes per evaluation, i.e. pure latency,
 * and a library all-reduce (protocol selection, proxy threads, a ring over the ranks) costs tens of microseconds there. mlh_p2p_mailbox allocates this rank's
 * mailbox in device memory and returns its 64-byte hipIpc handle; the caller exchanges the handles out of band (one all-gather of 64 bytes: MPI, torch.distributed,
 * a file ...) and hands all n_ranks of them, in rank order, to mlh_p2p_comm_init, which maps the peers' mailboxes (peer access over xGMI between GPUs; ranks may
 * also share a GPU). Every all-reduce of the solvers (and mlh_allreduce_f64, up to 512 doubles) is then ONE single-workgroup kernel per rank on the context's stream:
 * store my record into my slot of every mailbox, publish a sequence number, wait for the n sequence numbers in my own mailbox, add the n slots in rank order --
 * every rank ends with the same bits. A peer that never shows up raises an error on the next call instead of hanging the GPU (5 s bound). <= 16 ranks. */
int mlh_p2p_mailbox(mlh_ctx *ctx, void *ipc_handle_64_bytes);
int mlh_p2p_comm_init(mlh_ctx *ctx, int n_ranks, int rank, const void *ipc_handles /* n_ranks x 64 bytes, rank order */);
/* drops the communicator (either kind): the context is single-GPU again */
int mlh_comm_finalize(mlh_ctx *ctx);
/* in-place sum of n doubles (HOST buffer, any n) over the ranks -- the standalone "mlh_allreduce_normal_eq" of SURVEY 8b */
int mlh_allreduce_f64(mlh_ctx *ctx, double *host_inout, int n);

/* host-side helpers mirroring the reference's small functions (no GPU work) */
/* PoseLocalParameterization::Plus with V_update_ (row-major 6x6; NULL = identity) */
int mlh_pose_plus(const double x[7], const double delta[6], const double *V_update, double x_plus_delta[7]);
/* evalDegenracy: eigenvalues ascending, V_update = (V_f^T)^-1 V_p^T, returns 1 when degenerate */
int mlh_eval_degeneracy(const double H[36], double eig_thre, double eigval[6], double V_update[36]);

#ifdef __cplusplus
}
#endif
#endif /* MLOAM_HIP_H */
