#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY. Builds oracle/_ref/libmloam_ref.so from the reference's own source lines (see ref_shim.cpp).
Needs /root/reference (absent on the GPU box: the prebuilt .so travels there). Nothing of the reference is copied into the repository:
the cut-out line ranges live in oracle/_ref/gen/ only for the duration of the compile."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MLOAM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "_ref")
SRC = os.path.join(REF, "estimator", "src")
COMMON = os.path.join(REF, "mloam_common", "libs", "include", "common")
# (output, file, first line, last line, text the first line must contain) -- a drifted reference fails loudly instead of compiling something else
CUTS = [
    ("image_segmenter_class.inc", "imageSegmenter/image_segmenter.hpp", 36, 83, "class ImageSegmenter"),
    ("image_segmenter_project.inc", "imageSegmenter/image_segmenter.hpp", 87, 136, "template <typename PointType>"),
    ("image_segmenter_segment.inc", "imageSegmenter/image_segmenter.hpp", 138, 393, "template <typename PointType>"),
    ("image_segmenter_setparam.inc", "imageSegmenter/image_segmenter.cpp", 18, 63, "void ImageSegmenter::setParameter"),
    ("sqr_sum.inc", "algos/math.hpp", 10, 14, "template <typename T>"),
    ("comp_object.inc", "featureExtract/feature_extract.hpp", 48, 53, "class compObject"),
    ("extract_cloud.inc", "featureExtract/feature_extract.cpp", 118, 297, "void FeatureExtract::extractCloud"),
    ("match_point_decls.inc", "featureExtract/feature_extract.hpp", 110, 128, "template <typename PointType>"),
    ("cal_timestamp.inc", "featureExtract/feature_extract.cpp", 54, 114, "void FeatureExtract::findStartEndAngle"),
    ("match_batch_decls.inc", "featureExtract/feature_extract.hpp", 92, 108, "template <typename PointType>"),
    ("match_corner_batch.inc", "featureExtract/feature_extract.hpp", 378, 538, "template <typename PointType>"),
    ("match_surf_batch.inc", "featureExtract/feature_extract.hpp", 541, 643, "template <typename PointType>"),
    ("match_corner_point.inc", "featureExtract/feature_extract.hpp", 645, 788, "template <typename PointType>"),
    ("match_surf_point.inc", "featureExtract/feature_extract.hpp", 790, 883, "template <typename PointType>"),
    ("point_associate_to_map.inc", "utility/utility.h", 102, 117, "template <typename PointType>"),
    ("utility_head.inc", "utility/utility.h", 169, 195, "class Utility"),
    ("plane_factor_head.inc", "factor/lidar_map_factor.hpp", 26, 71, "class LidarMapPlaneNormFactor"),
    ("plane_factor_tail.inc", "factor/lidar_map_factor.hpp", 122, 126, "private:"),
    ("edge_factor_head.inc", "factor/lidar_map_factor.hpp", 130, 174, "class LidarMapEdgeFactor"),
    ("edge_factor_tail.inc", "factor/lidar_map_factor.hpp", 231, 235, "private:"),
    ("odom_plane_head.inc", "factor/lidar_pure_odom_factor.hpp", 27, 102, "class LidarPureOdomPlaneNormFactor"),
    ("odom_plane_tail.inc", "factor/lidar_pure_odom_factor.hpp", 191, 195, "private:"),
    ("odom_edge_head.inc", "factor/lidar_pure_odom_factor.hpp", 198, 282, "class LidarPureOdomEdgeFactor"),
    ("odom_edge_tail.inc", "factor/lidar_pure_odom_factor.hpp", 377, 381, "private:"),
    ("calib_plane_head.inc", "factor/lidar_online_calib_factor.hpp", 24, 62, "class LidarOnlineCalibPlaneNormFactor"),
    ("calib_plane_tail.inc", "factor/lidar_online_calib_factor.hpp", 117, 121, "private:"),
    ("calib_edge_head.inc", "factor/lidar_online_calib_factor.hpp", 125, 165, "class LidarOnlineCalibEdgeFactor"),
    ("calib_edge_tail.inc", "factor/lidar_online_calib_factor.hpp", 223, 227, "private:"),
    ("plp_class.inc", "factor/pose_local_parameterization.h", 21, 33, "class PoseLocalParameterization"),
    ("plp_plus.inc", "factor/pose_local_parameterization.cpp", 16, 45, "void PoseLocalParameterization::setParameter"),
    ("plp_jacobian.inc", "factor/pose_local_parameterization.cpp", 47, 55, "// calculate the jacobian of [p, q] w.r.t [dp, dq]"),
    ("point_cov_ctor_default.inc", "@mloam_pcl/include/mloam_pcl/point_with_cov.hpp", 57, 62, "inline PointXYZIWithCov()"),
    ("point_cov_ctor_from_point.inc", "@mloam_pcl/include/mloam_pcl/point_with_cov.hpp", 90, 100, "inline PointXYZIWithCov(const PointXYZI &p, const Eigen::Matrix3f &cov_matrix)"),
    ("voxel_filter_apply.inc", "@mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp", 68, 457, "template <typename PointT> void"),
    ("downsample_current_scan.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 356, 421, "void downsampleCurrentScan()"),
    ("pose_ctor_default.inc", "estimator/pose.cpp", 16, 23, "Pose::Pose()"),
    ("pose_ctor_copy.inc", "estimator/pose.cpp", 25, 32, "Pose::Pose(const Pose &pose)"),
    ("pose_ctor_qt.inc", "estimator/pose.cpp", 34, 41, "Pose::Pose(const Eigen::Quaterniond &q"),
    ("pose_ctor_T.inc", "estimator/pose.cpp", 52, 59, "Pose::Pose(const Eigen::Matrix4d &T"),
    ("pose_inverse_update.inc", "estimator/pose.cpp", 99, 108, "Pose Pose::inverse() const"),
    ("pose_mul.inc", "estimator/pose.cpp", 110, 113, "Pose Pose::operator * (const Pose &pose)"),
    ("update_cov.inc", "@mloam_pcl/include/mloam_pcl/point_with_cov.hpp", 191, 200, "void updateCov(pcl::PointXYZIWithCov &po"),
    ("uct_compound_pose.inc", "lidarMapper/associate_uct.hpp", 88, 147, "// fixed: topLeftCorner<3, 3>()"),
    ("cloud_uct_associate.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 1116, 1158, "void cloudUCTAssociateToMap"),
    ("eval_degeneracy.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 1171, 1204, "// TODO: still have some bugs"),
    ("scan_match_decls.inc", "featureExtract/feature_extract.hpp", 78, 90, "template <typename PointType>"),
    ("transform_start_end.inc", "utility/utility.h", 54, 100, "template <typename PointType>"),
    ("match_from_scan.inc", "featureExtract/feature_extract.hpp", 131, 376, "template <typename PointType>"),
    ("scan_plane_head.inc", "factor/lidar_scan_factor.hpp", 25, 62, "class LidarScanPlaneNormFactor"),
    ("scan_plane_tail.inc", "factor/lidar_scan_factor.hpp", 122, 126, "private:"),
    ("scan_edge_vec_head.inc", "factor/lidar_scan_factor.hpp", 236, 279, "class LidarScanEdgeFactorVector"),
    ("scan_edge_vec_tail.inc", "factor/lidar_scan_factor.hpp", 339, 343, "private:"),
    ("feature_structs.inc", "estimator/parameters.h", 163, 191, "class PointPlaneFeature"),
    ("extract_cov.inc", "@mloam_pcl/include/mloam_pcl/point_with_cov.hpp", 202, 214, "void extractCov"),
    ("log_det.inc", "algos/math.hpp", 172, 202, "template <typename MatrixType>"),
    ("rgi_head.inc", "@mloam_common/libs/include/common/random_generator.hpp", 52, 66, "template < typename T >"),
    ("afs_limits.inc", "lidarMapper/lidar_mapper.h", 82, 83, "#define MAX_FEATURE_SELECT_TIME"),
    ("afs_eval_jaco.inc", "lidarMapper/lidar_mapper.h", 130, 174, "void evaluateFeatJacobianMatching"),
    ("afs_full_hessian.inc", "lidarMapper/lidar_mapper.h", 176, 227, "void evalFullHessian"),
    ("afs_gfm.inc", "lidarMapper/lidar_mapper.h", 229, 573, "void goodFeatureMatching"),
    ("crs_to_sparse.inc", "utility/utility.h", 152, 166, "template <typename T>"),
    ("estimator_eval_degeneracy.inc", "estimator/estimator.cpp", 1598, 1680, "void Estimator::evalDegenracy"),
    ("estimator_optimize_map_head.inc", "estimator/estimator.cpp", 593, 866, "void Estimator::optimizeMap()"),
    ("estimator_vector_double.inc", "estimator/estimator.cpp", 1538, 1576, "void Estimator::vector2Double()"),
    ("estimator_eval_residual.inc", "estimator/estimator.cpp", 1578, 1595, "void Estimator::evalResidual(ceres::Problem &problem,"),
    ("estimator_eval_feat_jacobian.inc", "estimator/estimator.cpp", 1273, 1345, "void Estimator::evaluateFeatJacobian"),
    ("estimator_gfm.inc", "estimator/estimator.cpp", 1347, 1517, "void Estimator::goodFeatureMatching"),
    ("eval_hessian.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 1160, 1169, "void evalHessian"),
    ("vector2double.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 236, 252, "void vector2Double()"),
    ("scan2map_optimization.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 423, 639, "void scan2MapOptimization()"),
    ("transform_associate_update.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 145, 160, "void transformAssociateToMap()"),
    ("save_keyframe.inc", "lidarMapper/lidar_mapper_keyframe.cpp", 641, 683, "void saveKeyframe()"),
    ("track_cloud.inc", "lidarTracker/lidar_tracker.cpp", 23, 129, "Pose LidarTracker::trackCloud"),
    ("uct_compound.inc", "lidarMapper/associate_uct.hpp", 9, 86, "inline Eigen::Matrix<double, 6, 6> adjointMatrix"),
    ("uct_point_to_fs.inc", "lidarMapper/associate_uct.hpp", 150, 156, "inline Eigen::Matrix<double, 4, 6> pointToFS"),
    ("uct_eval_point_cov.inc", "lidarMapper/associate_uct.hpp", 164, 193, "template <typename PointType>"),
    ("uct_eval_point.inc", "lidarMapper/associate_uct.hpp", 195, 215, "template <typename PointType>"),
]


def build(force=False):
    lib = os.path.join(OUT, "libmloam_ref.so")
    if not os.path.isdir(SRC):
        return lib if os.path.exists(lib) else None          # GPU box: use what travelled
    srcs = [os.path.join(HERE, f) for f in ("ref_shim.cpp", "mini_eigen.hpp", "build_ref.py")] + [os.path.join(os.path.dirname(HERE), "lm.hpp")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in srcs):
        return lib
    gen = os.path.join(OUT, "gen")
    os.makedirs(gen, exist_ok=True)
    try:
        for name, rel, a, b, must in CUTS:
            base = REF if rel.startswith("@") else (COMMON if rel.startswith("algos/") else SRC)      # "@...": relative to the reference root
            lines = open(os.path.join(base, rel.lstrip("@"))).read().split("\n")
            assert must in lines[a - 1], f"{rel}:{a} is not '{must}' -- the reference tree differs from the surveyed one"
            open(os.path.join(gen, name), "w").write("\n".join(lines[a - 1:b]) + "\n")
        oracle_dir = os.path.dirname(HERE)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-o", lib, os.path.join(HERE, "ref_shim.cpp"),
               os.path.join(oracle_dir, "feature_extract.cpp")]
        subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(gen, ignore_errors=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
