// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (Ceres absent).
//
// CPU restatement of the mapper's scan-to-map optimisation:
//   scan2MapOptimization                    estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:423-639
//   evalHessian / evalDegenracy             lidar_mapper_keyframe.cpp:1160-1204
//   ActiveFeatureSelection::{evaluateFeatJacobianMatching, evalFullHessian, goodFeatureMatching}
//                                           estimator/src/lidarMapper/lidar_mapper.h:130-573
//   extractCov                              mloam_pcl/include/mloam_pcl/point_with_cov.hpp:202-214
//   ceres::Solve (trust-region Levenberg-Marquardt, DENSE_SCHUR on one 6-dof block, Jacobi scaling,
//   Huber loss) -- Ceres 1.12.0 (docker/Dockerfile:3), restated from its published algorithm
//   (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc); see SURVEY.md Appendix B.
#pragma once
#include "feature_extract.hpp"
#include "factors.hpp"
#include "lm.hpp"            // NormalEq, SolveSummary, the generic Levenberg-Marquardt iteration
#include <string>
#include <vector>
#include <random>

namespace orc {

// feature cloud in the PointXYZIWithCov role: xyz at [0..2], intensity at [3], cov_vec[6] at cov_off (or -1: zeros)
struct FeatureCloud {
    const float *pts = nullptr;
    size_t stride = 0;
    int n = 0;
    int cov_off = -1;
    const float *at(size_t i) const { return pts + i * stride; }
    double cov_trace(size_t i) const   // trace of extractCov(): f32 entries widened to f64
    {
        if (cov_off < 0) return 0.0;
        const float *c = at(i) + cov_off;
        return (double(c[0]) + double(c[3])) + double(c[5]);
    }
};

struct ResidualBlock {
    char type;          // 's' plane, 'c' edge
    double point[3];
    double coeffs[6];
    double sqrt_info;
};

// one pass over the residual blocks at pose x (loss-corrected as Ceres' ResidualBlock::Evaluate)
void evaluate_problem(const std::vector<ResidualBlock> &blocks, const double x[7], double huber_delta,
                      NormalEq &ne, bool with_jacobian);

struct Degeneracy {
    double eigval[6];
    double eigvec[36];     // columns
    double V_update[36];   // mat_P = (V_f^T)^-1 V_p^T
    bool is_degenerate;
};
void eval_degeneracy(const double H[36], double eig_thre, Degeneracy &out);

// Estimator::evalDegenracy on the odometry window's J^T J (estimator/src/estimator/estimator.cpp:1598-1680). Blocks 0 .. n_pose_blocks-1
// are the window's pose blocks (OPT_WINDOW_SIZE + 1), the rest one per LiDAR extrinsic. eig_thre (one per block) is read and, for the
// extrinsic blocks, UPDATED as the reference updates eig_thre_. Outputs per block: is_degenerate, V_update (row-major 6x6; identity when the
// block is left untouched, the reference's setParameter() value), eigval (ascending); d_factor_calib per extrinsic block.
void window_eval_degeneracy(const double *JtJ, int D, int n_pose_blocks, double *eig_thre, bool estimate_extrinsic, long frame_cnt,
                            int n_cumu_feature, double lambda_thre_calib, int *is_degenerate, double *V_update, double *eigval, double *d_factor_calib);

// Ceres-shaped LM on the single pose block. V_update only affects Plus.
void ceres_like_solve(const std::vector<ResidualBlock> &blocks, double x[7], const double V_update[36],
                      double huber_delta, int max_num_iterations, SolveSummary &summary);

struct SelectParams {
    std::string gf_method = "wo_gf";   // wo_gf | rnd | fps | gd_fix | gd_float
    double gf_ratio = 1.0;
    uint64_t seed = 0;                 // the reference seeds from std::random_device; fixed here
};

// lidar_mapper.h:229-573 (wall-clock cut-offs removed: MAX_FEATURE_SELECT_TIME is not applied)
// Estimator::goodFeatureMatching (estimator.cpp:1347-1517): the odometry's selection (gf_ratio = ODOM_GF_RATIO as a widened float)
void odom_good_feature_matching(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose_local, const double pivot[7], const double pose_i[7],
                                const double ext[7], std::vector<Feature> &all_features, std::vector<size_t> &sel_feature_idx, char feature_type,
                                double gf_ratio, const MatchParams &mp, std::mt19937 &rng);
void good_feature_matching(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose_local,
                           std::vector<Feature> &all_features, std::vector<size_t> &sel_feature_idx,
                           char feature_type, const SelectParams &sp, double sub_mat_H[36],
                           const MatchParams &mp, std::mt19937 &rng);

// lidar_mapper.h:176-227
void eval_full_hessian(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose_local, char feature_type,
                       double mat_H[36], int &feat_num, const MatchParams &mp);

struct MapperParams {
    MatchParams mp;
    double huber_delta = 0.1;          // lidar_mapper_keyframe.cpp:443
    double map_eig_thre = 100.0;       // config map_eig_thre
    bool with_ua = false;
    double cov_measurement_trace = 0.0075;   // trace(COV_MEASUREMENT), config uct_measurement 0.0025 x3
    int max_outer = 2;                 // cpp:439
    int max_lm_iterations = 30;        // cpp:590
    SelectParams sel;
    // per-block options of the odometry-side clients (BASELINE config 4): buildCalibMap matches the reference LiDAR with
    // N_NEIGH = 5 and the others with N_NEIGH = 10, CHECK_FOV = true (estimator.cpp:1135-1149); an extrinsic block whose
    // information is too weak is not updated (V_update_ = 0, estimator.cpp:1662-1676) instead of being projected.
    int n_neigh = 5;
    bool check_fov = false;
    bool freeze_when_degenerate = false;
};

struct OuterStat {
    int n_surf_sel = 0, n_corner_sel = 0;
    double H0[36];                     // evalHessian at the start of the outer iteration
    Degeneracy deg;
    SolveSummary solve;
    double pose_after[7];
};

struct Scan2MapResult {
    double pose[7];
    std::vector<OuterStat> outer;
    double H_final[36];                // with_ua: evalHessian after the last solve (cpp:600-610)
};

void scan2map_optimization(const MapCloud &surf_map, const MapCloud &corner_map,
                           const FeatureCloud &surf, const FeatureCloud &corner,
                           const double pose_init[7], const MapperParams &prm, Scan2MapResult &res);

// BASELINE "GN iteration": re-match at the current pose (wo_gf), linearise with Huber correction, evalDegenracy,
// solve H d = -g (Cholesky), x <- Plus(x, d).  One call = one iteration.
struct GnIterStat { NormalEq ne; int n_surf = 0, n_corner = 0; Degeneracy deg; double pose_after[7]; };
void gn_iteration(const MapCloud &surf_map, const MapCloud &corner_map, const FeatureCloud &surf, const FeatureCloud &corner,
                  double x[7], const MapperParams &prm, GnIterStat &st, int n_threads);

}  // namespace orc
