// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED: the reference ships
// no golden vectors for this path and PCL/Eigen/FLANN cannot be built here.
//
// CPU restatement of
//   FeatureExtract::extractCloud             estimator/src/featureExtract/feature_extract.cpp:118-297
//   pcl::VoxelGrid<PointXYZI>::applyFilter   (PCL 1.8.0, filters/impl/voxel_grid.hpp; call site cpp:266-271)
//   FeatureExtract::match{Corner,Surf}PointFromMap   feature_extract.hpp:646-883
//   FeatureExtract::match{Corner,Surf}FromMap        feature_extract.hpp:379-643
//   pointAssociateToMap                      estimator/src/utility/utility.h:103-117
#pragma once
#include <vector>
#include <cstdint>
#include "geometry.hpp"
#include "kdtree.hpp"

namespace orc {

struct PointI { float x, y, z, intensity; };

struct ExtractResult {
    std::vector<float> curvature;   // n, 0 outside [5, n-5)
    std::vector<int> label;         // n: 2 sharp, 1 less sharp, -1 flat, 0 other
    std::vector<int> picked;        // n: cloud_neighbor_picked at the end
    std::vector<int> sharp, less_sharp, flat;   // indices in emission order (cpp:177-178, 183, 225)
    std::vector<int> less_flat_raw;             // indices (label<=0 by position, ring asc, sector asc) before VoxelGrid
    std::vector<int> less_flat_raw_ring_off;    // offsets into less_flat_raw per processed ring (size rings+1)
    std::vector<PointI> less_flat_ds;           // after per-ring VoxelGrid(0.2) (cpp:266-271)
    long n_ties = 0;                            // equal-curvature neighbours in sorted sector order (std::sort is unstable)
};

void set_tie_rule(int rule);   // 0: reference comparator (default), 1: (curvature, index) total order with NaN last -- see feature_extract.cpp
void extract_cloud(const PointI *cloud, int n, const int *scan_start, const int *scan_end, int n_scans,
                   ExtractResult &out);

// pcl::VoxelGrid<PointXYZI> with downsample_all_data_=true, min_points_per_voxel_=0
void voxel_grid_xyzi(const PointI *in, int n, float leaf, std::vector<PointI> &out);

// a map cloud + its kd-tree (the reference passes the two separately; they hold the same points)
struct MapCloud {
    const float *pts = nullptr;   // xyz at pts[i*stride .. +2]
    size_t stride = 0;            // in floats
    int n = 0;
    KdTree tree;
    void set(const float *p, size_t stride_floats, int n_) { pts = p; stride = stride_floats; n = n_; tree.build(p, stride_floats, n_); }
};

// PointPlaneFeature (estimator/src/estimator/parameters.h:163-175)
struct Feature {
    size_t idx = 0;
    size_t laser_idx = 0;
    double point[3] = {0, 0, 0};
    double coeffs[6] = {0, 0, 0, 0, 0, 0};   // 's': n_hat, d ; 'c': X1, X2
    double jaco[6] = {0, 0, 0, 0, 0, 0};
    char type = 'n';
};

struct MatchParams {
    float min_match_sq_dis = 1.0f;   // parameters.cpp:232 / config min_match_sq_dis
    float min_plane_dis = 0.2f;      // parameters.cpp:233 / config min_plane_dis
};

// utility.h:103-117 : f64 q*p+t, stored back to f32
void point_associate_to_map(const float pi[3], float po[3], const Pose &pose);

bool match_corner_point_from_map(const MapCloud &map, const float *point_ori /*x y z intensity*/, const Pose &pose_local,
                                 Feature &feature, size_t idx, int n_neigh, bool check_fov, const MatchParams &mp);
bool match_surf_point_from_map(const MapCloud &map, const float *point_ori, const Pose &pose_local,
                               Feature &feature, size_t idx, int n_neigh, bool check_fov, const MatchParams &mp);

// batch versions: compact matches in input order
void match_corner_from_map(const MapCloud &map, const float *cloud_data, size_t stride_floats, int n, const Pose &pose_local,
                           std::vector<Feature> &features, int n_neigh, bool check_fov, const MatchParams &mp);
void match_surf_from_map(const MapCloud &map, const float *cloud_data, size_t stride_floats, int n, const Pose &pose_local,
                         std::vector<Feature> &features, int n_neigh, bool check_fov, const MatchParams &mp);

}  // namespace orc
