// TEST INFRASTRUCTURE: stands where the reference's own mloam_pcl/include/mloam_pcl/point_with_cov.hpp is found inside the reference tree -- ONLY for the
// INTEGRATION section-1 snippet test (tests/test_facade_refcut.py), which has to compile on boxes without /root/reference. The layout is that header's
// (:45-53: PCL_ADD_POINT4D, intensity, cov_vec[6], cov_trace -> 48 bytes). tests/host/refcut does NOT use this file: it compiles the reference's real lines.
#pragma once
#include <pcl/point_types.h>
namespace pcl {
struct EIGEN_ALIGN16 _PointXYZIWithCov {
    PCL_ADD_POINT4D;
    float intensity;
    float cov_vec[6];
    float cov_trace;
};
struct PointXYZIWithCov : public _PointXYZIWithCov {
    inline PointXYZIWithCov() { x = y = z = 0.0f; data[3] = 1.0f; intensity = 0; for (float &c : cov_vec) c = 0.0f; cov_trace = 0.0f; }
};
}  // namespace pcl
