// TEST INFRASTRUCTURE: the two PCL point types on M-LOAM's hot path, with PCL 1.8's layouts (pcl/impl/point_types.hpp: PointXYZ = float data[4], 16 bytes;
// PointXYZI = data[4] + { intensity | data_c[4] }, 32 bytes) and the macros the reference's own point type is written with (mloam_pcl/point_with_cov.hpp:45-53).
#pragma once
#include <ostream>
#ifndef EIGEN_ALIGN16
#define EIGEN_ALIGN16 __attribute__((aligned(16)))
#endif
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#define PCL_ADD_UNION_POINT4D union EIGEN_ALIGN16 { float data[4]; struct { float x; float y; float z; }; };
#define PCL_ADD_POINT4D PCL_ADD_UNION_POINT4D
#define POINT_CLOUD_REGISTER_POINT_STRUCT(...)
#define POINT_CLOUD_REGISTER_POINT_WRAPPER(...)
namespace pcl {
struct EIGEN_ALIGN16 PointXYZ {
    PCL_ADD_POINT4D;
    inline PointXYZ() { x = y = z = 0.0f; data[3] = 1.0f; }
    inline PointXYZ(float _x, float _y, float _z) { x = _x; y = _y; z = _z; data[3] = 1.0f; }
};
struct EIGEN_ALIGN16 PointXYZI {
    PCL_ADD_POINT4D;
    union { struct { float intensity; }; float data_c[4]; };
    inline PointXYZI() { x = y = z = 0.0f; data[3] = 1.0f; intensity = 0.0f; data_c[1] = data_c[2] = data_c[3] = 0.0f; }
    inline PointXYZI(float _intensity) { x = y = z = 0.0f; data[3] = 1.0f; intensity = _intensity; data_c[1] = data_c[2] = data_c[3] = 0.0f; }
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZI) == 32, "PCL layouts");
}  // namespace pcl
