// Host logic check (no GPU): the facade's per-factor classes of the odometry window and the calibration -- LidarPureOdom{PlaneNorm,Edge}Factor
// (lidar_pure_odom_factor.hpp:27-102, 198-282) and LidarOnlineCalib{PlaneNorm,Edge}Factor (lidar_online_calib_factor.hpp:24-62, 125-165). Compiled and run by
// tests/test_abi.py::test_facade_odometry_factors_are_the_references, which holds residuals and Jacobians against the reference's own lines (oracle/_ref).
// argv: dir      in: dir/ofactors.f64 (n x 32: type, point[3], coeff[6], s, pivot[7], pose_i[7], ext[7])
// out: dir/ofactors_out.f64 (n x 30: residual, J[21] (pivot | frame | ext), calib residual, calib J[7])
#include "mloam_facade.hpp"
#include <cstdio>
#include <fstream>

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    using namespace mloam_hip;
    const std::string d = std::string(argv[1]) + "/";
    std::ifstream f(d + "ofactors.f64", std::ios::binary | std::ios::ate);
    if (!f) return 2;
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<double> raw(size_t(bytes) / sizeof(double));
    f.read(reinterpret_cast<char *>(raw.data()), bytes);
    std::vector<double> out;
    for (size_t i = 0; i + 32 <= raw.size(); i += 32) {
        const double *r = raw.data() + i;
        const std::array<double, 3> p = {r[1], r[2], r[3]};
        const bool plane = r[0] == 0.0;
        const std::vector<double> coeff(r + 4, r + (plane ? 8 : 10));
        const double *prm[3] = {r + 11, r + 18, r + 25};
        double res = 0.0, J[21] = {0};
        double *jac[3] = {J, J + 7, J + 14};
        if (plane) { LidarPureOdomPlaneNormFactor fac(p, coeff, r[10]); fac.Evaluate(prm, &res, jac); }
        else { LidarPureOdomEdgeFactor fac(p, coeff, r[10]); fac.Evaluate(prm, &res, jac); }
        out.push_back(res);
        out.insert(out.end(), J, J + 21);
        double rc = 0.0, Jc[7] = {0};
        double *jc[1] = {Jc};
        const double *pc[1] = {r + 25};
        if (plane) { LidarOnlineCalibPlaneNormFactor fac(p, coeff, r[10]); fac.Evaluate(pc, &rc, jc); }
        else { LidarOnlineCalibEdgeFactor fac(p, coeff, r[10]); fac.Evaluate(pc, &rc, jc); }
        out.push_back(rc);
        out.insert(out.end(), Jc, Jc + 7);
    }
    std::ofstream o(d + "ofactors_out.f64", std::ios::binary);
    o.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size() * sizeof(double)));
    return 0;
}
