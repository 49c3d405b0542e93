// Host logic check (no GPU): the facade's per-factor classes LidarMapPlaneNormFactor / LidarMapEdgeFactor (lidar_map_factor.hpp:28-71, 132-174), for callers that keep
// the reference's AddResidualBlock loop (lidar_mapper_keyframe.cpp:537-571). Compiled and run by tests/test_abi.py::test_facade_map_factors_are_the_references, which holds
// residuals and Jacobians against the reference's own lines (oracle/_ref).
// argv: dir      in: dir/factors.f64 (n x 26: type, point[3], coeff[6], cov[9], pose[7])      out: dir/factors_out.f64 (n x 8: residual, J[7])
#include "mloam_facade.hpp"
#include <cstdio>
#include <fstream>

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    using namespace mloam_hip;
    const std::string d = std::string(argv[1]) + "/";
    std::ifstream f(d + "factors.f64", std::ios::binary | std::ios::ate);
    if (!f) return 2;
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<double> raw(size_t(bytes) / sizeof(double));
    f.read(reinterpret_cast<char *>(raw.data()), bytes);
    std::vector<double> out;
    for (size_t i = 0; i + 26 <= raw.size(); i += 26) {
        const double *r = raw.data() + i;
        const std::array<double, 3> p = {r[1], r[2], r[3]};
        std::array<double, 9> cov;
        for (int k = 0; k < 9; ++k) cov[size_t(k)] = r[10 + k];
        const double *pose = r + 19;
        double res = 0.0, J[7] = {0, 0, 0, 0, 0, 0, 0};
        double *jac[1] = {J};
        const double *prm[1] = {pose};
        if (r[0] == 0.0) { LidarMapPlaneNormFactor fac(p, std::vector<double>(r + 4, r + 8), cov); fac.Evaluate(prm, &res, jac); }
        else { LidarMapEdgeFactor fac(p, std::vector<double>(r + 4, r + 10), cov); fac.Evaluate(prm, &res, jac); }
        out.push_back(res);
        out.insert(out.end(), J, J + 7);
    }
    std::ofstream o(d + "factors_out.f64", std::ios::binary);
    o.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size() * sizeof(double)));
    return 0;
}
