// Host logic check (no GPU): the facade's FeatureExtract::calTimestamp (feature_extract.cpp:54-114) on raw clouds read from a file; compiled and run by
// tests/test_abi.py::test_facade_cal_timestamp_is_the_references and scripts/soak_ref_pin.py, which hold its output against the reference's own lines (oracle/_ref).
// argv: dir  scan_period      in: dir/cloud.f32 (n x 3)   out: dir/rel_time.f32 (n)
#include "mloam_facade.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    using namespace mloam_hip;
    const std::string d = std::string(argv[1]) + "/";
    std::ifstream f(d + "cloud.f32", std::ios::binary | std::ios::ate);
    if (!f) return 2;
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<float> raw(size_t(bytes) / sizeof(float));
    f.read(reinterpret_cast<char *>(raw.data()), bytes);
    PointCloud<PointXYZ> in;
    for (size_t i = 0; i + 3 <= raw.size(); i += 3) { PointXYZ p; p.x = raw[i]; p.y = raw[i + 1]; p.z = raw[i + 2]; in.push_back(p); }
    PointICloud out;
    FeatureExtract fe;
    fe.calTimestamp(in, out, float(std::atof(argv[2])));
    std::vector<float> t;
    for (const auto &q : out.points) t.push_back(q.intensity);
    std::ofstream o(d + "rel_time.f32", std::ios::binary);
    o.write(reinterpret_cast<const char *>(t.data()), std::streamsize(t.size() * sizeof(float)));
    return 0;
}
