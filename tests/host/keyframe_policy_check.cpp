// Host logic check (no GPU): the facade's Pose arithmetic (poseMul / poseInverse = Pose::operator* / Pose::inverse, pose.cpp:99-113) and KeyframePolicy
// (saveKeyframe's test, lidar_mapper_keyframe.cpp:641-657; extractSurroundingKeyFrames' radius search, cpp:266-272) on poses read from a file. Compiled and run by
// tests/test_abi.py::test_facade_keyframe_policy_and_pose_chain, which holds the chained start pose against the reference's own lines (oracle/_ref) and the
// decisions against a restatement.
// argv: dir  n_poses  distance_keyframes  orientation_keyframes_deg  radius
#include "mloam_facade.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>

template <typename T> static std::vector<T> read_file(const std::string &p)
{
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", p.c_str()); std::exit(2); }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v(size_t(n) / sizeof(T));
    f.read(reinterpret_cast<char *>(v.data()), n);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    using namespace mloam_hip;
    const std::string d = std::string(argv[1]) + "/";
    const int n = std::atoi(argv[2]);
    auto raw = read_file<double>(d + "poses.f64");            // n x 7: [t, q(xyzw)]
    std::vector<Pose> poses(n);
    for (int i = 0; i < n; ++i) poses[i].fromParam(raw.data() + 7 * i);
    std::vector<double> out;
    // the chain of lidar_mapper_keyframe.cpp:145-160 for consecutive triples: (wmap_curr = poses[i], wodom_prev = poses[i + 1], wodom_cur = poses[i + 2])
    for (int i = 0; i + 2 < n; ++i) {
        const Pose wmap_wodom = poseMul(poses[i], poseInverse(poses[i + 1]));
        const Pose start = poseMul(wmap_wodom, poses[i + 2]);
        double p[7];
        start.toParam(p);
        out.insert(out.end(), p, p + 7);
    }
    // the keyframe bookkeeping over the pose sequence
    KeyframePolicy kf(float(std::atof(argv[3])), float(std::atof(argv[4])), float(std::atof(argv[5])));
    for (int i = 0; i < n; ++i) {
        const bool would = kf.wouldSave(poses[i]);
        const int idx = kf.save(poses[i]);
        const std::vector<int> ids = kf.surrounding(poses[i]);
        out.push_back(would ? 1.0 : 0.0);
        out.push_back(double(idx));
        out.push_back(double(ids.size()));
        double h = 0.0;                                       // order-sensitive digest of the id list
        for (size_t k = 0; k < ids.size(); ++k) h = h * 31.0 + double(ids[k] + 1);
        out.push_back(h);
    }
    std::ofstream f(d + "out.f64", std::ios::binary);
    f.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size() * sizeof(double)));
    return 0;
}
