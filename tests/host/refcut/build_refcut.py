#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY. The reference's OWN CALL SITES of the hot path, cut verbatim out of /root/reference at build time and compiled, unchanged,
against the drop-in boundary: m-loam_amd/host/mloam_facade.hpp with -DMLOAM_FACADE_USE_PCL_TYPES -DMLOAM_FACADE_CERES_BASES over headers shaped like
PCL / boost / Eigen / Ceres (tests/host/pcl_stub). What is cut (file, lines, text the first line must hold):

  the estimator's OpenMP front end          estimator/src/estimator/estimator.cpp:248-270
  kdtree_*_from_map->setInputCloud          estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:433-434
  the AddResidualBlock loops                estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:537-571
and the reference's own TYPES those lines are written against:
  pcl::PointXYZIWithCov, common::extractCov mloam_pcl/include/mloam_pcl/point_with_cov.hpp:43-112, 202-214
  common::Point .. PointICloud typedefs     mloam_common/libs/include/common/types/type.h:13-23
  cloudFeature, PointPlaneFeature, FeatureWithScore, ScanInfo    estimator/src/estimator/parameters.h:161-207
  class Pose + the members used here        estimator/src/estimator/pose.h:38-66; pose.cpp:16-41, 99-113

The cut text lives in tests/host/refcut/_gen/ only for the duration of the compile (never committed); the executable goes to tests/host/refcut/_build/
(git-ignored, travels to the GPU box like the built .so files). Needs /root/reference: on a box without it the prebuilt executable is used."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
REF = os.environ.get("MLOAM_REFERENCE", "/root/reference")
GEN = os.path.join(HERE, "_gen")
OUT = os.path.join(HERE, "_build")
EXE = os.path.join(OUT, "refcut_selftest")
CUTS = [
    ("point_with_cov.inc", "mloam_pcl/include/mloam_pcl/point_with_cov.hpp", 43, 112, "namespace pcl"),
    ("extract_cov.inc", "mloam_pcl/include/mloam_pcl/point_with_cov.hpp", 202, 214, "void extractCov"),
    ("common_types.inc", "mloam_common/libs/include/common/types/type.h", 13, 23, "namespace common"),
    ("parameter_types.inc", "estimator/src/estimator/parameters.h", 161, 207, "typedef std::map<std::string, common::PointICloud> cloudFeature;"),
    ("pose_class.inc", "estimator/src/estimator/pose.h", 38, 66, "class Pose"),
    ("pose_ctor_default.inc", "estimator/src/estimator/pose.cpp", 16, 23, "Pose::Pose()"),
    ("pose_ctor_copy.inc", "estimator/src/estimator/pose.cpp", 25, 32, "Pose::Pose(const Pose &pose)"),
    ("pose_ctor_qt.inc", "estimator/src/estimator/pose.cpp", 34, 41, "Pose::Pose(const Eigen::Quaterniond &q"),
    ("pose_inverse_update.inc", "estimator/src/estimator/pose.cpp", 99, 108, "Pose Pose::inverse() const"),
    ("pose_mul.inc", "estimator/src/estimator/pose.cpp", 110, 113, "Pose Pose::operator * (const Pose &pose)"),
    ("estimator_front_end.inc", "estimator/src/estimator/estimator.cpp", 248, 270, "std::vector<cloudFeature *> feature_frame_ptr(NUM_OF_LASER);"),
    ("set_input_cloud.inc", "estimator/src/lidarMapper/lidar_mapper_keyframe.cpp", 433, 434, "kdtree_surf_from_map->setInputCloud(laser_cloud_surf_from_map_cov_ds);"),
    ("add_residual_blocks.inc", "estimator/src/lidarMapper/lidar_mapper_keyframe.cpp", 537, 571, "for (const size_t &fid : sel_surf_feature_idx)"),
]
SOURCES = ["refcut_selftest.cpp", "build_refcut.py"]
FACADE = os.path.join(ROOT, "m-loam_amd", "host", "mloam_facade.hpp")


def compile_cmd(out, src, extra=()):
    lib = os.path.join(ROOT, "m-loam_amd", "lib")
    return ["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-invalid-offsetof", "-Wno-pedantic", "-Wno-mismatched-new-delete", "-Wno-sign-compare",
            "-DMLOAM_FACADE_USE_PCL_TYPES", "-DMLOAM_FACADE_CERES_BASES", "-I", os.path.join(ROOT, "tests", "host", "pcl_stub"), "-I", os.path.join(ROOT, "m-loam_amd", "host"),
            "-I", os.path.join(ROOT, "include"), *extra, "-o", out, src, "-L", lib, "-lmloam_hip", "-Wl,-rpath,$ORIGIN/../../../../m-loam_amd/lib", f"-Wl,-rpath,{lib}",
            "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]


def build(force=False, out=None, verbose=False):
    """-> path of the executable, or None when neither the reference tree nor a prebuilt executable exists"""
    exe = out or EXE
    if not os.path.isdir(os.path.join(REF, "estimator", "src")):
        return exe if os.path.exists(exe) else None
    deps = [os.path.join(HERE, f) for f in SOURCES] + [FACADE, os.path.join(ROOT, "include", "mloam_hip.h"), os.path.join(ROOT, "oracle", "ref", "mini_eigen.hpp")]
    stub = os.path.join(ROOT, "tests", "host", "pcl_stub")
    for d, _, fs in os.walk(stub):
        deps += [os.path.join(d, f) for f in fs]
    lib = os.path.join(ROOT, "m-loam_amd", "lib", "libmloam_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("libmloam_hip.so is not built (python m-loam_amd/build.py)")
    if not force and os.path.exists(exe) and all(os.path.getmtime(s) <= os.path.getmtime(exe) for s in deps):
        return exe
    os.makedirs(GEN, exist_ok=True)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    try:
        for name, rel, a, b, must in CUTS:
            lines = open(os.path.join(REF, rel)).read().split("\n")
            assert must in lines[a - 1], f"{rel}:{a} is not '{must}' -- the reference tree differs from the surveyed one"
            open(os.path.join(GEN, name), "w").write("\n".join(lines[a - 1:b]) + "\n")
        cmd = compile_cmd(exe, os.path.join(HERE, "refcut_selftest.cpp"))
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(GEN, ignore_errors=True)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
