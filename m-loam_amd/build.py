"""Builds libmloam_hip.so (hand-written HIP kernels + C-ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the resulting m-loam_amd/lib/libmloam_hip.so travels with the repo snapshot to the
GPU box. Flags: -O3, -ffp-contract=off (f32 decisions must be bit-identical with the reference's un-fused x86 path).
"""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libmloam_hip.so")
SOURCES = ["capi.hip", "grid.hip", "match.hip", "solver.hip", "extract.hip", "comm.hip", "select.hip", "voxel.hip", "voxelgrid.hip", "odom.hip", "track.hip",
           "frontend.hip", "segment.hip", "stdsort.hip"]
HEADERS = ["ctx.hpp", "alive_pool.hpp", "dev_math.hpp", "solver_dev.hpp", "knn_dev.hpp", "reduce_dev.hpp", "sort_dev.hpp", "stdsort_dev.hpp", "std_sort_mt.hpp", "p2p_dev.hpp", os.path.join("..", "..", "include", "mloam_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _newer(src: str, dst: str) -> bool:
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
