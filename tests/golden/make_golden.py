#!/usr/bin/env python
"""Generates tests/golden/config1.npz from the CPU oracle on the seeded config-1 scene (16-ring scan, ~50k map).
The reference itself cannot run here (no Eigen/PCL/Ceres), so these are ORACLE outputs: a regression pin and the common
yardstick for the GPU parity tests, not reference-generated vectors. Run from the repo root: python tests/golden/make_golden.py"""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import conftest  # noqa: E402

synth = importlib.import_module("m-loam_amd.synth")
case = conftest._make_case(synth, "50k", 16, 1)
sc = case["scans"][0]
ex = O.extract(sc.points, sc.scan_start, sc.scan_end)
feats = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
ms, mc = O.Map(case["surf_map"]), O.Map(case["corner_map"])
vs, cs = ms.match("s", feats[0], case["p0"])
vc, cc = mc.match("c", feats[1], case["p0"])
r = O.scan2map(ms, mc, feats[0], feats[1], case["p0"], O.mapper_params())
g = O.gn_iterations(ms, mc, feats[0], feats[1], case["p0"], O.mapper_params(), 5)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config1.npz"),
                    points_sha=hashlib.sha256(np.ascontiguousarray(sc.points).tobytes()).hexdigest(),
                    label=ex["label"].astype(np.int8), less_sharp=ex["less_sharp"], flat=ex["flat"], sharp=ex["sharp"],
                    valid_surf=vs, valid_corner=vc, coeff_surf=cs.astype(np.float32), coeff_corner=cc.astype(np.float32),
                    scan2map_pose=r["pose"], gn5_pose=g["pose"])
print("wrote config1.npz", ex["n_ties"], vs.sum(), vc.sum(), r["pose"])
