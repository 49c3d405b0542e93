"""Shared fixtures. GPU tests are marked @pytest.mark.gpu and call the HIP path through the C-ABI; everything else runs
on CPU (oracle vs golden vectors / independent numpy-scipy cross-checks, host logic, ABI symbol checks)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


# The strongest pins compare against the reference's OWN lines: oracle/_ref/libmloam_ref.so and tests/host/refcut/_build/refcut_selftest, built where
# /root/reference exists and travelling to the GPU box as built files. A box where they are expected but absent must FAIL those tests, not skip them
# (VERDICT r04): MLOAM_REQUIRE_REF=1 turns every skip whose reason is the missing reference build into a failure. Default: required wherever the reference tree
# is mounted, and wherever the built product library travelled (then the checkers built next to it should have travelled too); MLOAM_REQUIRE_REF=0 opts out
# (a fresh clone that only built the product).
_REF_SKIP = ("reference build", "libmloam_ref.so", "refcut_selftest", "no reference tree")


def _require_ref():
    v = os.environ.get("MLOAM_REQUIRE_REF")
    if v is not None:
        return v == "1"
    return os.path.isdir("/root/reference") or os.path.exists(os.path.join(ROOT, "m-loam_amd", "lib", "libmloam_hip.so"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _require_ref():
        os.environ["MLOAM_REQUIRE_REF"] = "1"


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.skipped and os.environ.get("MLOAM_REQUIRE_REF") == "1":
        reason = str(rep.longrepr[2] if isinstance(rep.longrepr, tuple) else rep.longrepr)
        # (the one legitimate skip that names the tree: a CPU-only compile check that needs the SOURCE tree, on a box that has the prebuilt executable instead)
        if any(k in reason for k in _REF_SKIP) and "exercised by the GPU test" not in reason:
            rep.outcome = "failed"
            rep.longrepr = f"MLOAM_REQUIRE_REF=1: the reference-built checker is expected on this box but missing -- would have skipped: {reason}"


@pytest.fixture(scope="session")
def mla():
    # tests that hand torch device tensors to the library need torch's HIP context to exist before the library creates its own
    # (whatever the order the tests are selected in)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    return importlib.import_module("m-loam_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("m-loam_amd.synth")


@pytest.fixture(scope="session")
def orc():
    import oracle as O
    O.build()
    return O


def _make_case(synth, preset, n_rings, n_lidars, seed=42):
    """Scene + maps + per-LiDAR scans + (oracle-free) feature clouds for the mapper."""
    sc = synth.make_scene(seed=seed, **synth.SCENE_PRESETS[preset])
    surf_map, corner_map = synth.sample_maps(sc, seed=seed, kf_rings=n_rings, kf_lidars=n_lidars)
    gt = synth.gt_body_pose()
    scans = [synth.simulate_scan(sc, gt, synth.HERCULES_BODY_T_LASER[i], n_rings, seed=7 + i) for i in range(n_lidars)]
    return dict(scene=sc, surf_map=surf_map, corner_map=corner_map, gt=gt, scans=scans, p0=synth.perturbed_pose(gt, seed=43))


@pytest.fixture(scope="session")
def case16(synth):
    """BASELINE config 1: one 16-ring x 1800 scan, ~50k-point planar map."""
    return _make_case(synth, "50k", 16, 1)


def features_from_extraction(synth, scans, extract_fn):
    """Fuse per-LiDAR extraction results into the mapper's two feature clouds (reference-LiDAR frame, intensity = LiDAR id,
    visualization.cpp:40-52), then thin them at MAP_SURF_RES / MAP_CORNER_RES as downsampleCurrentScan does."""
    surf, corner = [], []
    for i, sc in enumerate(scans):
        ex = extract_fn(sc)
        T = np.eye(4)
        T[:3, :3] = synth.quat_to_rot(synth.HERCULES_BODY_T_LASER[i][:4])
        T[:3, 3] = synth.HERCULES_BODY_T_LASER[i][4:7]
        cpts = sc.points[ex["less_sharp"]][:, :3]
        spts = ex["less_flat_ds"][:, :3] if "less_flat_ds" in ex else sc.points[ex["less_flat_raw"]][:, :3]
        c = np.zeros((len(cpts), 4), np.float32)
        c[:, :3] = synth.transform_points(cpts, T)
        c[:, 3] = i
        s = np.zeros((len(spts), 4), np.float32)
        s[:, :3] = synth.transform_points(spts, T)
        s[:, 3] = i
        surf.append(s)
        corner.append(c)
    surf = synth.voxel_mean(np.concatenate(surf), 0.4)
    corner = synth.voxel_mean(np.concatenate(corner), 0.2)
    surf[:, 3] = np.round(surf[:, 3])
    corner[:, 3] = np.round(corner[:, 3])
    return np.ascontiguousarray(surf), np.ascontiguousarray(corner)


@pytest.fixture(scope="session")
def feats16(synth, orc, case16):
    return features_from_extraction(synth, case16["scans"], lambda s: orc.extract(s.points, s.scan_start, s.scan_end))


@pytest.fixture(scope="session")
def track_case(synth, orc):
    return _track_case(synth, orc)


def _track_case(synth, orc):
    """Two consecutive 16-ring scans of the 50k scene (intensity = ring id, as ImageSegmenter leaves it) and their LOAM features:
    prev = less-sharp corners / voxel-thinned less-flat surfs, cur = sharp corners / flat surfs (lidar_tracker.cpp:30-38)."""
    sc = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    gt0 = synth.gt_body_pose()
    motion = np.array([0.35, -0.12, 0.02, 0.0, 0.0, np.sin(np.deg2rad(1.5) / 2), np.cos(np.deg2rad(1.5) / 2)])
    T0 = synth.pose_to_mat(gt0)
    T1 = T0 @ synth.pose_to_mat(motion)
    gt1 = np.concatenate([T1[:3, 3], synth.rot_to_quat(T1[:3, :3])]) if hasattr(synth, "rot_to_quat") else None
    if gt1 is None:
        from scipy.spatial.transform import Rotation as Rot
        gt1 = np.concatenate([T1[:3, 3], Rot.from_matrix(T1[:3, :3]).as_quat()])
    out = {}
    for name, pose, seed in (("prev", gt0, 7), ("cur", gt1, 11)):
        scn = synth.simulate_scan(sc, pose, synth.HERCULES_BODY_T_LASER[0], 16, seed=seed)
        ring = np.zeros(len(scn.points), np.float32)
        begins = scn.scan_start - 5
        for r in range(scn.n_rings):
            e = begins[r + 1] if r + 1 < scn.n_rings else len(scn.points)
            ring[begins[r]:e] = r
        scn.points[:, 3] = ring
        ex = orc.extract(scn.points, scn.scan_start, scn.scan_end)
        out[name] = dict(scan=scn, ex=ex)
    pts0, ex0 = out["prev"]["scan"].points, out["prev"]["ex"]
    pts1, ex1 = out["cur"]["scan"].points, out["cur"]["ex"]
    return dict(corner_last=np.ascontiguousarray(pts0[ex0["less_sharp"]]), surf_last=np.ascontiguousarray(ex0["less_flat_ds"][:, :4]),
                corner_sharp=np.ascontiguousarray(pts1[ex1["sharp"]]), surf_flat=np.ascontiguousarray(pts1[ex1["flat"]]),
                motion=motion, scans=(out["prev"]["scan"], out["cur"]["scan"]))


def rows_f3_inputs():
    """seeded inputs of tests/golden/rows_f3.npz (front-end rows: transformCloudFeature, TransformToEnd)"""
    rng = np.random.default_rng(31)
    pts = np.concatenate([rng.uniform(-60, 60, (2000, 3)), (rng.integers(0, 64, (2000, 1)) + rng.uniform(0, 0.0999, (2000, 1)))], axis=1).astype(np.float32)
    q = np.array([0.012, -0.02, 0.031, 1.0]); q /= np.linalg.norm(q)
    pose = np.concatenate([[0.35, -0.12, 0.02], q])
    ext = np.array([0.1, -0.5, 0.02, 0.0, 0.0, 0.0998334166468, 0.995004165278])
    return pts, pose, ext


def make_window_case(synth, orc, n_frames=2, n_lidars=2, seed=3, n_rings=16, preset="50k"):
    """A sliding window as Estimator::optimizeMap sees it: a pivot pose, n_frames later poses, n_lidars extrinsics, and for every (frame, LiDAR)
    the features of that scan matched against the local map expressed in the pivot frame (buildLocalMap, estimator.cpp:1160-1268) ->
    the LidarPureOdom factor table (point in the LiDAR frame, plane / line coefficients in the pivot frame, block indices)."""
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    case = _make_case(synth, preset, n_rings, n_lidars)
    to_pose = lambda T: np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
    T_piv = synth.pose_to_mat(case["gt"])
    Tinv = np.linalg.inv(T_piv)
    maps = [synth.transform_points(m[:, :3], Tinv) for m in (case["surf_map"], case["corner_map"])]       # local map in the pivot frame
    oms, omc = orc.Map(maps[0]), orc.Map(maps[1])
    exts_T = []
    for n in range(n_lidars):
        r = synth.HERCULES_BODY_T_LASER[n]
        exts_T.append(synth.pose_to_mat(np.concatenate([r[4:7], r[:4] / np.linalg.norm(r[:4])])))
    frames_T = []
    for i in range(n_frames):
        d = np.eye(4)
        d[:3, :3] = Rot.from_rotvec(np.deg2rad([0.3, -0.2, 1.0 + i])).as_matrix()
        d[:3, 3] = [0.4 * (i + 1), 0.05 * i, 0.01]
        frames_T.append(T_piv @ d)
    types, points, coeffs, fi, ei = [], [], [], [], []
    for i, T_i in enumerate(frames_T):
        for n in range(n_lidars):
            scn = synth.simulate_scan(case["scene"], to_pose(T_i), synth.HERCULES_BODY_T_LASER[n], n_rings, seed=100 + 10 * i + n)
            ex = orc.extract(scn.points, scn.scan_start, scn.scan_end)
            rel = to_pose(Tinv @ T_i @ exts_T[n])
            rel = synth.perturbed_pose(rel, seed=200 + 10 * i + n, dt=0.05, drot_deg=0.5)
            for kind, om, f in (("s", oms, synth.voxel_mean(ex["less_flat_ds"][:, :3].copy(), 0.4)), ("c", omc, scn.points[ex["less_sharp"]][:, :3])):
                f4 = np.zeros((len(f), 4), np.float32)
                f4[:, :3] = f
                v, co = om.match(kind, f4, rel)
                m = v.astype(bool)
                types.append(np.full(m.sum(), 0 if kind == "s" else 1, np.int32))
                points.append(f4[m, :3].astype(np.float64))
                coeffs.append(co[m])
                fi.append(np.full(m.sum(), i, np.int32))
                ei.append(np.full(m.sum(), n, np.int32))
    perm = rng.permutation(sum(len(t) for t in types))       # the table arrives in no particular order
    cat = lambda a: np.concatenate(a)[perm]
    pert = lambda T, k: synth.perturbed_pose(to_pose(T), seed=300 + k, dt=0.03, drot_deg=0.3)
    return dict(types=cat(types), points=cat(points), coeffs=cat(coeffs), fi=cat(fi), ei=cat(ei),
                pivot=to_pose(T_piv), frames=np.stack([pert(T, k) for k, T in enumerate(frames_T)]),
                exts=np.stack([to_pose(exts_T[0])] + [pert(T, 10 + k) for k, T in enumerate(exts_T[1:])]))
