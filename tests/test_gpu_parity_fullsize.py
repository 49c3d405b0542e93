"""Oracle-vs-HIP parity at BASELINE sizes: config 2 (2 x 64 rings vs the ~500k-point map), through the C-ABI,
against the LIVE oracle on the same seeded inputs -- labels / picked flags / index lists bit-exact per 64-ring scan and for the joint
128-ring upload, k-NN indices + f32 distances, correspondence validity + coefficient bits, H / g 1e-9, per-iteration counts and the
5-GN pose, scan2map's LM bookkeeping. Every match path of the library is held to the same oracle results: the fused single-launch
kernel (strided and consecutive feature assignment) and the two-kernel path (MLH_FUSED=0)."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pose_err(a, b):
    dt = float(np.linalg.norm(np.asarray(a[:3]) - np.asarray(b[:3])))
    qa, qb = np.asarray(a[3:7]), np.asarray(b[3:7])
    dr = 2.0 * float(np.arccos(min(1.0, abs(float(np.dot(qa, qb))))))
    return dt, dr


@pytest.fixture(scope="module")
def cfg2(synth, orc):
    """BASELINE config 2 workload exactly as bench.py builds it, features from the ORACLE's extraction."""
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
    ex = [orc.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    surf, corner = bench.fuse_features(synth, scans, ex)
    return dict(surf_map=surf_map, corner_map=corner_map, gt=gt, scans=scans, ex=ex, surf=surf, corner=corner,
                p0=synth.perturbed_pose(gt, seed=43), oms=orc.Map(surf_map), omc=orc.Map(corner_map))


def _ctx_with_env(mla, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return mla.Context(0)          # the switches are read at mlh_create
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


MODES = {"fused_strided": dict(MLH_FUSED=1, MLH_FUSED_STRIDED=1), "fused_consecutive": dict(MLH_FUSED=1, MLH_FUSED_STRIDED=0),
         "two_kernel": dict(MLH_FUSED=0)}


@pytest.fixture(scope="module", params=list(MODES))
def staged(request, mla, cfg2):
    c = _ctx_with_env(mla, **MODES[request.param])
    c.map_set(mla.SURF, cfg2["surf_map"])
    c.map_set(mla.CORNER, cfg2["corner_map"])
    c.features_set(mla.SURF, cfg2["surf"])
    c.features_set(mla.CORNER, cfg2["corner"])
    yield c
    c.close()


def test_extract_labels_bit_exact_64_rings(mla, orc, cfg2):
    """feature labels bit-exact (north star) for both 64-ring scans and for the joint 128-ring upload."""
    c = mla.Context(0)
    scans, ex = cfg2["scans"], cfg2["ex"]
    for s, ref in zip(scans, ex):
        assert ref["n_ties"] == 0
        got = c.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
        assert np.array_equal(got["curvature"].view(np.uint32), ref["curvature"].view(np.uint32))
        for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw"):
            assert np.array_equal(got[k], ref[k]), k
        assert got["less_flat_ds"].shape == ref["less_flat_ds"].shape
        np.testing.assert_allclose(got["less_flat_ds"][:, :3], ref["less_flat_ds"][:, :3], atol=2e-6)   # centroid sum order (tolerance, not bits)
    offs = np.cumsum([0] + [len(s.points) for s in scans])
    pts = np.concatenate([s.points for s in scans])
    st = np.concatenate([s.scan_start + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
    en = np.concatenate([s.scan_end + offs[i] for i, s in enumerate(scans)]).astype(np.int32)
    both = c.extract(pts, st, en)
    assert np.array_equal(both["label"], np.concatenate([e["label"] for e in ex]))
    assert np.array_equal(both["picked"], np.concatenate([e["picked"] for e in ex]))
    for k in ("sharp", "less_sharp", "flat", "less_flat_raw"):
        assert np.array_equal(both[k], np.concatenate([e[k] + offs[i] for i, e in enumerate(ex)])), k
    c.close()


def test_knn_exact_full_size(mla, cfg2):
    """k-NN indices + f32 distances at the features' map-frame positions (inside the acceptance radius: exact)."""
    c = mla.Context(0)
    Tm = None
    import importlib
    synth = importlib.import_module("m-loam_amd.synth")
    Tm = synth.pose_to_mat(cfg2["p0"])
    for kind, feats, mp, om in ((mla.SURF, cfg2["surf"], cfg2["surf_map"], cfg2["oms"]), (mla.CORNER, cfg2["corner"], cfg2["corner_map"], cfg2["omc"])):
        c.map_set(kind, mp)
        q = synth.transform_points(feats[:, :3], Tm)
        idx, d2 = c.knn(kind, q)
        ridx, rd2 = om.knn(q)
        inside = rd2[:, 4] < 1.0
        assert inside.sum() > 0.3 * len(q)
        assert np.array_equal(d2[inside].view(np.uint32), rd2[inside].view(np.uint32))
        assert np.array_equal(idx[inside], ridx[inside])
        assert np.all(d2[~inside][:, 4] >= 1.0)
    c.close()


def test_match_and_linearise_full_size(staged, mla, orc, cfg2):
    """validity + coefficient bits for every feature, residuals / Jacobians / H / g at 1e-9, zero decision flips."""
    flips = 0
    for kind, ch, feats, om in ((mla.SURF, "s", cfg2["surf"], cfg2["oms"]), (mla.CORNER, "c", cfg2["corner"], cfg2["omc"])):
        got = staged.match_linearize(kind, cfg2["p0"])
        v, co = om.match(ch, feats, cfg2["p0"])
        flips += int(np.sum(got["valid"] != v))
        assert np.array_equal(got["valid"], v), f"{int(np.sum(got['valid'] != v))} decision flips ({ch})"
        m = v.astype(bool)
        assert np.array_equal(got["coeffs"][m].astype(np.float32).view(np.uint32), co[m].astype(np.float32).view(np.uint32))
        ref = orc.linearize(ch, feats, None, cfg2["p0"], v, co)
        np.testing.assert_allclose(got["r"][m], ref["r"][m], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(got["J"][m], ref["J"][m], rtol=1e-9, atol=1e-11)
        sc = max(1.0, float(np.abs(ref["H"]).max()))
        assert float(np.abs(got["H"] - ref["H"]).max()) <= 1e-9 * sc
        assert float(np.abs(got["g"] - ref["g"]).max()) <= 1e-9 * max(1.0, float(np.abs(ref["g"]).max()))
        assert got["count"] == ref["count"] and abs(got["cost"] - ref["cost"]) <= 1e-9 * max(1.0, ref["cost"])
    assert flips == 0


def test_gn5_full_size(staged, orc, cfg2):
    """per-iteration (n_surf, n_corner), H, g and the pose after 5 Gauss-Newton iterations."""
    pose, stats = staged.gn_solve(cfg2["p0"], 5)
    ref = orc.gn_iterations(cfg2["oms"], cfg2["omc"], cfg2["surf"], cfg2["corner"], cfg2["p0"], orc.mapper_params(), 5)
    for s, r in zip(stats, ref["iters"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"])
        assert float(np.abs(s["H"] - r["H"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["H"]).max()))
        assert float(np.abs(s["g"] - r["g"]).max()) <= 1e-9 * max(1.0, float(np.abs(r["g"]).max()))
        assert s["is_degenerate"] == r["is_degenerate"]
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)                      # north star: 1e-4 m / 1e-4 rad
    pose2, _ = staged.gn_solve(cfg2["p0"], 5, want_stats=False)   # the path bench.py times
    dt, dr = _pose_err(pose2, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_scan2map_full_size(staged, orc, cfg2):
    """the reference's per-frame call: LM iteration counts, successful steps, termination, costs, pose."""
    pose, stats = staged.scan2map(cfg2["p0"])
    ref = orc.scan2map(cfg2["oms"], cfg2["omc"], cfg2["surf"], cfg2["corner"], cfg2["p0"], orc.mapper_params())
    for s, r in zip(stats, ref["outer"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf_sel"], r["n_corner_sel"])
        assert (s["lm_iterations"], s["successful_steps"], s["termination"]) == (r["lm_iterations"], r["successful_steps"], r["termination"])
        assert s["is_degenerate"] == r["is_degenerate"]
        assert abs(s["cost"] - r["initial_cost"]) <= 1e-9 * max(1.0, r["initial_cost"])
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)
    pose2, _ = staged.scan2map(cfg2["p0"], want_stats=False)
    dt, dr = _pose_err(pose2, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)
