"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Bit-exact for integer / index / f32-decision outputs; 1e-9 relative for f64 residuals, Jacobians and normal equations
(summation order differs); pose within the north-star tolerance 1e-4 m / 1e-4 rad (observed ~1e-10)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(mla):
    c = mla.Context(0)
    yield c
    c.close()


def _pose_err(a, b):
    dt = np.linalg.norm(a[:3] - b[:3])
    qa, qb = a[3:7] / np.linalg.norm(a[3:7]), b[3:7] / np.linalg.norm(b[3:7])
    dr = 2.0 * np.arccos(min(1.0, abs(float(qa @ qb))))
    return dt, dr


def test_extract_labels_bit_exact_16(ctx, orc, case16):
    sc = case16["scans"][0]
    ref = orc.extract(sc.points, sc.scan_start, sc.scan_end)
    assert ref["n_ties"] == 0, "generator produced curvature ties: std::sort order would be implementation-defined"
    got = ctx.extract(sc.points, sc.scan_start, sc.scan_end)
    assert np.array_equal(got["curvature"].view(np.uint32), ref["curvature"].view(np.uint32))
    assert np.array_equal(got["label"], ref["label"])
    assert np.array_equal(got["picked"], ref["picked"])
    for k in ("sharp", "less_sharp", "flat", "less_flat_raw"):
        assert np.array_equal(got[k], ref[k]), k


def test_knn_exact(ctx, mla, orc, case16, feats16):
    surf_map = case16["surf_map"]
    ctx.map_set(mla.SURF, surf_map)
    rng = np.random.default_rng(0)
    q = surf_map[rng.integers(0, len(surf_map), 4000)] + rng.normal(0, 0.3, (4000, 3)).astype(np.float32)
    q = np.ascontiguousarray(q, np.float32)
    idx, d2 = ctx.knn(mla.SURF, q)
    om = orc.Map(surf_map)
    ridx, rd2 = om.knn(q, 5)
    within = rd2 < 1.0   # the grid search is exact inside the acceptance radius
    assert np.array_equal(idx[within], ridx[within])
    assert np.array_equal(d2[within].view(np.uint32), rd2[within].view(np.uint32))
    # beyond the radius the 27-cell search may miss points: it must then report >= 1.0 (never a false accept)
    assert np.all(d2[~within] >= 1.0)
    # queries with FEWER than five map points in their 27 cells (found by scripts/soak_api.py: the matching kernels' early exit used to answer "nothing" here
    # although a neighbour sat 0.67 m away): the corner features against the sparse corner map, and queries off the edge of the map's box
    corner_map = case16["corner_map"]
    ctx.map_set(mla.CORNER, corner_map)
    qc = np.concatenate([feats16[1][:, :3], corner_map[::40, :3] + np.float32([0.0, 0.0, -0.6]), corner_map[::40, :3] + np.float32([0.7, 0.0, 0.0])]).astype(np.float32)
    qc = np.ascontiguousarray(qc)
    idx, d2 = ctx.knn(mla.CORNER, qc)
    ridx, rd2 = orc.Map(corner_map).knn(qc, 5)
    within = rd2 < 1.0
    assert (within.sum(axis=1) < 5).sum() > 100 and ((within.sum(axis=1) > 0) & (within.sum(axis=1) < 5)).sum() > 50
    assert np.array_equal(idx[within], ridx[within])
    assert np.array_equal(d2[within].view(np.uint32), rd2[within].view(np.uint32))
    assert np.all(d2[~within] >= 1.0)


@pytest.mark.parametrize("kind_name", ["surf", "corner"])
def test_match_linearize_parity(ctx, mla, orc, case16, feats16, kind_name):
    kind = mla.SURF if kind_name == "surf" else mla.CORNER
    ch = "s" if kind_name == "surf" else "c"
    cloud = case16["surf_map"] if kind_name == "surf" else case16["corner_map"]
    feats = feats16[0] if kind_name == "surf" else feats16[1]
    p0 = case16["p0"]
    ctx.map_set(kind, cloud)
    ctx.features_set(kind, feats)
    got = ctx.match_linearize(kind, p0)
    om = orc.Map(cloud)
    valid, coeffs = om.match(ch, feats, p0)
    assert valid.sum() > 50
    assert np.array_equal(got["valid"], valid)
    assert np.array_equal(got["coeffs"].astype(np.float32).view(np.uint32), coeffs.astype(np.float32).view(np.uint32))
    ref = orc.linearize(ch, feats, np.full(len(feats), 0.0075), p0, valid, coeffs)
    np.testing.assert_allclose(got["r"], ref["r"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["J"], ref["J"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(got["g"], ref["g"], rtol=1e-8, atol=1e-8)
    assert got["count"] == ref["count"]
    assert abs(got["cost"] - ref["cost"]) <= 1e-9 * max(1.0, ref["cost"])
    # re-linearise the stored correspondences at another pose
    p1 = p0.copy()
    p1[:3] += [0.01, -0.02, 0.005]
    got2 = ctx.linearize(kind, p1)
    ref2 = orc.linearize(ch, feats, np.full(len(feats), 0.0075), p1, valid, coeffs)
    np.testing.assert_allclose(got2["r"], ref2["r"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got2["H"], ref2["H"], rtol=1e-9, atol=1e-7)


def _stage(ctx, mla, case, feats):
    ctx.map_set(mla.SURF, case["surf_map"])
    ctx.map_set(mla.CORNER, case["corner_map"])
    ctx.features_set(mla.SURF, feats[0])
    ctx.features_set(mla.CORNER, feats[1])


def test_gn_solve_parity(ctx, mla, orc, case16, feats16):
    _stage(ctx, mla, case16, feats16)
    pose, stats = ctx.gn_solve(case16["p0"], 5)
    ref = orc.gn_iterations(orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"]), feats16[0], feats16[1], case16["p0"],
                            orc.mapper_params(), 5)
    for s, r in zip(stats, ref["iters"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"])
        np.testing.assert_allclose(s["H"], r["H"], rtol=1e-8, atol=1e-6)
        dt, dr = _pose_err(s["pose_after"], r["pose_after"])
        assert dt < 1e-4 and dr < 1e-4
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)
    gdt, gdr = _pose_err(pose, case16["gt"])
    assert gdt < 0.05 and gdr < 0.01


@pytest.mark.parametrize("n_iters", [1, 2, 3, 5])
def test_gn_schedule_variants_agree_bit_for_bit(mla, orc, case16, feats16, n_iters):
    """mlh_set_gn_schedule: the finish done by the next iteration's correspondence launch (every workgroup sums the tiles' records in the fixed order and
    solves for itself) and the search bounded by the previous iteration's neighbours change where work happens, not what is computed: the four combinations,
    the statistics path (always the classic finish) and the split / chained submissions give the SAME pose bits; the oracle agrees to 1e-7."""
    ref = orc.gn_iterations(orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"]), feats16[0], feats16[1], case16["p0"], orc.mapper_params(), n_iters)
    poses = {}
    for defer, warm, final in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)):
        if True:
            c = mla.Context(0)
            try:
                c.set_gn_schedule(defer, warm, final)
                _stage(c, mla, case16, feats16)
                p_plain = c.gn_solve(case16["p0"], n_iters, want_stats=False)[0]
                p_stats, st = c.gn_solve(case16["p0"], n_iters, want_stats=True)
                c.gn_solve_begin(case16["p0"], n_iters)
                p_split = c.gn_solve_end()
                # chained: the start pose is made on the device from the pose the previous solve left there; identity odometry -> the same frame again
                ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
                c.gn_solve_begin_chained(ident, ident, n_iters)
                p_chain = c.gn_solve_end()
                c.gn_solve_begin(p_split, n_iters)
                p_restart = c.gn_solve_end()
            finally:
                c.close()
            assert np.array_equal(p_plain, p_stats) and np.array_equal(p_plain, p_split), (defer, warm, final)
            # the chained solve starts from Pose(q, t) products of the previous result (normalisations): equal to a restart from that result to rounding
            assert float(np.abs(p_chain - p_restart).max()) < 1e-12, (defer, warm, final)
            assert [(x["n_surf"], x["n_corner"]) for x in st] == [(r["n_surf"], r["n_corner"]) for r in ref["iters"]]
            poses[(defer, warm, final)] = (p_plain, p_chain)
    for k, v in poses.items():
        assert np.array_equal(v[0], poses[(0, 0, 0)][0]) and np.array_equal(v[1], poses[(0, 0, 0)][1]), k
    dt, dr = _pose_err(poses[(1, 1, 1)][0], ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_gn_last_iteration_completed_by_the_successor_or_by_whoever_comes_next(mla, case16, feats16):
    """final_in_successor: a submitted solve leaves its LAST iteration as records. Whoever comes next completes it -- a chained successor in its first launch (two
    frames in flight, both poses bit-equal to the classic schedule), mlh_gn_solve_end by itself, or any other solver call (a synchronous solve, scan2map, a new
    unchained submission, a feature set with MORE tiles than the record buffer holds) -- and the pose is the classic one every time."""
    p0 = case16["p0"]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    ref = {}
    for final in (0, 1):
        c = mla.Context(0)
        try:
            c.set_gn_schedule(1, 1, final)
            _stage(c, mla, case16, feats16)
            out = {}
            # three frames pipelined: k + 1 submitted (chained) before k is collected
            c.gn_solve_begin(p0, 5)
            c.gn_solve_begin_chained(ident, ident, 5)
            a = c.gn_solve_end()
            c.gn_solve_begin_chained(ident, ident, 5)
            b = c.gn_solve_end()
            d = c.gn_solve_end()
            out["pipeline"] = np.stack([a, b, d])
            # collected with nothing chained behind it
            c.gn_solve_begin(p0, 5)
            out["alone"] = c.gn_solve_end()
            # a synchronous solve / scan2map / an unchained submission arrive while the last iteration is still pending
            c.gn_solve_begin(p0, 5)
            out["sync_after"] = c.gn_solve(out["alone"], 2, want_stats=False)[0]
            out["pending_1"] = c.gn_solve_end()
            c.gn_solve_begin(p0, 5)
            out["s2m_after"] = c.scan2map(p0, want_stats=False)[0]
            out["pending_2"] = c.gn_solve_end()
            c.gn_solve_begin(p0, 5)
            c.gn_solve_begin(out["alone"], 3)
            out["pending_3"] = c.gn_solve_end()
            out["unchained_after"] = c.gn_solve_end()
            # the next frame has many more features than the record buffer was sized for: the records must not be lost to the reallocation
            c.gn_solve_begin(p0, 5)
            big_s = np.ascontiguousarray(np.tile(feats16[0], (12, 1)))
            big_c = np.ascontiguousarray(np.tile(feats16[1], (12, 1)))
            c.features_set(mla.SURF, big_s); c.features_set(mla.CORNER, big_c)
            c.gn_solve_begin_chained(ident, ident, 2)
            out["pending_4"] = c.gn_solve_end()
            out["big_after"] = c.gn_solve_end()
            ref[final] = out
        finally:
            c.close()
    for k in ref[0]:
        assert np.array_equal(ref[0][k], ref[1][k]), k
    for k in ("pending_1", "pending_2", "pending_3", "pending_4"):
        assert np.array_equal(ref[1][k], ref[1]["alone"]), k


def test_host_driven_calls_while_a_solve_is_pending(mla, case16, feats16):
    """with final_in_successor a submitted solve's last iteration is a set of tile records and its pose sits in SolverState::xi[slot]. The
    host-driven entry points (mlh_match_linearize, mlh_linearize, mlh_good_feature_matching, mlh_pure_odom_add_matches[_gf]) begin by uploading THEIR pose with a
    launch that zeroes the whole state -- so the pending solve has to be completed first (upload_pose flushes it). begin -> such a call -> end: the pending
    solve's pose and the interleaved call's outputs both equal what the classic schedule (final_in_successor = 0) gives."""
    p0 = case16["p0"]
    other = p0.copy(); other[:3] += np.array([0.05, -0.03, 0.02])
    ref = {}
    for final in (0, 1):
        c = mla.Context(0)
        try:
            c.set_gn_schedule(1, 1, final)
            _stage(c, mla, case16, feats16)
            out = {}
            c.gn_solve_begin(p0, 5)
            m = c.match_linearize(mla.SURF, other)
            out["ml_H"], out["ml_g"], out["ml_valid"] = m["H"], m["g"], m["valid"]
            out["pend_ml"] = c.gn_solve_end()
            c.gn_solve_begin(p0, 5)
            l = c.linearize(mla.SURF, other)
            out["lin_H"], out["lin_g"] = l["H"], l["g"]
            out["pend_lin"] = c.gn_solve_end()
            c.gn_solve_begin(p0, 5)
            g = c.good_feature_matching(mla.CORNER, other, gf_method="rnd", gf_ratio=0.5, seed=3)
            out["gf_sel"], out["gf_H"] = g["sel"], g["H"]
            out["pend_gf"] = c.gn_solve_end()
            c.gn_solve_begin(p0, 5)
            c.pure_odom_begin()
            c.pure_odom_add_matches(mla.SURF, other, 0, 0)
            out["pend_odom"] = c.gn_solve_end()
            c.gn_solve_begin(p0, 5)
            out["odom_gf_sel"] = c.pure_odom_add_matches_gf(mla.SURF, other, other, other, np.array([0, 0, 0, 0, 0, 0, 1.0]), 0, 0, gf_ratio=0.8, seed=5)
            out["pend_odom_gf"] = c.gn_solve_end()
            # and the solve that follows starts from a sane state
            out["after"] = c.gn_solve(p0, 5, want_stats=False)[0]
            ref[final] = out
        finally:
            c.close()
    for k in ref[0]:
        assert np.array_equal(ref[0][k], ref[1][k]), k
    for k in ("pend_ml", "pend_lin", "pend_gf", "pend_odom", "pend_odom_gf", "after"):
        assert np.array_equal(ref[1][k], ref[1]["after"]), k
        assert np.all(np.isfinite(ref[1][k])) and abs(np.linalg.norm(ref[1][k][3:]) - 1.0) < 1e-12, k


def test_maps_staged_twice_beside_one_solve(mla, case16, feats16):
    """mlh_map_set_pair_overlapped twice while ONE submitted solve is uncollected: the second call's target is the set that solve reads, so it may only be
    rewritten behind it (scripts/soak_schedule.py found the index rebuilt under a correspondence launch, a rare 1e-5 m difference). A long solve (12 iterations on
    12 x the features) keeps the launch window open; every pose equals the synchronous one, and the next frame on the re-staged maps too."""
    p0 = case16["p0"]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    big_s = np.ascontiguousarray(np.tile(feats16[0], (12, 1)))
    big_c = np.ascontiguousarray(np.tile(feats16[1], (12, 1)))
    import torch
    d_sm, d_cm = torch.from_numpy(case16["surf_map"]).cuda(), torch.from_numpy(case16["corner_map"]).cuda()
    for sched in ((1, 1, 1), (0, 0, 0)):
        c = mla.Context(0)
        try:
            c.set_gn_schedule(*sched)
            c.map_set_pair(d_sm, d_cm)
            c.features_set(mla.SURF, big_s); c.features_set(mla.CORNER, big_c)
            want = c.gn_solve(p0, 12, want_stats=False)[0]
            want2 = c.gn_solve(want, 3, want_stats=False)[0]
            for _ in range(20):
                c.gn_solve_begin(p0, 12)
                c.map_set_pair_overlapped(d_sm, d_cm)
                c.map_set_pair_overlapped(d_sm, d_cm)          # back into the set the solve in flight reads
                c.map_set_pair_overlapped(d_sm, d_cm)
                got = c.gn_solve_end()
                assert np.array_equal(got, want), sched
                c.gn_solve_begin(got, 3)
                c.map_set_pair_overlapped(d_sm, d_cm)
                c.map_set_pair_overlapped(d_sm, d_cm)
                assert np.array_equal(c.gn_solve_end(), want2), sched
        finally:
            c.close()


def test_gn_deferred_finish_on_a_degenerate_problem(mla, orc, synth):
    """the deferred finish's slow path: a map that is ONE plane leaves three directions unconstrained (eigenvalues below MAP_EIG_THRE), so every workgroup of the
    next correspondence launch runs evalDegenracy's projection itself -- same bits as the classic finish, and the oracle's degenerate update"""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(np.arange(-12, 12, 0.4), np.arange(-12, 12, 0.4)), -1).reshape(-1, 2)
    plane = np.concatenate([g + rng.uniform(-0.05, 0.05, g.shape), rng.normal(0, 0.005, (len(g), 1))], 1).astype(np.float32)
    line = np.stack([np.full(200, 3.0), np.full(200, 2.0), np.linspace(0, 4, 200)], 1).astype(np.float32) + rng.normal(0, 0.003, (200, 3)).astype(np.float32)
    feats_s = np.concatenate([rng.uniform(-8, 8, (3000, 2)), np.zeros((3000, 1))], 1).astype(np.float32)
    feats_s = np.concatenate([feats_s, np.zeros((3000, 1), np.float32)], 1)
    feats_c = np.concatenate([line[::4] + np.float32(0.01), np.zeros((50, 1), np.float32)], 1)
    p0 = np.array([0.0, 0.0, 0.08, 0.004, -0.003, 0.0, 1.0]); p0[3:] /= np.linalg.norm(p0[3:])
    ref = orc.gn_iterations(orc.Map(plane), orc.Map(line), feats_s, feats_c, p0, orc.mapper_params(), 4)
    assert any(r["is_degenerate"] for r in ref["iters"])
    out = {}
    for defer in (0, 1):
        c = mla.Context(0)
        try:
            c.set_gn_schedule(defer, 1)
            c.map_set_pair(plane, line)
            c.features_set(mla.SURF, feats_s)
            c.features_set(mla.CORNER, feats_c)
            out[defer] = c.gn_solve(p0, 4, want_stats=False)[0]
            _, st = c.gn_solve(p0, 4, want_stats=True)
        finally:
            c.close()
        assert [x["is_degenerate"] for x in st] == [r["is_degenerate"] for r in ref["iters"]]
    assert np.array_equal(out[0], out[1])
    dt, dr = _pose_err(out[1], ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_scan2map_parity(ctx, mla, orc, case16, feats16):
    _stage(ctx, mla, case16, feats16)
    pose, stats = ctx.scan2map(case16["p0"])
    ref = orc.scan2map(orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"]), feats16[0], feats16[1], case16["p0"],
                       orc.mapper_params())
    for s, r in zip(stats, ref["outer"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf_sel"], r["n_corner_sel"])
        assert s["lm_iterations"] == r["lm_iterations"]
        assert s["successful_steps"] == r["successful_steps"]
        assert s["termination"] == r["termination"]
        assert s["is_degenerate"] == r["is_degenerate"]
        np.testing.assert_allclose(s["eigval"], r["eigval"], rtol=1e-8)
        assert abs(s["cost"] - r["initial_cost"]) <= 1e-9 * max(1.0, r["initial_cost"])
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_full_size_properties_and_idempotence(ctx, mla, synth):
    """BASELINE config 2 sizes (2 x 64 rings vs ~500k map): properties that do not need the oracle at full size --
    repeated solves are bit-identical (deterministic reduction), a solve started from the solution stays there,
    the pose lands on the ground truth, every matched count is stable across a map index rebuild."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = synth.make_scene(seed=42, **synth.SCENE_PRESETS["500k"])
        surf_map, corner_map = synth.sample_maps(sc)
        gt = synth.gt_body_pose()
        scans = [synth.simulate_scan(sc, gt, synth.HERCULES_BODY_T_LASER[i], 64, seed=7 + i) for i in range(2)]
    surf, corner = [], []
    for i, s in enumerate(scans):
        ex = ctx.extract(s.points, s.scan_start, s.scan_end)
        lab = ex["label"]
        # list / label consistency at full size
        assert np.all(lab[ex["sharp"]] == 2) and np.all(lab[ex["flat"]] == -1) and np.all(lab[ex["less_flat_raw"]] <= 0)
        assert len(ex["sharp"]) <= 2 * 6 * 64 and len(ex["less_sharp"]) <= 20 * 6 * 64 and len(ex["flat"]) <= 4 * 6 * 64
        T = np.eye(4)
        T[:3, :3] = synth.quat_to_rot(synth.HERCULES_BODY_T_LASER[i][:4])
        T[:3, 3] = synth.HERCULES_BODY_T_LASER[i][4:7]
        for lst, key in ((corner, "less_sharp"), (surf, "less_flat_raw")):
            a = np.zeros((len(ex[key]), 4), np.float32)
            a[:, :3] = synth.transform_points(s.points[ex[key]][:, :3], T)
            a[:, 3] = i
            lst.append(a)
    surf = synth.voxel_mean(np.concatenate(surf), 0.4)
    corner = synth.voxel_mean(np.concatenate(corner), 0.2)
    ctx.map_set(mla.SURF, surf_map)
    ctx.map_set(mla.CORNER, corner_map)
    ctx.features_set(mla.SURF, surf)
    ctx.features_set(mla.CORNER, corner)
    p0 = synth.perturbed_pose(gt, seed=43)
    pose_a, st_a = ctx.gn_solve(p0, 5)
    ctx.map_rebuild(mla.ALL_KINDS)
    pose_b, st_b = ctx.gn_solve(p0, 5)
    assert np.array_equal(pose_a, pose_b), "not deterministic / rebuild changed the result"
    assert [(s["n_surf"], s["n_corner"]) for s in st_a] == [(s["n_surf"], s["n_corner"]) for s in st_b]
    dt, dr = _pose_err(pose_a, gt)
    assert dt < 0.03 and dr < 2e-3, (dt, dr)
    pose_c, _ = ctx.gn_solve(pose_a, 3)       # fixed point
    dt, dr = _pose_err(pose_c, pose_a)
    assert dt < 2e-3 and dr < 2e-4
    # without per-iteration records the fused path takes the Cholesky fast path for evalDegenracy: same answer
    pose_d, _ = ctx.gn_solve(p0, 5, want_stats=False)
    dt, dr = _pose_err(pose_d, pose_a)
    assert dt < 1e-12 and dr < 1e-12


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_map_matches_unsharded_on_gpu(mla, synth, case16, feats16, world):
    """The N > 1 data path on one GPU: one context per map wedge (mlh_shard_set + that wedge's map subset), summed on the
    host. Sum over shards == the unsharded evaluation: same owners, same correspondences, same normal equations."""
    shard = importlib.import_module("m-loam_amd.shard")
    p0 = case16["p0"]
    full = mla.Context(0)
    ref = {}
    for kind, cloud, f in ((mla.SURF, case16["surf_map"], feats16[0]), (mla.CORNER, case16["corner_map"], feats16[1])):
        full.map_set(kind, cloud)
        full.features_set(kind, f)
        ref[kind] = full.match_linearize(kind, p0)
    full.close()
    acc = {k: dict(H=np.zeros((6, 6)), g=np.zeros(6), cost=0.0, count=0, valid=np.zeros(len(ref[k]["valid"]), np.int64)) for k in ref}
    for r in range(world):
        c = mla.Context(0)
        lo, hi = shard.wedge_planes(p0[:2], world, r)
        c.shard_set(lo, hi)
        for kind, cloud, f in ((mla.SURF, case16["surf_map"], feats16[0]), (mla.CORNER, case16["corner_map"], feats16[1])):
            keep = shard.shard_points_mask(cloud, p0[:2], world, r)
            if keep.sum() == 0:
                continue
            c.map_set(kind, np.ascontiguousarray(cloud[keep]))
            c.features_set(kind, f)
            out = c.match_linearize(kind, p0)
            a = acc[kind]
            a["H"] += out["H"]; a["g"] += out["g"]; a["cost"] += out["cost"]; a["count"] += out["count"]
            a["valid"] += out["valid"]
        c.close()
    for kind in ref:
        assert np.array_equal(acc[kind]["valid"], ref[kind]["valid"].astype(np.int64))
        assert acc[kind]["count"] == ref[kind]["count"]
        np.testing.assert_allclose(acc[kind]["H"], ref[kind]["H"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(acc[kind]["g"], ref[kind]["g"], rtol=1e-9, atol=1e-8)


def _sharded_gn(mla, shard, world, mode, surf_map, corner_map, surf, corner, p0, n_iters, eig_thre=100.0):
    """The multi-rank Gauss-Newton loop with the all-reduce done by the host: one context per rank (map wedge + halo and half-space ownership, or the
    whole map and round-robin feature ownership), every iteration = each rank's owned features matched and linearised at the common pose
    (mlh_match_linearize -> its packed J^T J / J^T r / cost / count), the records summed, the summed system solved ONCE and the identical update
    applied to the common pose -- exactly what comm.hip does with ncclAllReduce + the redundant per-rank solve. Returns pose, per-iteration counts."""
    ctxs = []
    for r in range(world):
        c = mla.Context(0)
        if mode == "map":
            lo, hi = shard.wedge_planes(p0[:2], world, r)
            c.shard_set(lo, hi)
            ms = np.ascontiguousarray(surf_map[shard.shard_points_mask(surf_map, p0[:2], world, r)])
            mc = np.ascontiguousarray(corner_map[shard.shard_points_mask(corner_map, p0[:2], world, r)])
            far = np.full((1, 3), 1.0e6, np.float32)
            c.map_set(mla.SURF, ms if len(ms) else far)
            c.map_set(mla.CORNER, mc if len(mc) else far)
        else:
            c.shard_set_features(world, r)
            c.map_set(mla.SURF, surf_map)
            c.map_set(mla.CORNER, corner_map)
        c.features_set(mla.SURF, surf)
        c.features_set(mla.CORNER, corner)
        ctxs.append(c)
    pose = np.array(p0, np.float64)
    counts, owned = [], np.zeros((world, 2), np.int64)
    for it in range(n_iters):
        H, g, ns, nc = np.zeros((6, 6)), np.zeros(6), 0, 0
        for r, c in enumerate(ctxs):
            a = c.match_linearize(mla.SURF, pose, dense=False)
            b = c.match_linearize(mla.CORNER, pose, dense=False)
            H += a["H"] + b["H"]; g += a["g"] + b["g"]; ns += a["count"]; nc += b["count"]
            owned[r] = (a["count"], b["count"])
        deg = mla.eval_degeneracy(H, eig_thre)
        d = np.linalg.solve(H, -g)
        pose = mla.pose_plus(pose, d, deg["V_update"] if deg["is_degenerate"] else None)
        counts.append((ns, nc))
    for c in ctxs:
        c.close()
    return pose, counts, owned


@pytest.mark.parametrize("mode", ["map", "features"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_solver_matches_unsharded(mla, synth, case16, feats16, world, mode):
    """the whole N > 1 solver on one GPU, ranks as contexts, all-reduce on the host: both partitions of SURVEY 8(e) (map wedges with halo; replicated
    map with round-robin feature ownership) reproduce the single-context device-resident solve -- same matched counts every iteration, same pose."""
    shard = importlib.import_module("m-loam_amd.shard")
    one = mla.Context(0)
    one.map_set(mla.SURF, case16["surf_map"]); one.map_set(mla.CORNER, case16["corner_map"])
    one.features_set(mla.SURF, feats16[0]); one.features_set(mla.CORNER, feats16[1])
    pose_ref, st = one.gn_solve(case16["p0"], 4)
    one.close()
    pose, counts, owned = _sharded_gn(mla, shard, world, mode, case16["surf_map"], case16["corner_map"], feats16[0], feats16[1], case16["p0"], 4)
    assert counts == [(s["n_surf"], s["n_corner"]) for s in st]
    dt, dr = _pose_err(pose, pose_ref)
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)
    if mode == "features":
        assert owned.sum(axis=1).min() > 0.6 * owned.sum() / world          # round-robin ownership is balanced by construction


def test_rccl_single_rank_path(mla, orc, case16, feats16):
    """The multi-GPU solver path (local reduce -> ncclAllReduce -> update kernel) with a 1-rank communicator gives the
    same iterates as the fused single-GPU path."""
    a = mla.Context(0)
    b = mla.Context(0)
    try:
        b.comm_init(1, 0, mla.comm_unique_id())
    except mla.MlhError as e:
        pytest.skip(f"RCCL not usable here: {e}")
    for c in (a, b):
        _stage(c, mla, case16, feats16)
    pa, sa = a.gn_solve(case16["p0"], 4)
    pb, sb = b.gn_solve(case16["p0"], 4)
    assert np.allclose(pa, pb, rtol=0, atol=1e-13)
    for x, y in zip(sa, sb):
        assert (x["n_surf"], x["n_corner"]) == (y["n_surf"], y["n_corner"])
    qa, _ = a.scan2map(case16["p0"])
    qb, _ = b.scan2map(case16["p0"])
    # the fused path runs the LM begin / step on a wavefront (rows on lanes), the communicator path runs the one-thread form of the same bodies in stand-alone
    # kernels: every element is computed by the same operations in the same order, so the poses are EQUAL, not close
    assert np.array_equal(qa, qb), np.abs(qa - qb).max()
    red = b.allreduce_f64(np.arange(29, dtype=np.float64))
    assert np.array_equal(red, np.arange(29, dtype=np.float64))
    red = b.allreduce_f64(np.arange(326, dtype=np.float64))          # the 24-dimensional window record (D (D + 1) / 2 + D + 2)
    assert np.array_equal(red, np.arange(326, dtype=np.float64))
    # pose-block mode (config 4) under the communicator: per-block records, one all-reduce, identical per-block updates
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    half_s, half_c = len(feats16[0]) // 2, len(feats16[1]) // 2
    poses0 = np.stack([case16["p0"], case16["p0"]])
    res = []
    for c in (a, b):
        c.features_set_blocks(mla.SURF, [feats16[0][:half_s], feats16[0][half_s:]])
        c.features_set_blocks(mla.CORNER, [feats16[1][:half_c], feats16[1][half_c:]])
        res.append(c.gn_solve_blocks(poses0, 3, [5, 10], [100.0, 100.0], [0, 1]))
    assert np.allclose(res[0][0], res[1][0], rtol=0, atol=1e-13)
    for it_a, it_b in zip(res[0][1], res[1][1]):
        for x, y in zip(it_a, it_b):
            assert (x["n_surf"], x["n_corner"], x["is_degenerate"]) == (y["n_surf"], y["n_corner"], y["is_degenerate"])
    a.close()
    b.close()


@pytest.mark.parametrize("method", ["rnd", "fps", "gd_fix", "gd_float"])
def test_good_feature_selection_parity(ctx, mla, orc, case16, feats16, method):
    """BASELINE config 5 building block: same seed -> the same selected features and information matrix as the oracle's
    restatement of ActiveFeatureSelection::goodFeatureMatching (lidar_mapper.h:229-573)."""
    surf = feats16[0]
    ctx.map_set(mla.SURF, case16["surf_map"])
    ctx.features_set(mla.SURF, surf)
    got = ctx.good_feature_matching(mla.SURF, case16["p0"], gf_method=method, gf_ratio=0.2, seed=11)
    ref = orc.good_feature_matching(orc.Map(case16["surf_map"]), "s", surf, case16["p0"],
                                    orc.mapper_params(gf_method=method, gf_ratio=0.2, seed=11))
    assert np.array_equal(got["sel"], ref["sel"])
    np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-9, atol=1e-9)
    # after the call only the selected correspondences are live on the device
    lin = ctx.linearize(mla.SURF, case16["p0"])
    assert lin["count"] == len(ref["sel"])


@pytest.mark.parametrize("variant", ["lattice", "duplicates", "five_points", "ratio_zero", "ratio_one", "nothing_matches", "host_loop", "dense_kernel",
                                     "lattice_dense_kernel", "nan_points", "inf_point", "one_cluster", "sixty_five_points", "all_the_same_point", "one_point",
                                     "sixteen_k_points_repeated", "one_more_than_the_device_takes"])
def test_fps_selection_on_the_device_ties_and_edge_sizes(mla, orc, case16, feats16, variant, monkeypatch):
    """'fps' runs its arg-max loop on the device (select.hip: fps_order_pruned_kernel -- slots of 64 Morton-ordered points that cannot change are skipped --, and
    fps_order_kernel, MLH_FPS_DENSE=1, which re-measures every point every round). Equal distances are decided by the lowest index, as the host loop's strict `>`
    over ascending indices decides them: features snapped to a 0.25 m lattice and features repeated verbatim produce such ties by the thousand. Edge sizes:
    five features, 65 (one slot and one point), num_use_features = 0, every feature wanted, nothing matched (the loop then visits every point and keeps none), every
    feature the same point (every distance 0); NaN / inf coordinates (the box test is switched off for such a cloud); MLH_FPS_HOST=1 runs the host loop through the
    same entry point. All against the oracle's restatement of the reference loop (lidar_mapper.h:352-408)."""
    surf = feats16[0].copy()
    ratio, pose = 0.2, case16["p0"]
    if variant.endswith("dense_kernel"):
        monkeypatch.setenv("MLH_FPS_DENSE", "1")
        variant = variant[:-len("dense_kernel")].rstrip("_") or "plain"
    if variant == "nan_points":
        surf[37::211, 0] = np.nan; surf[5::977, 2] = np.nan
    elif variant == "inf_point":
        surf[123, 1] = np.inf
    elif variant == "one_cluster":
        surf[:, :3] = surf[0, :3] + (surf[:, :3] - surf[0, :3]) * np.float32(1e-4)
    elif variant == "sixty_five_points":
        surf = np.ascontiguousarray(surf[100:165]); ratio = 0.9
    elif variant == "all_the_same_point":
        surf[:, :3] = surf[40, :3]
    elif variant == "one_point":
        surf = np.ascontiguousarray(surf[7:8]); ratio = 1.0
    elif variant in ("sixteen_k_points_repeated", "one_more_than_the_device_takes"):
        # the largest cloud the device loops take (16 384: every register slot of every wavefront in use) and the first one they leave to the host loop; the cloud
        # repeated verbatim: every distance several times over, every arg-max decided by the lowest index
        n = 16384 + (1 if variant.startswith("one_more") else 0)
        surf = np.ascontiguousarray(np.tile(surf, (n // len(surf) + 1, 1))[:n]); ratio = 0.05
    if variant == "lattice":
        surf[:, :3] = np.round(surf[:, :3] * 4.0) / 4.0
    elif variant == "duplicates":
        surf = np.ascontiguousarray(np.concatenate([surf[:700], surf[:700], surf[:300][::-1], surf[700:1500]]))
    elif variant == "five_points":
        surf = np.ascontiguousarray(surf[:5]); ratio = 0.6
    elif variant == "ratio_zero":
        ratio = 1e-5
    elif variant == "ratio_one":
        ratio = 1.0
    elif variant == "nothing_matches":
        surf[:, :3] += np.array([0.0, 0.0, 500.0], np.float32)
    elif variant == "host_loop":
        monkeypatch.setenv("MLH_FPS_HOST", "1")
    c = mla.Context(0)
    try:
        c.map_set(mla.SURF, case16["surf_map"])
        c.features_set(mla.SURF, surf)
        got = c.good_feature_matching(mla.SURF, pose, gf_method="fps", gf_ratio=ratio, seed=23)
        ref = orc.good_feature_matching(orc.Map(case16["surf_map"]), "s", surf, pose, orc.mapper_params(gf_method="fps", gf_ratio=ratio, seed=23))
        assert np.array_equal(got["sel"], ref["sel"]), (variant, len(got["sel"]), len(ref["sel"]))
        np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-12, atol=1e-12)
        if variant == "nothing_matches": assert len(ref["sel"]) == 0
        if variant == "ratio_one": assert len(ref["sel"]) > 0.5 * len(surf)
    finally:
        c.close()


def test_fps_asked_for_more_features_than_match(mla, orc, case16, feats16):
    """'fps' with gf_ratio * N above the number of matching features: the reference fills the list up with feature 1 when it matches (its loop has no "everything
    visited" exit, lidar_mapper.h:391-399) -- oracle pinned to the reference's lines in tests/test_oracle_ref_pin.py. The HIP path: the same list (feature 1 repeated),
    the same information matrix, and in scan2MapOptimization the repeated feature weighs as that many residual blocks (Corr::valid carries the count): residual
    block counts, LM iteration counts, terminations and pose equal to the oracle's."""
    p0 = case16["p0"]
    surf, corner = feats16[0][:600].copy(), feats16[1][:600].copy()
    surf[::2, :3] += 500.0
    corner[::2, :3] += 500.0
    c = mla.Context(0)
    try:
        c.map_set_pair(case16["surf_map"], case16["corner_map"])
        filled = 0
        for kind, ch, cloud, f in ((mla.SURF, "s", case16["surf_map"], surf), (mla.CORNER, "c", case16["corner_map"], corner)):
            for first_matches in (True, False):
                g = f.copy()
                if not first_matches:
                    g[1, :3] += 500.0
                c.features_set(kind, g)
                got = c.good_feature_matching(kind, p0, gf_method="fps", gf_ratio=0.8, seed=5)
                ref = orc.good_feature_matching(orc.Map(cloud), ch, g, p0, orc.mapper_params(gf_method="fps", gf_ratio=0.8, seed=5))
                assert np.array_equal(got["sel"], ref["sel"]), (ch, first_matches, len(got["sel"]), len(ref["sel"]))
                np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-9, atol=1e-9)
                lin = c.linearize(kind, p0)
                assert lin["count"] == len(ref["sel"])                    # the repeated feature counts as that many residual blocks
                filled += int((ref["sel"] == 1).sum() > 1)
        assert filled >= 1
        c.features_set(mla.SURF, surf); c.features_set(mla.CORNER, corner)
        opts = mla.default_opts(gf_method=mla.GF_METHODS["fps"], gf_ratio=0.8, gf_seed=5)
        pose, st = c.scan2map(p0, opts)
        ref = orc.scan2map(orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"]), surf, corner, p0, orc.mapper_params(gf_method="fps", gf_ratio=0.8, seed=5))
        for s_, r_ in zip(st, ref["outer"]):
            assert (s_["n_surf"], s_["n_corner"]) == (r_["n_surf_sel"], r_["n_corner_sel"])
            assert (s_["lm_iterations"], s_["termination"]) == (r_["lm_iterations"], r_["termination"])
        assert st[0]["n_surf"] + st[0]["n_corner"] > 600             # more residual blocks than features that can match: the repeats are in
        dt, dr = _pose_err(pose, ref["pose"])
        assert dt < 1e-7 and dr < 1e-7, (dt, dr)
    finally:
        c.close()


@pytest.mark.parametrize("method", ["gd_fix", "rnd"])
def test_selection_with_repeated_features_parity(mla, orc, case16, feats16, method):
    """Features repeated verbatim: subsets of the greedy loop whose members score exactly alike are replayed through the reference's heap, so the HIP path
    picks what the oracle picks (and the oracle what the reference's own lines pick: tests/test_oracle_ref_pin.py::test_selection_with_repeated_features_is_the_references)."""
    f = feats16[0][:1200]
    f = np.ascontiguousarray(np.concatenate([f[:400], f[:400], f[:400], f[400:800], f[400:800]]))
    c = mla.Context(0)
    try:
        c.map_set(mla.SURF, case16["surf_map"])
        c.features_set(mla.SURF, f)
        for seed in (3, 99):
            for ratio in (0.2, 0.5):
                got = c.good_feature_matching(mla.SURF, case16["p0"], gf_method=method, gf_ratio=ratio, seed=seed)
                ref = orc.good_feature_matching(orc.Map(case16["surf_map"]), "s", f, case16["p0"], orc.mapper_params(gf_method=method, gf_ratio=ratio, seed=seed))
                assert len(ref["sel"]) > 50
                assert np.array_equal(got["sel"], ref["sel"]), (method, seed, ratio)
                np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-9, atol=1e-9)
    finally:
        c.close()


def test_greedy_selection_lemma_scoring_equals_literal_scoring(ctx, mla, case16, feats16, monkeypatch):
    """The greedy loop ranks a subset's members by j H^-1 j^T (matrix determinant lemma) and replays near ties with the reference's
    logdet arithmetic; MLH_SELECT_EXACT=1 scores every member with the literal Cholesky logdet (lidar_mapper.h:497-520). Same picks,
    in the same order, for every seed and both kinds."""
    ctx.map_set(mla.SURF, case16["surf_map"])
    ctx.map_set(mla.CORNER, case16["corner_map"])
    for kind, f in ((mla.SURF, feats16[0]), (mla.CORNER, feats16[1])):
        ctx.features_set(kind, f)
        for seed in (1, 2, 3, 11, 12345):
            for ratio in (0.2, 0.8):
                monkeypatch.delenv("MLH_SELECT_EXACT", raising=False)
                a = ctx.good_feature_matching(kind, case16["p0"], gf_method="gd_fix", gf_ratio=ratio, seed=seed)
                monkeypatch.setenv("MLH_SELECT_EXACT", "1")
                b = ctx.good_feature_matching(kind, case16["p0"], gf_method="gd_fix", gf_ratio=ratio, seed=seed)
                monkeypatch.delenv("MLH_SELECT_EXACT", raising=False)
                assert len(a["sel"]) > 50 and np.array_equal(a["sel"], b["sel"]), (kind, seed, ratio)
                np.testing.assert_allclose(a["H"], b["H"], rtol=1e-12, atol=0)


def test_eval_full_hessian_logdet_and_ratio_policy(ctx, mla, orc, synth, case16, feats16):
    """row a12: ActiveFeatureSelection::evalFullHessian (lidar_mapper.h:176-227) -> common::logDet (math.hpp:173-187) -> gf_deg_factor ->
    the gf_ratio policy (lidar_mapper_keyframe.cpp:456-494). The C-ABI form of evalFullHessian is mlh_match_linearize with
    MLH_FLAG_NO_LOSS | MLH_FLAG_WITH_UA (reduced outputs only) on top of the caller's 1e-6 * I seed."""
    ext = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    covs = np.diag([0.0025] * 3 + [0.00030461] * 3)[None]
    meas = np.diag([0.0025] * 3)
    ctx.map_set(mla.SURF, case16["surf_map"])
    ctx.map_set(mla.CORNER, case16["corner_map"])
    fs = ctx.downsample_current_scan(mla.SURF, feats16[0], 0.4, ext, covs, meas, True, 0.6)      # records with cov_vec (with_ua)
    fc = ctx.downsample_current_scan(mla.CORNER, feats16[1], 0.2, ext, covs, meas, True, 0.6)
    H = np.eye(6) * 1e-6
    n_tot = 0
    Href, nref = None, 0
    for kind, ch, f, cloud in ((mla.SURF, "s", fs, case16["surf_map"]), (mla.CORNER, "c", fc, case16["corner_map"])):
        out = ctx.match_linearize(kind, case16["p0"], flags=mla.FLAG_WITH_UA | mla.FLAG_NO_LOSS, huber_delta=0.0, dense=False)
        H = H + out["H"]
        n_tot += out["count"]
        Href, nref = orc.eval_full_hessian(orc.Map(cloud), ch, f, case16["p0"], Href, nref)
    assert n_tot == nref and n_tot > 1000
    assert float(np.abs(H - Href).max()) <= 1e-9 * float(np.abs(Href).max())
    # NO_LOSS really is "not loss-corrected": with the Huber corrector the matrix is a different one
    out_l = ctx.match_linearize(mla.SURF, case16["p0"], flags=mla.FLAG_WITH_UA, huber_delta=0.1, dense=False)
    out_n = ctx.match_linearize(mla.SURF, case16["p0"], flags=mla.FLAG_WITH_UA | mla.FLAG_NO_LOSS, huber_delta=0.1, dense=False)
    assert float(np.abs(out_l["H"] - out_n["H"]).max()) > 1e-6 * float(np.abs(out_n["H"]).max())
    # gf_deg_factor = logDet(mat_H, use_cholesky = true); both matrices through the oracle's LLT restatement and through numpy
    ld_gpu, ld_ref = orc.logdet(H), orc.logdet(Href)
    assert abs(ld_gpu - ld_ref) < 1e-9 * abs(ld_ref)
    assert abs(ld_ref - np.linalg.slogdet(Href)[1]) < 1e-9 * abs(ld_ref)
    # the policy around the threshold (MAP_DEG_THRE), decided on the GPU-side factor and on the oracle's: same gf_ratio_cur
    for method in ("wo_gf", "rnd", "fps", "gd_fix", "gd_float"):
        for thre in (ld_ref - 1.0, ld_ref + 1.0):
            a = orc.gf_ratio_policy(method, 0.2, ld_gpu, thre, 0.55)
            b = orc.gf_ratio_policy(method, 0.2, ld_ref, thre, 0.55)
            assert a == b
            assert a == (1.0 if method == "wo_gf" else (0.2 if method != "gd_float" or ld_ref > thre else 0.8))


def test_scan2map_with_greedy_selection_parity(ctx, mla, orc, case16, feats16):
    _stage(ctx, mla, case16, feats16)
    opts = mla.default_opts(gf_method=mla.GF_METHODS["gd_fix"], gf_ratio=0.2, gf_seed=5)
    pose, stats = ctx.scan2map(case16["p0"], opts)
    ref = orc.scan2map(orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"]), feats16[0], feats16[1], case16["p0"],
                       orc.mapper_params(gf_method="gd_fix", gf_ratio=0.2, seed=5))
    for s, r in zip(stats, ref["outer"]):
        assert (s["n_surf"], s["n_corner"]) == (r["n_surf_sel"], r["n_corner_sel"])
        assert s["lm_iterations"] == r["lm_iterations"] and s["termination"] == r["termination"]
    dt, dr = _pose_err(pose, ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)


def test_per_ring_voxel_grid_parity(ctx, orc, case16):
    """row a3: the per-ring pcl::VoxelGrid(0.2) on the less-flat points. PCL sums a voxel's members in the order an unstable std::sort
    (comparator on the voxel index only) leaves them; with the default member order the HIP path sums along that same permutation
    (stdsort.hip, one std::sort per ring), so the centroids are the oracle's literal restatement's bit for bit. With
    mlh_set_voxel_member_order(ctx, 0) members are summed in position order: same voxels, centroids equal to f32 rounding."""
    for sc in case16["scans"]:
        ref = orc.extract(sc.points, sc.scan_start, sc.scan_end)["less_flat_ds"]
        got = ctx.extract(sc.points, sc.scan_start, sc.scan_end, voxel_leaf=0.2)["less_flat_ds"]
        assert got.shape == ref.shape and len(ref) > 1000
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
        ctx.set_voxel_member_order(0)
        try:
            fast = ctx.extract(sc.points, sc.scan_start, sc.scan_end, voxel_leaf=0.2)["less_flat_ds"]
        finally:
            ctx.set_voxel_member_order(1)
        assert fast.shape == ref.shape
        np.testing.assert_allclose(fast, ref, rtol=2e-6, atol=2e-6)
        assert 0.9 < np.mean(fast.view(np.uint32) == ref.view(np.uint32)) < 1.0     # the position-order sums do differ in some voxels: the case is not vacuous


def test_per_ring_voxel_grid_carries_intensity(ctx, orc, track_case):
    """The intensity field travels through extractCloud's VoxelGrid (averaged with the coordinates): with intensity = ring id, as
    ImageSegmenter sets it, every thinned less-flat point keeps its ring -- what the scan-to-scan tracker reads."""
    sc = track_case["scans"][0]
    got = ctx.extract(sc.points, sc.scan_start, sc.scan_end, voxel_leaf=0.2)["less_flat_ds"]
    ref = orc.extract(sc.points, sc.scan_start, sc.scan_end)["less_flat_ds"]
    assert got.shape == ref.shape and ref[:, 3].max() == sc.n_rings - 1
    np.testing.assert_array_equal(got[:, 3], ref[:, 3])
    assert np.all(np.diff(got[:, 3]) >= 0)


def test_point_uncertainty_parity(ctx, mla, orc, synth, feats16):
    rng = np.random.default_rng(3)
    pts = feats16[0][:4000].copy()
    pts[:, 3] = rng.integers(0, 4, len(pts))
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])   # -> [t, q]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    cov_ext = np.diag([0.0025, 0.0025, 0.0025, 0.00030461, 0.00030461, 0.00030461])
    covs = np.stack([np.zeros((6, 6)), cov_ext, cov_ext * 4, cov_ext * 9])
    meas = np.diag([0.0025] * 3)
    cov, keep = ctx.point_uncertainty(pts, ext, covs, meas, trace_threshold=0.6)
    for n in range(4):
        m = pts[:, 3] == n
        # the oracle evaluates evalPointUncertainty(point_sel, pose_ext) with point_sel = pose_ext^-1 * p (f32)
        q = ext[n][3:]
        R = synth.quat_to_rot(q)
        sel = ((pts[m, :3].astype(np.float64) - ext[n][:3]) @ R).astype(np.float32)
        ref = orc.eval_point_uncertainty(sel, ext[n], covs[n], meas)
        ref6 = np.stack([ref[:, 0, 0], ref[:, 0, 1], ref[:, 0, 2], ref[:, 1, 1], ref[:, 1, 2], ref[:, 2, 2]], axis=1)
        np.testing.assert_allclose(cov[m], ref6, rtol=2e-5, atol=1e-7)
        tr = ref[:, 0, 0] + ref[:, 1, 1] + ref[:, 2, 2]
        clear = np.abs(tr - 0.6) > 1e-4
        assert np.array_equal(keep[m][clear], (tr <= 0.6)[clear])
    assert 0 < keep.sum() < len(keep)


def test_pose_blocks_config4_parity(mla, orc, synth, case16):
    """BASELINE config 4 structure: one pose block per LiDAR (block 0 = body pose from the reference LiDAR with N_NEIGH 5,
    blocks 1..3 = the other LiDARs with N_NEIGH 10, CHECK_FOV on, frozen when degenerate), all blocks and both feature kinds
    in the same two launches per GN iteration. Each block must reproduce the oracle's single-block iteration on its own cloud."""
    import conftest
    case = conftest._make_case(synth, "50k", 16, 4)
    surf_b, corner_b, poses0 = [], [], []
    rng = np.random.default_rng(9)
    for i, sc in enumerate(case["scans"]):
        ex = orc.extract(sc.points, sc.scan_start, sc.scan_end)
        c = np.zeros((len(ex["less_sharp"]), 4), np.float32)
        c[:, :3] = sc.points[ex["less_sharp"]][:, :3]
        s = ex["less_flat_ds"].copy()
        surf_b.append(np.ascontiguousarray(synth.voxel_mean(s, 0.4)))
        corner_b.append(np.ascontiguousarray(synth.voxel_mean(c, 0.2)))
        # pose of LiDAR i in the map frame, perturbed
        T = synth.pose_to_mat(case["gt"]) @ np.block([[synth.quat_to_rot(synth.HERCULES_BODY_T_LASER[i][:4]), synth.HERCULES_BODY_T_LASER[i][4:7, None]], [np.zeros((1, 3)), np.ones((1, 1))]])
        # rotation matrix -> quaternion through the perturbation helper
        w = np.sqrt(max(0.0, 1 + T[0, 0] + T[1, 1] + T[2, 2])) / 2
        q = np.array([(T[2, 1] - T[1, 2]) / (4 * w), (T[0, 2] - T[2, 0]) / (4 * w), (T[1, 0] - T[0, 1]) / (4 * w), w])
        gt_i = np.concatenate([T[:3, 3], q / np.linalg.norm(q)])
        poses0.append(synth.perturbed_pose(gt_i, seed=50 + i, dt=0.1, drot_deg=1.0))
    poses0 = np.array(poses0)
    k_neigh, thre, freeze = [5, 10, 10, 10], [100.0, 70.0, 70.0, 70.0], [0, 1, 1, 1]
    ctx = mla.Context(0)
    ctx.map_set(mla.SURF, case["surf_map"])
    ctx.map_set(mla.CORNER, case["corner_map"])
    ctx.features_set_blocks(mla.SURF, surf_b)
    ctx.features_set_blocks(mla.CORNER, corner_b)
    opts = mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=1.0)
    n_it = 3
    poses, stats = ctx.gn_solve_blocks(poses0, n_it, k_neigh, thre, freeze, opts)
    ms, mc = orc.Map(case["surf_map"]), orc.Map(case["corner_map"])
    for b in range(4):
        prm = orc.mapper_params(huber_delta=1.0, map_eig_thre=thre[b], n_neigh=k_neigh[b], check_fov=True, freeze_when_degenerate=bool(freeze[b]))
        ref = orc.gn_iterations(ms, mc, surf_b[b], corner_b[b], poses0[b], prm, n_it)
        for it in range(n_it):
            s, r = stats[it][b], ref["iters"][it]
            assert (s["n_surf"], s["n_corner"]) == (r["n_surf"], r["n_corner"]), (b, it)
            np.testing.assert_allclose(s["H"], r["H"], rtol=1e-8, atol=1e-6)
            assert s["is_degenerate"] == r["is_degenerate"]
        dt, dr = _pose_err(poses[b], ref["pose"])
        assert dt < 1e-7 and dr < 1e-7, (b, dt, dr)
    # the statistics path above is the classic finish. Without statistics the finish of every iteration but the last runs in the next correspondence launch, per
    # block (a workgroup sums and solves only its own block's records), with the search bounded by the previous iteration's 5 / 10 neighbours: the same bits -- also
    # with 1 LiDAR's blocks mixed K = 5 / 10, for 2, 3 and 5 iterations
    for nit in (2, 3, 5):
        ctx.set_gn_schedule(1, 1, 1)
        fast = ctx.gn_solve_blocks(poses0, nit, k_neigh, thre, freeze, opts, want_stats=False)[0]
        ctx.set_gn_schedule(0, 0, 0)
        classic = ctx.gn_solve_blocks(poses0, nit, k_neigh, thre, freeze, opts, want_stats=False)[0]
        assert np.array_equal(fast, classic), nit
        if nit == n_it:
            assert np.array_equal(classic, poses)
    # the blocks are independent: any subset of them solved by itself (bench.py deals the blocks over the ranks that way: one block with N_NEIGH 10 alone, two of
    # four, ...) returns the bits those blocks have among all four, under both schedules
    for subset in ([0], [1], [3], [0, 2], [1, 3], [1, 2, 3]):
        ctx.features_set_blocks(mla.SURF, [surf_b[b] for b in subset])
        ctx.features_set_blocks(mla.CORNER, [corner_b[b] for b in subset])
        arg = (np.ascontiguousarray(poses0[subset]), n_it, [k_neigh[b] for b in subset], [thre[b] for b in subset], [freeze[b] for b in subset], opts)
        for sched in ((1, 1, 1), (0, 0, 0)):
            ctx.set_gn_schedule(*sched)
            got = ctx.gn_solve_blocks(*arg, want_stats=False)[0]
            assert np.array_equal(np.asarray(got), np.asarray(poses)[subset]), (subset, sched, np.abs(np.asarray(got) - np.asarray(poses)[subset]).max())
    # a LiDAR without corner features in this frame (an empty block of one kind): its block is solved on its surf features alone, the others are not disturbed
    empty = np.zeros((0, 4), np.float32)
    ctx.features_set_blocks(mla.SURF, [surf_b[0], surf_b[1]])
    ctx.features_set_blocks(mla.CORNER, [corner_b[0], empty])
    prm = orc.mapper_params(huber_delta=1.0, map_eig_thre=thre[1], n_neigh=k_neigh[1], check_fov=True, freeze_when_degenerate=bool(freeze[1]))
    ref = orc.gn_iterations(ms, mc, surf_b[1], empty, poses0[1], prm, n_it)
    res = []
    for sched in ((1, 1, 1), (0, 0, 0)):
        ctx.set_gn_schedule(*sched)
        res.append(np.asarray(ctx.gn_solve_blocks(np.ascontiguousarray(poses0[:2]), n_it, k_neigh[:2], thre[:2], freeze[:2], opts, want_stats=False)[0]))
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0][0], np.asarray(poses)[0])
    dt, dr = _pose_err(res[0][1], ref["pose"])
    assert dt < 1e-7 and dr < 1e-7, (dt, dr)
    ctx.close()


def test_voxel_filter_covariance_parity(ctx, orc, synth, case16):
    """(f1) VoxelGridCovarianceMLOAM: voxel membership, output order and count are integer work; the weighted sums are added in the order the reference's
    (unstable) std::sort leaves a voxel's members in, which the device reproduces -- every output field bit for bit (voxels of several hundred members at
    the 2 m leaf included)."""
    rng = np.random.default_rng(11)
    base = case16["surf_map"][:30000, :3]
    xyz = np.concatenate([base, base + rng.normal(0, 0.1, base.shape).astype(np.float32)])[rng.permutation(2 * len(base))]
    n = len(xyz)
    pts = np.zeros((n, 11), np.float32)
    pts[:, :3] = xyz
    pts[:, 3] = rng.integers(0, 4, n)
    sd = rng.uniform(0.01, 0.9, (n, 3)).astype(np.float32)
    pts[:, 4] = sd[:, 0] ** 2; pts[:, 7] = sd[:, 1] ** 2; pts[:, 9] = sd[:, 2] ** 2
    pts[:, 5] = 0.1 * sd[:, 0] * sd[:, 1]; pts[:, 6] = -0.05 * sd[:, 0] * sd[:, 2]; pts[:, 8] = 0.02 * sd[:, 1] * sd[:, 2]
    pts[:, 10] = pts[:, 4] + pts[:, 7] + pts[:, 9]
    for leaf, thr in ((0.4, 1.0), (0.8, 0.5), (2.0, 1.0)):
        got = ctx.voxel_filter(pts, leaf, thr)
        ref = orc.voxel_grid_cov(pts, leaf, thr)
        assert got.shape == ref.shape and len(ref) < n
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    # every member above the trace threshold: the voxel survives with mu = 0 / 1 (weight_total forced to 1), as the reference
    hot = pts[:2000].copy()
    hot[:, 4] = 5.0
    got = ctx.voxel_filter(hot, 0.4, 1.0)
    ref = orc.voxel_grid_cov(hot, 0.4, 1.0)
    np.testing.assert_array_equal(got, ref)
    assert np.all(got[:, :3] == 0)


def test_voxel_filter_points_on_voxel_faces_and_repeated_points(ctx, orc):
    """Both branches of the filter on input that sits on the decision boundaries -- coordinates that are exact multiples of the leaf size (negative side
    included), a finer lattice inside the voxels, points repeated verbatim (hundreds of equal keys inside std::sort's order), two LiDAR ids mixed inside
    voxels: every output bit against the oracle, which is held to the reference's own lines on the same construction
    (tests/test_oracle_ref_pin.py::test_voxel_grid_points_on_voxel_faces_and_repeated_points)."""
    rng = np.random.default_rng(78)
    for leaf in (0.4, 0.2, 0.25):
        n = 4000
        rec = np.zeros((n, 11), np.float32)
        rec[:, :3] = rng.uniform(-5.0, 5.0, (n, 3)).astype(np.float32)
        rec[:, 2] = (0.05 * rng.standard_normal(n)).astype(np.float32)
        rec[n // 2:, 0] = (2.5 + 0.05 * rng.standard_normal(n - n // 2)).astype(np.float32)
        rec[n // 2:, 2] = rng.uniform(0, 3, n - n // 2).astype(np.float32)
        k = np.round(rec[:, :3] / np.float32(leaf))
        rec[: n // 3, :3] = (k[: n // 3] * np.float32(leaf)).astype(np.float32)
        rec[n // 3: 2 * n // 3, :3] = (np.round(rec[n // 3: 2 * n // 3, :3] * (4.0 / leaf)) * np.float32(leaf / 4.0)).astype(np.float32)
        sd = rng.uniform(0.05, 0.6, (n, 3)).astype(np.float32)
        rec[:, 4] = sd[:, 0] ** 2; rec[:, 7] = sd[:, 1] ** 2; rec[:, 9] = sd[:, 2] ** 2
        rec[:, 5] = 0.1 * sd[:, 0] * sd[:, 1]; rec[:, 6] = -0.05 * sd[:, 0] * sd[:, 2]; rec[:, 8] = 0.02 * sd[:, 1] * sd[:, 2]
        rec[:, 10] = rec[:, 4] + rec[:, 7] + rec[:, 9]
        rec = np.ascontiguousarray(np.concatenate([rec, rec[:700], rec[300:900][::-1]]))
        rec[:, 3] = rng.integers(0, 2, len(rec)).astype(np.float32)
        got = ctx.voxel_filter(np.ascontiguousarray(rec[:, :4]), leaf)
        ref = orc.voxel_grid_mloam_plain(np.ascontiguousarray(rec[:, :4]), leaf, member_order=0)
        assert got.shape == ref.shape and len(ref) > 100
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
        for thr in (0.6, 2.0):
            got = ctx.voxel_filter(rec, leaf, thr)
            ref = orc.voxel_grid_cov(rec, leaf, thr)
            assert got.shape == ref.shape
            np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_voxel_filter_plain_parity(ctx, orc, case16):
    """PointXYZI branch (no covariance field; voxel_grid_covariance_mloam_impl.hpp:393-431): xyz mean over the members, intensity of the
    voxel's LAST member. The reference takes "last" in the order an unstable std::sort (comparator on the voxel index only) leaves the
    members in. The HIP path reproduces that order by default (bit for bit, sums included); with mlh_set_voxel_member_order(ctx, 0) it walks
    members in point-index order on the device and is pinned on THAT rule. Both on a cloud whose voxels mix intensities (a fused multi-LiDAR
    cloud carries the LiDAR id there). How often the two orders disagree is measured, not hidden: see
    tests/test_oracle_pipeline.py::test_plain_voxel_filter_member_order_dependence and DESIGN.md section 2."""
    rng = np.random.default_rng(5)
    xyz = case16["corner_map"][:30000, :3]
    try:
        for ids in (np.arange(len(xyz), dtype=np.float32), rng.integers(0, 2, len(xyz)).astype(np.float32)):   # unique tags; two LiDAR ids mixed
            pts = np.concatenate([xyz, ids[:, None]], axis=1).astype(np.float32)
            for leaf in (0.2, 0.4, 1.0):
                ctx.set_voxel_member_order(True)
                got = ctx.voxel_filter(pts, leaf)
                ref = orc.voxel_grid_mloam_plain(pts, leaf, member_order=0)
                np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))   # same order, same association: every bit
                ctx.set_voxel_member_order(False)
                got = ctx.voxel_filter(pts, leaf)
                ref = orc.voxel_grid_mloam_plain(pts, leaf, member_order=1)
                assert got.shape == ref.shape
                np.testing.assert_array_equal(got[:, 3], ref[:, 3])                       # which member survives: a selection, exact
                np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=2e-6, atol=2e-6)  # f32 sums in a different association
                avg = orc.voxel_grid(pts, leaf)                                           # pcl::VoxelGrid<PointXYZI>: same voxels, same centroid rule
                assert avg.shape == ref.shape
        one = ctx.voxel_filter(pts[:1], 0.4)
        np.testing.assert_array_equal(one, pts[:1])
    finally:
        ctx.set_voxel_member_order(True)


def test_downsample_current_scan_mixed_lidar_voxels(ctx, mla, orc, synth, feats16):
    """downsampleCurrentScan on a fused cloud whose voxels hold points of BOTH LiDARs (lidar_mapper_keyframe.cpp:356-398): the thinned
    point's LiDAR id -- the voxel's last member's -- selects the extrinsic the uncertainty is propagated through, so a wrong member would
    show in the covariance and in the trace gate. This is the device-only leg (mlh_set_voxel_member_order(ctx, 0): point-index order);
    the default order is held against the reference in test_voxel_filters_in_the_references_member_order and
    tests/test_gpu_parity_fullsize.py::test_downsample_current_scan_against_the_references_own_lines."""
    rng = np.random.default_rng(21)
    base = feats16[0][:, :3]
    xyz = np.concatenate([base, base + rng.normal(0, 0.08, base.shape).astype(np.float32)])
    pts = np.zeros((len(xyz), 4), np.float32)
    pts[:, :3] = xyz
    pts[:, 3] = rng.integers(0, 2, len(xyz))                     # ids mixed inside voxels
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.diag([0.0004] * 3 + [0.0001] * 3), np.diag([0.0025] * 3 + [0.00030461] * 3) * 30])
    meas = np.diag([0.0025] * 3)
    thr = 0.05
    ds = orc.voxel_grid_mloam_plain(pts, 0.4, member_order=1)
    keep_ref, cov_ref, traces = [], [], []
    for p in ds:
        n = int(p[3])
        R = synth.quat_to_rot(ext[n][3:])
        sel = ((p[:3].astype(np.float64) - ext[n][:3]) @ R).astype(np.float32)
        c = orc.eval_point_uncertainty(sel[None, :], ext[n], covs[n], meas)[0]
        traces.append(np.trace(c))
        keep_ref.append(np.trace(c) <= thr)
        cov_ref.append([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])
    keep_ref, cov_ref, traces = np.array(keep_ref), np.array(cov_ref), np.array(traces)
    assert np.all(np.abs(traces - thr) > 1e-4 * thr)             # no voxel sits on the gate, where a 2e-6 centroid difference could flip it
    ctx.set_voxel_member_order(False)
    try:
        got = ctx.downsample_current_scan(mla.SURF, pts, 0.4, ext, covs, meas, True, thr)
        dsg = ctx.voxel_filter(pts, 0.4)
    finally:
        ctx.set_voxel_member_order(True)
    np.testing.assert_array_equal(dsg[:, 3], ds[:, 3])
    assert 0 < keep_ref.sum() < len(ds) and len(set(ds[:, 3])) == 2
    assert len(got) == keep_ref.sum()
    np.testing.assert_array_equal(got[:, 3], ds[keep_ref][:, 3])
    np.testing.assert_allclose(got[:, :3], ds[keep_ref][:, :3], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(got[:, 4:10], cov_ref[keep_ref], rtol=5e-5, atol=1e-9)


def test_voxel_filters_in_the_references_member_order(mla, orc, synth, case16, feats16):
    """The default member order: a voxel's members in the order libstdc++'s std::sort leaves them (the reference's own order,
    voxel_grid_covariance_mloam_impl.hpp:227), reproduced by running that sort on the host over the same sequence. With the order equal,
    everything is: surviving ids, and the f32 sums BIT FOR BIT (same association) -- plain branch, covariance branch, and
    downsampleCurrentScan on a cloud whose voxels mix LiDAR ids."""
    c = mla.Context(0)                                             # a fresh context: the default is what is tested
    try:
        rng = np.random.default_rng(5)
        xyz = case16["corner_map"][:30000, :3]
        pts = np.concatenate([xyz, rng.integers(0, 2, len(xyz)).astype(np.float32)[:, None]], axis=1).astype(np.float32)
        n_id_diff = 0
        for leaf in (0.2, 0.4, 1.0):
            got = c.voxel_filter(pts, leaf)
            ref = orc.voxel_grid_mloam_plain(pts, leaf, member_order=0)
            np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
            n_id_diff += int(np.sum(ref[:, 3] != orc.voxel_grid_mloam_plain(pts, leaf, member_order=1)[:, 3]))
        assert n_id_diff > 0                                       # the case is not vacuous: point-index order gives other ids here
        # covariance branch
        n = 40000
        base = case16["surf_map"][:n // 2, :3]
        x2 = np.concatenate([base, base + rng.normal(0, 0.1, base.shape).astype(np.float32)])[rng.permutation(n)]
        p11 = np.zeros((n, 11), np.float32)
        p11[:, :3] = x2
        p11[:, 3] = rng.integers(0, 4, n)
        sd = rng.uniform(0.01, 0.9, (n, 3)).astype(np.float32)
        p11[:, 4] = sd[:, 0] ** 2; p11[:, 7] = sd[:, 1] ** 2; p11[:, 9] = sd[:, 2] ** 2
        p11[:, 10] = p11[:, 4] + p11[:, 7] + p11[:, 9]
        for leaf, thr in ((0.4, 1.0), (2.0, 1.0)):
            got = c.voxel_filter(p11, leaf, thr)
            ref = orc.voxel_grid_cov(p11, leaf, thr)
            assert got.shape == ref.shape
            np.testing.assert_array_equal(got[:, :4].view(np.uint32), ref[:, :4].view(np.uint32))     # mu, intensity: same association
            np.testing.assert_allclose(got[:, 4:], ref[:, 4:], rtol=1e-6, atol=1e-12)
        # downsampleCurrentScan, ids mixed inside voxels: the feature set the reference's order yields
        b3 = feats16[0][:, :3]
        xyz = np.concatenate([b3, b3 + rng.normal(0, 0.08, b3.shape).astype(np.float32)])
        f4 = np.zeros((len(xyz), 4), np.float32)
        f4[:, :3] = xyz
        f4[:, 3] = rng.integers(0, 2, len(xyz))
        ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
        for e in ext:
            e[3:] /= np.linalg.norm(e[3:])
        covs = np.stack([np.diag([0.0004] * 3 + [0.0001] * 3), np.diag([0.0025] * 3 + [0.00030461] * 3) * 30])
        meas = np.diag([0.0025] * 3)
        thr = 0.05
        ds = orc.voxel_grid_mloam_plain(f4, 0.4, member_order=0)
        keep = []
        for p in ds:
            k = int(p[3])
            R = synth.quat_to_rot(ext[k][3:])
            sel = ((p[:3].astype(np.float64) - ext[k][:3]) @ R).astype(np.float32)
            keep.append(np.trace(orc.eval_point_uncertainty(sel[None, :], ext[k], covs[k], meas)[0]) <= thr)
        keep = np.array(keep)
        got = c.downsample_current_scan(mla.SURF, f4, 0.4, ext, covs, meas, True, thr)
        assert 0 < keep.sum() < len(ds) and len(got) == keep.sum()
        np.testing.assert_array_equal(got[:, :4].view(np.uint32), ds[keep].view(np.uint32))
        # the same through the host pass (the platform's own std::sort)
        c.set_voxel_member_order("host")
        np.testing.assert_array_equal(c.voxel_filter(pts, 0.4).view(np.uint32), orc.voxel_grid_mloam_plain(pts, 0.4, member_order=0).view(np.uint32))
        # the point-index order is per context and switchable
        c.set_voxel_member_order(False)
        again = c.voxel_filter(pts, 0.4)
        np.testing.assert_array_equal(again[:, 3], orc.voxel_grid_mloam_plain(pts, 0.4, member_order=1)[:, 3])
    finally:
        c.close()


def test_device_std_sort_equals_std_sort(mla, orc):
    """stdsort.hip: libstdc++'s std::sort (key-only comparator) restated data-parallel on the device -- the permutation must equal the
    library's own, element for element: random keys with many duplicates (the voxel-slot case), all-equal keys, already sorted / reversed /
    organ-pipe / sawtooth inputs (several of which exhaust introsort's depth budget and take the heap-sort path), sizes around the
    16-element insertion threshold, and the two-clouds-in-one-call form. Also against the host pass (mode 2: std_sort_mt.hpp)."""
    rng = np.random.default_rng(11)
    c = mla.Context(0)
    try:
        cases = []
        # (8 191 .. 12 289: around the threshold above which a range is partitioned by SEVERAL workgroups and around its 4 096-element workgroup chunks -- round 5)
        for n in (1, 2, 15, 16, 17, 18, 33, 100, 257, 1025, 5000, 8191, 8192, 8193, 12288, 12289, 40000, 62365):
            for nv in sorted({1, 2, 7, n // 4 + 1, n + 1}):
                i = np.arange(n)
                cases += [rng.integers(0, nv, n), i % nv, (n - i) // (n // nv + 1), np.where(i < n // 2, i, n - i) % nv, (n - i) % nv]
        cases.append(rng.integers(0, 50000, 200000))
        cases.append(rng.integers(0, 3, 262144))           # the longest range the wide levels take, almost all pairs crossing
        cases.append(rng.integers(0, 100000, 262145))      # one element more: that range stays with one workgroup
        big = np.arange(100000) % 7 + 1                     # (keys are voxel indices: non-negative, compared as PCL's unsigned idx)
        big[50000] = 0                                      # the first level's median candidates are positions 1, mid, last - 1: the median sits at position 1
        cases.append(big)
        for keys in cases:
            keys = keys.astype(np.int32)
            want = orc.std_sort_permutation(keys)
            got = c.std_sort_permutation(keys, mode=1)
            np.testing.assert_array_equal(got, want)
            if len(keys) in (5000, 62365):
                np.testing.assert_array_equal(c.std_sort_permutation(keys, mode=2), want)
        # two clouds in one call = two std::sort calls
        keys = rng.integers(0, 9000, 62365 + 15161).astype(np.int32)
        keys[62365:] += 9000
        want = np.concatenate([orc.std_sort_permutation(keys[:62365]), 62365 + orc.std_sort_permutation(keys[62365:])])
        for mode in (1, 2):
            np.testing.assert_array_equal(c.std_sort_permutation(keys, n0=62365, mode=mode), want)
    finally:
        c.close()


@pytest.mark.parametrize("lanes", [8, 16, 32])
def test_correspondence_lane_widths(mla, orc, case16, feats16, lanes, monkeypatch):
    """The correspondence kernel runs with 8 lanes per query on chip-filling launches and 16 on small ones (32 -- two DPP rows per query -- was built and measured in
    round 5 and is not selected by any rule: profiles/r05_knockout_experiments.txt); every width must give the oracle's matches bit for bit (MLH_KNN_LANES pins the
    width of a context)."""
    monkeypatch.setenv("MLH_KNN_LANES", str(lanes))
    c = mla.Context(0)
    try:
        for kind, ch, cloud, feats in ((mla.SURF, "s", case16["surf_map"], feats16[0]), (mla.CORNER, "c", case16["corner_map"], feats16[1])):
            c.map_set(kind, cloud)
            c.features_set(kind, feats)
            got = c.match_linearize(kind, case16["p0"])
            valid, coeffs = orc.Map(cloud).match(ch, feats, case16["p0"])
            assert np.array_equal(got["valid"], valid)
            assert np.array_equal(got["coeffs"].astype(np.float32).view(np.uint32), coeffs.astype(np.float32).view(np.uint32))
    finally:
        c.close()


def _keyframe_cloud(rng, xyz, n_lidar):
    pts = np.zeros((len(xyz), 11), np.float32)
    pts[:, :3] = xyz
    pts[:, 3] = rng.integers(0, n_lidar, len(xyz))
    pts[:, 4:10] = rng.uniform(0, 1, (len(xyz), 6))       # stale covariance fields: must be overwritten
    pts[:, 10] = 7.0
    return pts


def _uct_setup(synth, n_lidar=2):
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:n_lidar]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    cov_ext = np.stack([np.zeros((6, 6))] + [np.diag([0.0025] * 3 + [0.00030461] * 3) * (k + 1) for k in range(n_lidar - 1)])
    rng = np.random.default_rng(21)
    A = rng.normal(size=(6, 6))
    cov_global = A @ A.T * 2e-5
    pose_global = np.array([12.0, -7.5, 0.4, 0.01, -0.02, np.sin(0.4), np.cos(0.4)])
    pose_global[3:] /= np.linalg.norm(pose_global[3:])
    return ext, cov_ext, pose_global, cov_global, np.diag([0.0025] * 3)


def test_cloud_uct_associate_to_map_parity(ctx, orc, synth, feats16):
    """(f1) cloudUCTAssociateToMap: kept set and order exact, map-frame coordinates bit-exact (f64 math, f32 store), covariance
    to f32 rounding of an f64 product whose summation order differs (rtol 2e-5)."""
    rng = np.random.default_rng(4)
    ext, cov_ext, pose_global, cov_global, meas = _uct_setup(synth)
    pts = _keyframe_cloud(rng, feats16[0][:6000, :3] * np.float32(1.5), 2)
    compound_tr = [np.trace(orc.compound_pose_with_cov(pose_global, cov_global, ext[l], cov_ext[l])[1]) for l in range(2)]
    assert all(np.isfinite(compound_tr))
    ref_all = orc.cloud_uct_associate_to_map(pts, pose_global, cov_global, ext, cov_ext, meas, True, 1e9)
    thr = float(np.median(ref_all[:, 10]))                   # drops about half of the cloud
    for with_ua, t in ((True, thr), (True, 1e9), (False, thr)):
        got = ctx.cloud_uct_associate_to_map(pts, pose_global, cov_global, ext, cov_ext, meas, with_ua, t)
        ref = orc.cloud_uct_associate_to_map(pts, pose_global, cov_global, ext, cov_ext, meas, with_ua, t)
        clear = np.abs(ref_all[:, 10] - t) > 1e-6 * t
        if clear.all() or not with_ua:
            assert got.shape == ref.shape
        else:                                                # a trace within rounding of the threshold may fall either way
            assert abs(len(got) - len(ref)) <= int((~clear).sum())
            continue
        assert np.array_equal(got[:, :4].view(np.uint32), ref[:, :4].view(np.uint32))
        np.testing.assert_allclose(got[:, 4:], ref[:, 4:], rtol=2e-5, atol=1e-9)
        if not with_ua:
            assert len(got) == len(pts) and not got[:, 4:].any()
    assert 0 < len(ctx.cloud_uct_associate_to_map(pts, pose_global, cov_global, ext, cov_ext, meas, True, thr)) < len(pts)


def test_local_map_assembled_on_device(mla, orc, synth, case16, feats16):
    """The map can be born in HBM: keyframe clouds -> cloudUCTAssociateToMap -> VoxelGridCovarianceMLOAM -> map index, device
    buffers throughout (MLH_MEM_DEVICE), must index exactly the cloud the host-buffer path produces."""
    import torch
    rng = np.random.default_rng(8)
    ext, cov_ext, pose_global, cov_global, meas = _uct_setup(synth)
    base = case16["surf_map"][:40000, :3]
    kfs = [_keyframe_cloud(rng, base[i::3] + rng.normal(0, 0.05, base[i::3].shape).astype(np.float32), 2) for i in range(3)]
    poses = [pose_global, pose_global + np.array([0.5, 0.2, 0, 0, 0, 0, 0]), pose_global + np.array([-0.4, 0.3, 0.05, 0, 0, 0, 0])]
    c = mla.Context(0)
    try:
        # host-buffer path
        host = np.concatenate([c.cloud_uct_associate_to_map(k, p, cov_global, ext, cov_ext, meas, True, 0.2) for k, p in zip(kfs, poses)])
        host_ds = c.voxel_filter(host, 0.4, 0.2)
        # device path: one accumulation buffer, records appended in place
        total = sum(len(k) for k in kfs)
        d_acc = torch.zeros((total, 11), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        fill = 0
        for k, p in zip(kfs, poses):
            d_in = torch.from_numpy(k).cuda()
            torch.cuda.synchronize()
            fill += c.cloud_uct_associate_to_map_device(d_in, d_acc[fill:], p, cov_global, ext, cov_ext, meas, True, 0.2)
        assert fill == len(host)
        d_ds = torch.zeros((fill, 11), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        n_ds = c.voxel_filter_device(d_acc[:fill], d_ds, 0.4, 0.2)
        assert n_ds == len(host_ds)
        np.testing.assert_array_equal(d_ds[:n_ds].cpu().numpy(), host_ds)
        c.map_set(mla.SURF, d_ds[:n_ds])
        qm = host_ds[rng.integers(0, len(host_ds), 500), :3] + rng.normal(0, 0.2, (500, 3)).astype(np.float32)
        idx, d2 = c.knn(mla.SURF, qm.astype(np.float32))
        ridx, rd2 = orc.Map(np.ascontiguousarray(host_ds[:, :3])).knn(qm.astype(np.float32))
        within = rd2 < 1.0
        assert within.sum() > 100
        assert np.array_equal(idx[within], ridx[within]) and np.array_equal(d2[within].view(np.uint32), rd2[within].view(np.uint32))
    finally:
        c.close()


def test_pure_odom_batch_parity(ctx, orc):
    """(a19) LidarPureOdom{PlaneNorm,Edge}Factor::Evaluate for a whole window in one launch: residual and the three 1x7 Jacobians of
    every factor against the oracle's per-factor restatement (f64, rtol 1e-12 -- same expressions, different association)."""
    rng = np.random.default_rng(31)

    def rand_pose(scale):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        return np.concatenate([rng.uniform(-scale, scale, 3), q])

    n_frames, n_ext, n = 4, 3, 5000
    pivot = rand_pose(20.0)
    frames = np.stack([rand_pose(20.0) for _ in range(n_frames)])
    exts = np.stack([rand_pose(1.0) for _ in range(n_ext)])
    types = rng.integers(0, 2, n).astype(np.int32)
    points = rng.uniform(-40, 40, (n, 3))
    coeffs = np.zeros((n, 6))
    for i in range(n):
        if types[i] == 0:
            v = rng.normal(size=3); v /= np.linalg.norm(v)
            coeffs[i, :3] = v; coeffs[i, 3] = rng.uniform(-5, 5)
        else:
            c = rng.uniform(-40, 40, 3); v = rng.normal(size=3); v /= np.linalg.norm(v)
            coeffs[i, :3] = c + 0.1 * v; coeffs[i, 3:] = c - 0.1 * v
    fi = rng.integers(0, n_frames, n).astype(np.int32)
    ei = rng.integers(0, n_ext, n).astype(np.int32)
    sq = rng.uniform(0.3, 1.0, n)
    ctx.pure_odom_set(types, points, coeffs, fi, ei, sq)
    r, J = ctx.pure_odom_evaluate(pivot, frames, exts)
    for i in range(0, n, 7):
        kind = "s" if types[i] == 0 else "c"
        rr, JJ = orc.pure_odom_eval(kind, points[i], coeffs[i, :4] if kind == "s" else coeffs[i], pivot, frames[fi[i]], exts[ei[i]], sq[i])
        assert abs(r[i] - rr) <= 1e-12 * max(1.0, abs(rr))
        np.testing.assert_allclose(J[i], JJ, rtol=1e-11, atol=1e-11)
    assert np.all(J[:, :, 6] == 0)
    r2, J2 = ctx.pure_odom_evaluate(pivot, frames, exts, want_jacobians=False)
    assert J2 is None and np.array_equal(r, r2)
    # default weight 1.0 and the error paths
    ctx.pure_odom_set(types[:10], points[:10], coeffs[:10], fi[:10], ei[:10])
    r3, _ = ctx.pure_odom_evaluate(pivot, frames, exts)
    np.testing.assert_allclose(r3, r[:10] / sq[:10], rtol=1e-13)
    with pytest.raises(Exception):
        ctx.pure_odom_evaluate(pivot, frames[: fi[:10].max()], exts)      # pose array shorter than the largest frame index


def _window_case(synth, orc, n_frames=2, n_lidars=2, seed=3):
    import conftest
    return conftest.make_window_case(synth, orc, n_frames, n_lidars, seed)


@pytest.mark.parametrize("shape", [(1, 2), (3, 4)])
def test_pure_odom_window_normal_equations(ctx, mla, orc, synth, shape):
    """(a19, BASELINE config 4 proper) the COUPLED window problem of Estimator::optimizeMap (estimator.cpp:687-848): J^T J / J^T r / cost over the
    local parameters [pivot | frames | extrinsics] reduced on the device (24-dimensional for 1 frame + 2 LiDARs, the hercules shape) against
    the oracle's factor-by-factor accumulation; the diagonal blocks feed evalDegenracy (estimator.cpp:1598-1680) identically, and the
    Gauss-Newton step of the coupled system (pivot and reference extrinsic held constant, as the reference does) is the same on both sides."""
    n_frames, n_lidars = shape
    w = _window_case(synth, orc, n_frames, n_lidars)
    assert len(w["types"]) > 3000 * n_frames * n_lidars // 2
    ctx.pure_odom_set(w["types"], w["points"], w["coeffs"], w["fi"], w["ei"])
    got = ctx.pure_odom_normal_eq(w["pivot"], w["frames"], w["exts"], huber_delta=1.0)
    ref = orc.pure_odom_normal_eq(w["types"], w["points"], w["coeffs"], None, w["fi"], w["ei"], w["pivot"], w["frames"], w["exts"], 1.0)
    D = 6 * (1 + n_frames + n_lidars)
    assert got["H"].shape == (D, D) and got["count"] == ref["count"] == len(w["types"])
    sc = float(np.abs(ref["H"]).max())
    assert float(np.abs(got["H"] - ref["H"]).max()) <= 1e-9 * sc
    assert float(np.abs(got["g"] - ref["g"]).max()) <= 1e-9 * float(np.abs(ref["g"]).max())
    assert abs(got["cost"] - ref["cost"]) <= 1e-9 * ref["cost"]
    assert np.array_equal(got["H"], got["H"].T)
    again = ctx.pure_odom_normal_eq(w["pivot"], w["frames"], w["exts"], huber_delta=1.0)
    assert np.array_equal(again["H"], got["H"]) and np.array_equal(again["g"], got["g"])                  # deterministic reduction
    # blocks a factor set never touches together stay exactly zero (frame i x frame j, extrinsic m x extrinsic n)
    if n_frames > 1:
        assert not got["H"][6:12, 12:18].any()
    # evalDegenracy on every diagonal block
    for b in range(1 + n_frames + n_lidars):
        blk = slice(6 * b, 6 * b + 6)
        dg, dr_ = mla.eval_degeneracy(got["H"][blk, blk], 100.0), orc.eval_degeneracy(ref["H"][blk, blk], 100.0)
        assert dg["is_degenerate"] == dr_["is_degenerate"]
        np.testing.assert_allclose(dg["eigval"], dr_["eigval"], rtol=1e-8, atol=1e-8 * sc)
    # the coupled Gauss-Newton step with para_pose_[0] and para_ex_pose_[IDX_REF] constant (estimator.cpp:636, 642)
    free = np.r_[6:6 * (1 + n_frames), 6 * (2 + n_frames):D]
    step_g = np.linalg.solve(got["H"][np.ix_(free, free)], -got["g"][free])
    step_r = np.linalg.solve(ref["H"][np.ix_(free, free)], -ref["g"][free])
    np.testing.assert_allclose(step_g, step_r, rtol=1e-6, atol=1e-9)
    assert np.linalg.norm(step_r[:3]) > 1e-3                      # the window really is off its optimum: the step is not noise
    # the coupled problem SOLVED on the device (mlh_pure_odom_gn_solve): 4 Gauss-Newton iterations, pivot and reference extrinsic constant, against the
    # same iterations driven by the oracle's normal equations and a host solve
    def host_gn(n_it):
        fr, ex = w["frames"].copy(), w["exts"].copy()
        for _ in range(n_it):
            ne = orc.pure_odom_normal_eq(w["types"], w["points"], w["coeffs"], None, w["fi"], w["ei"], w["pivot"], fr, ex, 1.0)
            step = np.zeros(D); step[free] = np.linalg.solve(ne["H"][np.ix_(free, free)], -ne["g"][free])
            for i in range(n_frames):
                fr[i] = mla.pose_plus(fr[i], step[6 * (1 + i):6 * (2 + i)])
            for k in range(1, n_lidars):
                ex[k] = mla.pose_plus(ex[k], step[6 * (1 + n_frames + k):6 * (2 + n_frames + k)])
        return fr, ex, ne
    sol = ctx.pure_odom_gn_solve(w["pivot"], w["frames"], w["exts"], n_iters=4, huber_delta=1.0)
    fr_r, ex_r, ne_last = host_gn(4)
    assert sol["status"] == 0 and sol["count"] == len(w["types"])
    assert np.abs(sol["frames"] - fr_r).max() < 1e-9 and np.abs(sol["exts"] - ex_r).max() < 1e-9
    assert np.array_equal(sol["exts"][0], w["exts"][0])                                # the reference extrinsic is not touched
    assert np.abs(sol["frames"] - w["frames"]).max() > 1e-3                            # ... and the window really moved
    assert abs(sol["cost"] - ne_last["cost"]) <= 1e-9 * ne_last["cost"]
    # a frozen block (V_update = 0, what evalDegenracy leaves in an extrinsic that must not be updated) stays put, the others still move
    Vz = np.tile(np.eye(6).ravel(), (1 + n_frames + n_lidars, 1))
    Vz[-1] = 0.0
    frozen = ctx.pure_odom_gn_solve(w["pivot"], w["frames"], w["exts"], n_iters=2, huber_delta=1.0, V_update=Vz)
    assert np.abs(frozen["exts"][-1] - w["exts"][-1]).max() < 1e-15 and np.abs(frozen["frames"] - w["frames"]).max() > 1e-3      # (Plus re-normalises the quaternion: last bit)
    # no loss (huber_delta <= 0): plain J^T J
    nl_g = ctx.pure_odom_normal_eq(w["pivot"], w["frames"], w["exts"], huber_delta=0.0)
    nl_r = orc.pure_odom_normal_eq(w["types"], w["points"], w["coeffs"], None, w["fi"], w["ei"], w["pivot"], w["frames"], w["exts"], 1e9)
    assert float(np.abs(nl_g["H"] - nl_r["H"]).max()) <= 1e-9 * float(np.abs(nl_r["H"]).max())


def test_scan2map_split_submission_equals_the_synchronous_call(mla, case16, feats16):
    """mlh_scan2map_begin / _end (+ _begin_chained): the whole solve enqueued without the host reading the LM loop's verdict. Inside the look-ahead the pose is
    mlh_scan2map's bit for bit; a look-ahead that is too short is detected on the device -- never returned as a result -- and the frame is solved again inside
    _end (status 2) or handed back to the caller (status 1) when a younger solve is chained behind it."""
    p0 = case16["p0"]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    c = mla.Context(0)
    try:
        _stage(c, mla, case16, feats16)
        ref, st = c.scan2map(p0)
        need = max(int(x["lm_iterations"]) for x in st)
        assert 2 <= need <= 10
        ref2 = c.scan2map(p0, want_stats=False)[0]
        assert np.array_equal(ref, ref2)
        # default look-ahead (10): converged inside it
        c.scan2map_begin(p0)
        pose, status = c.scan2map_end()
        assert status == 0 and np.array_equal(pose, ref)
        # exactly enough launches, and one too few
        c.scan2map_begin(p0, lm_lookahead=need)
        pose, status = c.scan2map_end()
        assert status == 0 and np.array_equal(pose, ref)
        c.scan2map_begin(p0, lm_lookahead=need - 1)
        pose, status = c.scan2map_end()
        assert status == 2 and np.array_equal(pose, ref)        # too short: found out on the device, solved again inside the call
        # the synchronous call is unaffected by what the split calls left behind (the overflow flag is per solve)
        assert np.array_equal(c.scan2map(p0, want_stats=False)[0], ref)
        # two frames in flight, the second chained on the device from the first one's result (identity odometry: it starts where frame 0 ended)
        c.scan2map_begin(p0)
        c.scan2map_begin_chained(ident, ident)
        a, sa = c.scan2map_end()
        b, sb = c.scan2map_end()
        assert sa == 0 and sb == 0 and np.array_equal(a, ref)
        b_ref = c.scan2map(a, want_stats=False)[0]
        assert float(np.abs(b - b_ref).max()) < 1e-12           # Pose(q, t) products normalise the quaternion: equal to a restart from `a` to rounding
        # a frame that overflows WITH a younger frame chained behind it is handed back (status 1, pose = its start pose), not re-solved on changed state
        c.scan2map_begin(p0, lm_lookahead=1)
        c.scan2map_begin_chained(ident, ident, lm_lookahead=1)
        a, sa = c.scan2map_end()
        assert sa == 1 and np.array_equal(a, p0)
        b, sb = c.scan2map_end()
        assert sb == 3          # chained behind a frame that was handed back: whatever its own loops did (here: they overflowed too), it began from a pose that is not a result
        # mixing the two kinds of solves: each is collected by its own _end
        c.gn_solve_begin(p0, 3)
        with pytest.raises(mla.MlhError):
            c.scan2map_end()
        c.gn_solve_end()
        assert np.array_equal(c.scan2map(p0, want_stats=False)[0], ref)
    finally:
        c.close()


def test_scan2map_consumer_side_lm_equals_the_classic_launches(mla, case16, feats16, monkeypatch):
    """The Levenberg-Marquardt loop of scan2map in its three forms -- the step in the last workgroup of the launch that evaluated (MLH_LM_CONSUMER=0), in every
    workgroup of the NEXT launch (MLH_LM_LOOP=0), or the whole loop in one launch whose workgroups synchronise among themselves (the default without statistics) --
    is the same arithmetic on the same records: the same pose bits from the synchronous call, from
    the split submission at every look-ahead, and the same look-ahead verdicts (exactly enough / one too few)."""
    p0 = case16["p0"]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    starts = [p0]
    rng = np.random.default_rng(5)
    for _ in range(3):
        d = p0.copy()
        d[:3] += rng.normal(0, 0.05, 3)
        d[3:] += rng.normal(0, 0.004, 4)
        d[3:] /= np.linalg.norm(d[3:])
        starts.append(d)
    c = mla.Context(0)
    try:
        _stage(c, mla, case16, feats16)
        for s0 in starts:
            with_stats, st = c.scan2map(s0)                       # statistics asked for: always the classic launches
            need = max(int(x["lm_iterations"]) for x in st)
            got = {}
            # classic launches | consumer-side launches | the loop as one launch, its records tagged and summed by polling (the default) | ... behind a grid barrier
            for mode in ("00", "10", "11", "11b"):
                monkeypatch.setenv("MLH_LM_CONSUMER", mode[0])
                monkeypatch.setenv("MLH_LM_LOOP", mode[1])
                monkeypatch.setenv("MLH_LOOP_TAGGED", "0" if mode.endswith("b") else "1")
                sync = c.scan2map(s0, want_stats=False)[0]
                c.scan2map_begin(s0)
                split, status = c.scan2map_end()
                assert status == 0
                c.scan2map_begin(s0, lm_lookahead=need)
                exact, status_exact = c.scan2map_end()
                c.scan2map_begin(s0, lm_lookahead=max(need - 1, 1))
                short, status_short = c.scan2map_end()
                c.scan2map_begin(s0)
                c.scan2map_begin_chained(ident, ident)
                a, sa = c.scan2map_end()
                b, sb = c.scan2map_end()
                assert sa == 0 and sb == 0
                got[mode] = (sync, split, exact, status_exact, short, status_short, a, b)
            monkeypatch.delenv("MLH_LOOP_TAGGED", raising=False)
            for other in ("10", "11", "11b"):
                for x, y in zip(got["00"], got[other]):
                    assert np.array_equal(x, y)
            assert np.array_equal(got["11"][0], with_stats)
            assert got["11"][3] == 0 and (need == 1 or got["11"][5] == 2)      # (an explicit look-ahead keeps the launches it counts, in every mode)
    finally:
        c.close()


def test_scan2map_lm_forms_over_loop_lengths_and_degenerate_frames(mla, case16, feats16, monkeypatch):
    """The same three forms of the LM loop where the loop is cut short (max_lm_iterations 1 .. 4: the device loop must end on the iteration count exactly as Ceres' does),
    with one and three outer iterations, without the Huber loss, and on a frame the degeneracy test fires on (a map_eig_thre above the spectrum: the eigen-decomposition
    and the projected update run inside every workgroup of the loop kernel): poses bit for bit those of the launches that also fill the statistics."""
    p0 = case16["p0"]
    c = mla.Context(0)
    try:
        _stage(c, mla, case16, feats16)
        variants = [dict(max_lm_iterations=k) for k in (1, 2, 3, 4)] + [dict(max_outer=1), dict(max_outer=3), dict(huber_delta=0.0), dict(map_eig_thre=1e9),
                                                                         dict(map_eig_thre=3.0e4, max_outer=3)]
        # ... and behind a good-feature selection (the selected rows' loop: one launch per outer iteration; the split submission does not take selections)
        G = mla.GF_METHODS
        gf_variants = [dict(gf_method=G[m_], gf_ratio=0.3, gf_seed=5) for m_ in ("rnd", "fps", "gd_fix")] + [dict(gf_method=G["gd_float"], gf_ratio=0.3, gf_seed=7, max_outer=3)]
        for kw in gf_variants:
            opts = mla.default_opts(**kw)
            monkeypatch.delenv("MLH_LM_CONSUMER", raising=False)
            monkeypatch.delenv("MLH_LM_LOOP", raising=False)
            ref, st = c.scan2map(p0, opts)
            for mode in ("11", "10", "00"):
                monkeypatch.setenv("MLH_LM_CONSUMER", mode[0])
                monkeypatch.setenv("MLH_LM_LOOP", mode[1])
                assert np.array_equal(c.scan2map(p0, opts, want_stats=False)[0], ref), (kw, mode)
        n_deg = 0
        for kw in variants:
            opts = mla.default_opts(**kw)
            monkeypatch.delenv("MLH_LM_CONSUMER", raising=False)
            monkeypatch.delenv("MLH_LM_LOOP", raising=False)
            ref, st = c.scan2map(p0, opts)                         # statistics: the classic launches
            n_deg += sum(int(x["is_degenerate"]) for x in st)
            for mode in ("11", "11b", "10", "00"):
                monkeypatch.setenv("MLH_LM_CONSUMER", mode[0])
                monkeypatch.setenv("MLH_LM_LOOP", mode[1])
                monkeypatch.setenv("MLH_LOOP_TAGGED", "0" if mode.endswith("b") else "1")
                got = c.scan2map(p0, opts, want_stats=False)[0]
                assert np.array_equal(got, ref), (kw, mode)
                c.scan2map_begin(p0, opts)
                split, status = c.scan2map_end()
                assert status == 0 and np.array_equal(split, ref), (kw, mode)
        monkeypatch.delenv("MLH_LOOP_TAGGED", raising=False)
        assert n_deg >= 2                                          # the degenerate variants really took the projected update
    finally:
        c.close()


def test_scan2map_loop_kernel_on_a_context_the_tracker_used_first(mla, case16, feats16, track_case):
    """The arrival counters of the fused finishes and of the LM loop kernel's barrier live in one small buffer that the first user allocates: a context whose first
    solver call is the tracker's must leave ALL of its words allocated and zeroed, or the mapper's loop kernel starts its barrier on whatever the allocation held (latent: found in round 5 by
    reading -- a fresh allocation happens to be zero-filled, so the order below passed before the fix too; there is one allocation helper now, and this pins the order)."""
    tc = track_case
    c = mla.Context(0)
    try:
        c.track_set_prev(mla.CORNER, tc["corner_last"]); c.track_set_prev(mla.SURF, tc["surf_last"])
        c.track_set_cur(mla.CORNER, tc["corner_sharp"]); c.track_set_cur(mla.SURF, tc["surf_flat"])
        ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
        pose_t = c.track_cloud(ident, want_stats=False)[0]
        _stage(c, mla, case16, feats16)
        lean = c.scan2map(case16["p0"], want_stats=False)[0]        # the loop kernel, first launch of its kind on this context
        full, _ = c.scan2map(case16["p0"])
        assert np.array_equal(lean, full)
        assert np.array_equal(c.track_cloud(ident, want_stats=False)[0], pose_t)
    finally:
        c.close()


def test_track_cloud_lm_loop_in_one_launch_equals_the_launches(mla, track_case, monkeypatch):
    """LidarTracker::trackCloud without statistics: a round's LM loop as ONE launch (track_lm_loop_kernel, the default) against 2 + max_lm_iterations launches per
    round (MLH_TRACK_LOOP=0) and against the launches that fill the statistics -- the same pose bits over start poses, round counts, iteration caps, with and without
    the Huber loss, and with too few correspondences for a round to run (distance threshold ~0: every round skipped, the pose comes back as it went in)."""
    tc = track_case
    rng = np.random.default_rng(9)
    c = mla.Context(0)
    try:
        c.track_set_prev(mla.CORNER, tc["corner_last"]); c.track_set_prev(mla.SURF, tc["surf_last"])
        c.track_set_cur(mla.CORNER, tc["corner_sharp"]); c.track_set_cur(mla.SURF, tc["surf_flat"])
        starts = [np.array([0, 0, 0, 0, 0, 0, 1.0])]
        for _ in range(3):
            p = np.array([*rng.normal(0, 0.15, 3), *rng.normal(0, 0.01, 3), 1.0]); p[3:] /= np.linalg.norm(p[3:])
            starts.append(p)
        variants = [dict(), dict(max_outer=1), dict(max_outer=3), dict(max_lm_iterations=1), dict(max_lm_iterations=2), dict(max_lm_iterations=6), dict(huber_delta=0.0),
                    dict(distance_sq_threshold=1e-6)]
        for p0 in starts:
            for kw in variants:
                opts = mla.default_track_opts(**kw)
                monkeypatch.delenv("MLH_TRACK_LOOP", raising=False)
                full, _ = c.track_cloud(p0, opts)
                loop, _ = c.track_cloud(p0, opts, want_stats=False)
                monkeypatch.setenv("MLH_TRACK_LOOP", "0")
                launches, _ = c.track_cloud(p0, opts, want_stats=False)
                assert np.array_equal(loop, launches) and np.array_equal(loop, full), (kw, p0)
    finally:
        c.close()


def test_scan2map_without_stats_matches(ctx, mla, case16, feats16):
    """mlh_scan2map(stats = NULL) takes the Cholesky shortcut for evalDegenracy; the pose must be the one the full procedure gives."""
    _stage(ctx, mla, case16, feats16)
    a, stats = ctx.scan2map(case16["p0"])
    b, none = ctx.scan2map(case16["p0"], want_stats=False)
    assert none is None and not any(s["is_degenerate"] for s in stats)
    np.testing.assert_array_equal(a, b)


def test_track_match_parity(ctx, mla, orc, track_case):
    """(f4) matchCornerFromScan / matchSurfFromScan: validity and the f32 coefficients bit for bit (1-NN, both directional walks with
    their strict-< tie rule, the plane normal in f32)."""
    tc = track_case
    rng = np.random.default_rng(1)
    for pose in (np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.3, -0.1, 0.02, 0.0, 0.0, 0.012, 0.99992800])):
        pose[3:] /= np.linalg.norm(pose[3:])
        for kind, ch, prev, cur in ((mla.CORNER, "c", tc["corner_last"], tc["corner_sharp"]), (mla.SURF, "s", tc["surf_last"], tc["surf_flat"])):
            ctx.track_set_prev(kind, prev)
            ctx.track_set_cur(kind, cur)
            valid, coeffs = ctx.track_match(kind, pose)
            rv, rc = orc.track_match(ch, prev, cur, pose)
            assert rv.sum() > 30
            assert np.array_equal(valid, rv)
            assert np.array_equal(coeffs.astype(np.float32).view(np.uint32), rc.astype(np.float32).view(np.uint32))
    # a cloud that is not ordered by ring id is refused
    bad = tc["corner_last"].copy()
    bad[5, 3] = 9
    with pytest.raises(Exception):
        ctx.track_set_prev(mla.CORNER, bad)


def test_track_cloud_parity(ctx, mla, orc, track_case):
    """(f4) LidarTracker::trackCloud: same correspondences, same LM iteration counts and termination, pose to 1e-9 of the oracle's
    (north-star tolerance 1e-4 m / 1e-4 rad), and both recover the simulated motion."""
    tc = track_case
    ctx.track_set_prev(mla.CORNER, tc["corner_last"]); ctx.track_set_prev(mla.SURF, tc["surf_last"])
    ctx.track_set_cur(mla.CORNER, tc["corner_sharp"]); ctx.track_set_cur(mla.SURF, tc["surf_flat"])
    p0 = np.array([0, 0, 0, 0, 0, 0, 1.0])
    pose, stats = ctx.track_cloud(p0)
    ref = orc.track_cloud(tc["corner_last"], tc["surf_last"], tc["corner_sharp"], tc["surf_flat"], p0)
    assert len(stats) == len(ref["outer"]) == 2
    for s, o in zip(stats, ref["outer"]):
        assert (s["n_corner"], s["n_surf"], s["lm_iterations"], s["termination"]) == (o["n_corner"], o["n_surf"], o["lm_iterations"], o["termination"])
        assert abs(s["cost"] - o["initial_cost"]) <= 1e-9 * max(1.0, o["initial_cost"])
        assert abs(s["final_cost"] - o["final_cost"]) <= 1e-9 * max(1.0, o["final_cost"])
    assert max(_pose_err(pose, ref["pose"])) < 1e-9
    assert np.linalg.norm(pose[:3] - tc["motion"][:3]) < 0.08
    pose2, none = ctx.track_cloud(p0, want_stats=False)      # lean path: pose in through kernel arguments, out from the last launch
    assert none is None
    np.testing.assert_array_equal(pose2, pose)
    # a round with too few correspondences is skipped on both paths (lidar_tracker.cpp:66-70): the pose comes back unchanged
    ctx.track_set_cur(mla.CORNER, tc["corner_sharp"][:3]); ctx.track_set_cur(mla.SURF, tc["surf_flat"][:4])
    p_in = np.array([0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    pose3, st3 = ctx.track_cloud(p_in)
    pose4, _ = ctx.track_cloud(p_in, want_stats=False)
    np.testing.assert_array_equal(pose3, p_in)
    np.testing.assert_array_equal(pose4, p_in)


def test_downsample_current_scan_device_resident(ctx, mla, orc, synth, case16, feats16):
    """(f2) downsampleCurrentScan as one device-resident call: voxel thinning -> evalPointUncertainty -> trace gate -> the kind's
    feature set. Checked piecewise against the oracle, and the resulting feature set must behave exactly like the same points
    handed over through mlh_features_set."""
    rng = np.random.default_rng(12)
    base = feats16[0][:, :3]
    xyz = np.concatenate([base, base + rng.normal(0, 0.08, base.shape).astype(np.float32)])
    pts = np.zeros((len(xyz), 4), np.float32)
    pts[:, :3] = xyz
    pts[:, 3] = (xyz[:, 0] > 0).astype(np.float32)            # LiDAR id constant per region: voxels do not mix ids
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3) * 30])
    meas = np.diag([0.0025] * 3)
    thr = 0.05
    ds = ctx.voxel_filter(pts, 0.4)
    got = ctx.downsample_current_scan(mla.SURF, pts, 0.4, ext, covs, meas, True, thr)
    keep_ref, cov_ref = [], []
    for i, p in enumerate(ds):
        n = int(p[3])
        R = synth.quat_to_rot(ext[n][3:])
        sel = ((p[:3].astype(np.float64) - ext[n][:3]) @ R).astype(np.float32)
        c = orc.eval_point_uncertainty(sel[None, :], ext[n], covs[n], meas)[0]
        keep_ref.append(np.trace(c) <= thr)
        cov_ref.append([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])
    keep_ref, cov_ref = np.array(keep_ref), np.array(cov_ref)
    assert 0 < keep_ref.sum() < len(ds)
    assert len(got) == keep_ref.sum()
    np.testing.assert_array_equal(got[:, :4], ds[keep_ref])
    np.testing.assert_allclose(got[:, 4:10], cov_ref[keep_ref], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(got[:, 10], cov_ref[keep_ref][:, [0, 3, 5]].sum(axis=1), rtol=2e-5)
    # the device-resident feature set == the same records staged from the host
    ctx.map_set(mla.SURF, case16["surf_map"])
    opts_flags = mla.FLAG_WITH_UA if hasattr(mla, "FLAG_WITH_UA") else 2
    a = ctx.match_linearize(mla.SURF, case16["p0"], flags=opts_flags)
    ctx.features_set(mla.SURF, got)
    b = ctx.match_linearize(mla.SURF, case16["p0"], flags=opts_flags)
    assert np.array_equal(a["valid"], b["valid"]) and a["count"] == b["count"]
    np.testing.assert_array_equal(a["H"], b["H"])
    # without uncertainty: nothing dropped, zero covariance
    plain = ctx.downsample_current_scan(mla.SURF, pts, 0.4, ext, covs, meas, False, thr)
    assert len(plain) == len(ds) and not plain[:, 4:].any()


def test_front_end_stays_on_device(mla, orc, track_case):
    """extractCloud -> LidarTracker::trackCloud without host hops: the scan held by the context feeds the tracker device to device
    (mlh_track_set_from_scan) and gives exactly the pose of the host hand-over of the same GPU-extracted clouds."""
    prev, cur = track_case["scans"]
    p0 = np.array([0, 0, 0, 0, 0, 0, 1.0])
    c = mla.Context(0)
    try:
        c.scan_upload(prev.points, prev.scan_start, prev.scan_end); c.extract_run(); exp = c.extract_fetch(); lfp = c.extract_voxel(0.2)
        c.track_set_from_scan(1)                      # previous frame <- this scan
        c.scan_upload(cur.points, cur.scan_start, cur.scan_end); c.extract_run(); exc = c.extract_fetch()
        c.track_set_from_scan(0)                      # current frame <- this scan
        pose_dev, stats_dev = c.track_cloud(p0)
        # host hand-over of the same clouds
        c.track_set_prev(mla.CORNER, prev.points[exp["less_sharp"]]); c.track_set_prev(mla.SURF, lfp)
        c.track_set_cur(mla.CORNER, cur.points[exc["sharp"]]); c.track_set_cur(mla.SURF, cur.points[exc["flat"]])
        pose_host, stats_host = c.track_cloud(p0)
        np.testing.assert_array_equal(pose_dev, pose_host)
        assert [(s["n_corner"], s["n_surf"], s["lm_iterations"]) for s in stats_dev] == [(s["n_corner"], s["n_surf"], s["lm_iterations"]) for s in stats_host]
        assert np.linalg.norm(pose_dev[:3] - track_case["motion"][:3]) < 0.08
    finally:
        c.close()


def _device_to_host(ptr, n_bytes):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = np.empty(n_bytes, np.uint8)
    rc = hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(n_bytes), 2)   # hipMemcpyDeviceToHost
    assert rc == 0
    return buf


def test_mapper_inputs_stay_on_device(mla, orc, synth, case16):
    """extractCloud -> transformCloudFeature -> downsampleCurrentScan -> scan2MapOptimization without a host hop (mlh_fuse_*):
    the fused clouds are bit-equal to the float32 restatement of transformCloudFeature on the host-fetched extraction results,
    and the mapper pose equals the one of the host-staged hand-over of the same clouds."""
    scans = case16["scans"] * 2                       # the same scan through two extrinsics: two LiDARs' worth of features
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
    meas = np.diag([0.0025] * 3)
    c = mla.Context(0)
    try:
        c.map_set(mla.SURF, case16["surf_map"]); c.map_set(mla.CORNER, case16["corner_map"])
        c.fuse_reset()
        ref_surf, ref_corner = [], []
        for i, s in enumerate(scans):
            c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run()
            ex = c.extract_fetch(); lf = c.extract_voxel(0.2)
            c.fuse_add_scan(i, ext[i])
            ref_surf.append(orc.transform_cloud_feature(lf, ext[i], i))
            ref_corner.append(orc.transform_cloud_feature(s.points[ex["less_sharp"]], ext[i], i))
        ref = {mla.SURF: np.concatenate(ref_surf), mla.CORNER: np.concatenate(ref_corner)}
        for kind in (mla.SURF, mla.CORNER):
            dc = c.fused_cloud(kind)
            assert dc.n == len(ref[kind]) > 100
            got = _device_to_host(dc.ptr, dc.n * 16).view(np.float32).reshape(-1, 4)
            np.testing.assert_array_equal(got, ref[kind])
        opts = mla.default_opts(flags=mla.FLAG_WITH_UA)
        # device-resident hand-over
        m_dev = [c.downsample_current_scan(k, c.fused_cloud(k), leaf, ext, covs, meas, True, 0.6, fetch=False) for k, leaf in ((mla.SURF, 0.4), (mla.CORNER, 0.2))]
        pose_dev, _ = c.scan2map(case16["p0"], opts, want_stats=False)
        # host hand-over of the same clouds
        f_host = [c.downsample_current_scan(k, ref[k], leaf, ext, covs, meas, True, 0.6) for k, leaf in ((mla.SURF, 0.4), (mla.CORNER, 0.2))]
        pose_host, _ = c.scan2map(case16["p0"], opts, want_stats=False)
        assert m_dev == [len(f) for f in f_host]
        np.testing.assert_array_equal(pose_dev, pose_host)
        # both kinds through ONE thinning pipeline (mlh_downsample_current_scan_pair on the fused clouds) and through the fall-back of
        # two single calls (host buffers): the same feature sets, the same pose
        for src in ((c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER)), (ref[mla.SURF], ref[mla.CORNER])):
            m_pair = c.downsample_current_scan_pair(src[0], src[1], 0.4, 0.2, ext, covs, meas, True, 0.6)
            assert list(m_pair) == m_dev
            a = c.match_linearize(mla.SURF, case16["p0"], flags=mla.FLAG_WITH_UA)
            b = c.match_linearize(mla.CORNER, case16["p0"], flags=mla.FLAG_WITH_UA)
            pose_pair, _ = c.scan2map(case16["p0"], opts, want_stats=False)
            np.testing.assert_array_equal(pose_pair, pose_host)
            c.features_set(mla.SURF, f_host[0]); c.features_set(mla.CORNER, f_host[1])
            a2 = c.match_linearize(mla.SURF, case16["p0"], flags=mla.FLAG_WITH_UA)
            b2 = c.match_linearize(mla.CORNER, case16["p0"], flags=mla.FLAG_WITH_UA)
            assert np.array_equal(a["valid"], a2["valid"]) and np.array_equal(b["valid"], b2["valid"])
            np.testing.assert_array_equal(a["H"], a2["H"]); np.testing.assert_array_equal(b["H"], b2["H"])
        # thinning + solve as ONE call with no host read between them (mlh_downsample_scan2map: the solve's launches are sized for the input clouds and read the
        # thinned counts on the device): the same counts, the same pose bits, the same staged features -- also with a different start pose, with three outer
        # iterations, and where the fused form does not apply (host buffers: the two calls inside)
        for start, o in ((case16["p0"], opts), (case16["p0"] + np.array([0.05, -0.03, 0.02, 0, 0, 0, 0]), opts), (case16["p0"], mla.default_opts(flags=mla.FLAG_WITH_UA, max_outer=3))):
            c.downsample_current_scan_pair(c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, True, 0.6)
            want = c.scan2map(start, o, want_stats=False)[0]
            got, cnt = c.downsample_scan2map(c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER), 0.4, 0.2, ext, covs, meas, start, o)
            assert list(cnt) == m_dev
            np.testing.assert_array_equal(got, want)
            a3 = c.match_linearize(mla.SURF, case16["p0"], flags=mla.FLAG_WITH_UA)        # the feature sets it left behind are the thinned ones, with their real counts
            assert np.array_equal(a3["valid"], a2["valid"])
            np.testing.assert_array_equal(a3["H"], a2["H"])
        got, cnt = c.downsample_scan2map(ref[mla.SURF], ref[mla.CORNER], 0.4, 0.2, ext, covs, meas, case16["p0"], opts)
        assert list(cnt) == m_dev
        np.testing.assert_array_equal(got, pose_host)
        # a second frame reuses the buffers from the start
        c.fuse_reset()
        assert c.fused_cloud(mla.SURF).n == 0
    finally:
        c.close()


def test_fuse_ring_ranges_of_a_joint_scan(mla, orc, synth, case16):
    """Two LiDARs uploaded as ONE scan (rings back to back) and fused one ring range at a time (mlh_fuse_add_rings) give exactly the
    fused clouds of two separate scans: extraction is per ring, and the ranges cut the joint lists where the scans end."""
    s = case16["scans"][0]
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    both = np.concatenate([s.points, s.points[::-1] * np.float32(1.0)])       # second LiDAR: the same rings traversed backwards
    n1 = len(s.points)
    # ring tables of the reversed copy: ring r of the original occupies [start-5, end+6) -> mirrored
    rs, re_ = s.scan_start - 5, s.scan_end + 6
    start2 = (n1 - re_)[::-1] + 5 + n1
    end2 = (n1 - rs)[::-1] - 6 + n1
    c = mla.Context(0)
    try:
        c.fuse_reset()
        c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run(); c.extract_voxel_run(0.2); c.fuse_add_scan(0, ext[0])
        c.scan_upload(np.ascontiguousarray(both[n1:]), (start2 - n1).astype(np.int32), (end2 - n1).astype(np.int32)); c.extract_run(); c.extract_voxel_run(0.2)
        c.fuse_add_scan(1, ext[1])
        sep = {k: _device_to_host(c.fused_cloud(k).ptr, c.fused_cloud(k).n * 16).view(np.float32).reshape(-1, 4).copy() for k in (mla.SURF, mla.CORNER)}
        c.fuse_reset()
        c.scan_upload(both, np.concatenate([s.scan_start, start2]).astype(np.int32), np.concatenate([s.scan_end, end2]).astype(np.int32))
        c.extract_run(); c.extract_voxel_run(0.2)
        c.fuse_add_rings(0, s.n_rings, 0, ext[0]); c.fuse_add_rings(s.n_rings, 2 * s.n_rings, 1, ext[1])
        for k in (mla.SURF, mla.CORNER):
            dc = c.fused_cloud(k)
            joint = _device_to_host(dc.ptr, dc.n * 16).view(np.float32).reshape(-1, 4)
            assert len(sep[k]) > 100 and set(np.unique(sep[k][:, 3])) == {0.0, 1.0}
            np.testing.assert_array_equal(joint, sep[k])
        with pytest.raises(mla.MlhError):
            c.fuse_add_rings(3, 3, 0, ext[0])
    finally:
        c.close()


def _timed_cloud(rng, n):
    """rows [x y z intensity] with intensity = ring id + time inside the sweep (ImageSegmenter + calTimestamp), SCAN_PERIOD 0.1"""
    pts = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    inten = (rng.integers(0, 64, n) + rng.uniform(0, 0.0999, n)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([pts, inten[:, None]], axis=1))


@pytest.mark.parametrize("distortion", [True, False])
def test_transform_to_end_parity(ctx, orc, distortion):
    """TransformToEnd (utility.h:79-100), the per-point body of Estimator::undistortMeasurements: slerp-interpolated motion inside
    the sweep. f64 math with two f32 roundings, acos / sin from the device's libm: equal to the oracle up to one f32 ulp, and
    bit-equal for the overwhelming majority of the words."""
    rng = np.random.default_rng(21)
    pts = _timed_cloud(rng, 5000)
    q = np.array([0.01, -0.02, 0.03, 1.0]); q /= np.linalg.norm(q)
    pose = np.concatenate([[0.35, -0.12, 0.02], q])
    got = ctx.transform_to_end(pts, pose, distortion)
    ref = orc.transform_to_end(pts, pose, distortion)
    np.testing.assert_array_equal(got[:, 3], pts[:, 3])
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=8e-6)        # 1 ulp at 60 m = 3.8e-6
    assert np.mean(got[:, :3].view(np.uint32) == ref[:, :3].view(np.uint32)) > 0.98
    if not distortion:                                                            # s = 1: T^-1 T p = p up to the two roundings
        np.testing.assert_allclose(got[:, :3], pts[:, :3], rtol=0, atol=2e-5)
    else:
        assert np.abs(got[:, :3] - pts[:, :3]).max() > 0.05
    # the negative-w branch of slerp and the identity short-cut
    for pq in (-q, np.array([0, 0, 0, 1.0])):
        p2 = np.concatenate([[0.1, 0.2, -0.3], pq])
        np.testing.assert_allclose(ctx.transform_to_end(pts[:200], p2, True)[:, :3], orc.transform_to_end(pts[:200], p2, True)[:, :3], rtol=0, atol=8e-6)


def test_scan_undistort_on_device(mla, orc, case16):
    """Estimator::undistortMeasurements without a host hop: the scan's points and its thinned less-flat cloud move to the end of the
    sweep in place; the hand-overs that follow (tracker, fusion) read the undistorted clouds."""
    s = case16["scans"][0]
    rng = np.random.default_rng(5)
    pts = s.points.copy()
    begins = s.scan_start - 5
    for r in range(s.n_rings):
        e = begins[r + 1] if r + 1 < s.n_rings else len(pts)
        pts[begins[r]:e, 3] = r + np.linspace(0, 0.0999, e - begins[r], dtype=np.float32)
    q = np.array([0.0, 0.0, np.sin(np.deg2rad(0.75)), np.cos(np.deg2rad(0.75))])
    pose = np.concatenate([[0.35, -0.12, 0.02], q])
    c = mla.Context(0)
    try:
        c.scan_upload(pts, s.scan_start, s.scan_end); c.extract_run(); ex = c.extract_fetch(); lf = c.extract_voxel(0.2)
        c.fuse_reset(); c.scan_undistort(pose); c.fuse_add_scan(0, np.array([0, 0, 0, 0, 0, 0, 1.0]))
        lf_u = orc.transform_to_end(lf, pose, True)
        cn_u = orc.transform_to_end(pts[ex["less_sharp"]], pose, True)
        for kind, ref in ((mla.SURF, lf_u), (mla.CORNER, cn_u)):
            dc = c.fused_cloud(kind)
            got = _device_to_host(dc.ptr, dc.n * 16).view(np.float32).reshape(-1, 4)
            assert got.shape == ref.shape
            np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=1e-5)      # identity extrinsic: the fusion adds nothing
            assert not got[:, 3].any()
        with pytest.raises(mla.MlhError):
            mla.Context(0).scan_undistort(pose)
    finally:
        c.close()


def test_window_factor_table_built_on_the_device(mla, orc, synth, case16, feats16):
    """Estimator::optimizeMap's factor construction without the host (estimator.cpp:700-780): every (frame, LiDAR, kind) is one match pass
    against the resident local map at T_pivot^-1 T_frame T_ext whose valid correspondences are appended to the LidarPureOdom factor table in
    HBM (mlh_pure_odom_begin / _add_matches). The coupled normal equations of that table must equal those of the table staged from the host
    out of the SAME matches (oracle-checked validity + coefficients), for N_NEIGH 5 and 10, with CHECK_FOV."""
    from scipy.spatial.transform import Rotation as Rot
    c = mla.Context(0)
    try:
        to_pose = lambda T: np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
        c.map_set(mla.SURF, case16["surf_map"])
        c.map_set(mla.CORNER, case16["corner_map"])
        oms, omc = orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"])
        # two "LiDARs": the case's features split in halves, moved into each LiDAR's own frame with a made-up extrinsic
        exts = np.array([[0, 0, 0, 0, 0, 0, 1.0], np.concatenate([[0.3, -0.2, 0.1], Rot.from_rotvec([0.02, -0.01, 0.4]).as_quat()])])
        frame = case16["p0"]
        pivot = np.array([0, 0, 0, 0, 0, 0, 1.0])
        feats = {}
        for kind, f in ((mla.SURF, feats16[0]), (mla.CORNER, feats16[1])):
            half = len(f) // 2
            for n, part in enumerate((f[:half], f[half:])):
                Tinv = np.linalg.inv(synth.pose_to_mat(exts[n]))
                q = part.copy()
                q[:, :3] = synth.transform_points(part[:, :3], Tinv)             # body-frame feature -> LiDAR n's frame
                feats[(kind, n)] = np.ascontiguousarray(q)
        host_tab = [[], [], [], [], []]
        c.pure_odom_begin()
        for n in range(2):
            rel = to_pose(synth.pose_to_mat(frame) @ synth.pose_to_mat(exts[n]))
            kn = 5 if n == 0 else 10
            for kind, ch, om in ((mla.SURF, "s", oms), (mla.CORNER, "c", omc)):
                f = feats[(kind, n)]
                c.features_set(kind, f)
                c.pure_odom_add_matches(kind, rel, 0, n, k_neigh=kn, flags=mla.FLAG_CHECK_FOV)
                v, co = om.match(ch, f, rel, n_neigh=kn, check_fov=True)              # what the device pass must have found
                m = v.astype(bool)
                assert m.sum() > 30
                host_tab[0].append(np.full(m.sum(), kind, np.int32)); host_tab[1].append(f[m, :3].astype(np.float64)); host_tab[2].append(co[m])
                host_tab[3].append(np.zeros(m.sum(), np.int32)); host_tab[4].append(np.full(m.sum(), n, np.int32))
        got = c.pure_odom_normal_eq(pivot, frame[None, :], exts, huber_delta=1.0)
        tab = [np.concatenate(a) for a in host_tab]
        assert got["count"] == len(tab[0])                                           # exactly the oracle's valid correspondences, no padding counted
        ref = orc.pure_odom_normal_eq(tab[0], tab[1], tab[2], None, tab[3], tab[4], pivot, frame[None, :], exts, 1.0)
        sc = float(np.abs(ref["H"]).max())
        assert float(np.abs(got["H"] - ref["H"]).max()) <= 1e-9 * sc
        assert float(np.abs(got["g"] - ref["g"]).max()) <= 1e-9 * float(np.abs(ref["g"]).max())
        assert abs(got["cost"] - ref["cost"]) <= 1e-9 * ref["cost"]
        again = c.pure_odom_normal_eq(pivot, frame[None, :], exts, huber_delta=1.0)
        assert np.array_equal(again["H"], got["H"])                                  # deterministic packing and reduction
        # the host-staged path on the same context afterwards gives the same system (different summation order: 1e-12)
        c.pure_odom_set(*tab)
        host = c.pure_odom_normal_eq(pivot, frame[None, :], exts, huber_delta=1.0)
        assert float(np.abs(host["H"] - got["H"]).max()) <= 1e-12 * sc and host["count"] == got["count"]
        # a device-built table AFTER a host-staged one starts from a clean slate (the tile bookkeeping of the host table must not leak in)
        c.pure_odom_begin()
        for n in range(2):
            rel = to_pose(synth.pose_to_mat(frame) @ synth.pose_to_mat(exts[n]))
            for kind in (mla.SURF, mla.CORNER):
                c.features_set(kind, feats[(kind, n)])
                c.pure_odom_add_matches(kind, rel, 0, n, k_neigh=5 if n == 0 else 10, flags=mla.FLAG_CHECK_FOV)
        rebuilt = c.pure_odom_normal_eq(pivot, frame[None, :], exts, huber_delta=1.0)
        assert np.array_equal(rebuilt["H"], got["H"]) and np.array_equal(rebuilt["g"], got["g"]) and rebuilt["count"] == got["count"]
        # per-factor outputs are refused on a device-built table
        c.pure_odom_begin()
        c.features_set(mla.SURF, feats[(mla.SURF, 0)])
        c.pure_odom_add_matches(mla.SURF, to_pose(synth.pose_to_mat(frame)), 0, 0)
        with pytest.raises(mla.MlhError):
            c.pure_odom_evaluate(pivot, frame[None, :], exts)
    finally:
        c.close()


def test_odometry_good_feature_matching_parity(mla, orc, synth):
    """(a19) Estimator::goodFeatureMatching -- the odometry's selection in front of the window's residual blocks (estimator.cpp:1273-1517; ODOM_GF_RATIO = 0.8 in
    every shipped configuration) -- through mlh_pure_odom_add_matches_gf: all features of a (frame, LiDAR) group matched and their scored rows evaluated on the
    GPU, the draw loop on the host. The same features in the same order as the oracle (which is pinned to the reference's own lines,
    tests/test_oracle_ref_pin.py::test_odometry_good_feature_matching_is_the_references) for surf and corner features at ratios 1.0 / 0.8 / 0.3 / 0.05, and the factor
    table that comes out of it gives the normal equations of exactly the selected correspondences."""
    import conftest
    w = conftest.make_window_case(synth, orc, 1, 2)
    case = conftest._make_case(synth, "50k", 16, 2)
    Tinv = np.linalg.inv(synth.pose_to_mat(case["gt"]))
    maps = [np.ascontiguousarray(synth.transform_points(m[:, :3], Tinv).astype(np.float32)) for m in (case["surf_map"], case["corner_map"])]
    maps4 = []
    for m in maps:
        a = np.zeros((len(m), 4), np.float32); a[:, :3] = m
        maps4.append(a)
    feats = conftest.features_from_extraction(synth, case["scans"][:1], lambda s: orc.extract(s.points, s.scan_start, s.scan_end))
    pivot, pose_i, ext = w["pivot"], w["frames"][0], w["exts"][1]
    # rel_pose as a caller builds it: T_pivot^-1 T_i T_ext (here with numpy; both sides are handed the same one)
    from scipy.spatial.transform import Rotation as Rot
    T = Tinv @ synth.pose_to_mat(pose_i) @ synth.pose_to_mat(ext)
    T = np.linalg.inv(synth.pose_to_mat(pivot)) @ synth.pose_to_mat(pose_i) @ synth.pose_to_mat(ext)
    rel = np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    c = mla.Context(0)
    try:
        c.map_set_pair(maps4[0], maps4[1])
        for kind, ch, mp, f in ((mla.SURF, "s", maps[0], feats[0]), (mla.CORNER, "c", maps[1], feats[1])):
            c.features_set(kind, f)
            om = orc.Map(mp)
            for ratio in (1.0, 0.8, 0.3, 0.05):
                for seed in (1, 7):
                    c.pure_odom_begin()
                    got = c.pure_odom_add_matches_gf(kind, rel, pivot, pose_i, ext, 0, 1, gf_ratio=ratio, seed=seed)
                    ref = orc.odom_good_feature_matching(om, ch, f, rel, pivot, pose_i, ext, ratio, seed)
                    assert np.array_equal(got, ref["sel"]), (ch, ratio, seed, len(got), len(ref["sel"]))
                    assert len(got) > 50
                    # the table holds exactly the selected correspondences: its normal equations equal those of the oracle's factors on that selection
                    ne = c.pure_odom_normal_eq(pivot, np.array([pose_i]), np.array([ident, ext]), huber_delta=1.0)
                    assert ne["count"] == len(got)
                    valid, coeffs = om.match(ch, f, rel)
                    s_ = np.sort(got)
                    want = orc.pure_odom_normal_eq(np.full(len(s_), kind, np.int32), f[s_, :3].astype(np.float64), coeffs[s_], None, np.zeros(len(s_), np.int32),
                                                   np.ones(len(s_), np.int32), pivot, np.array([pose_i]), np.array([ident, ext]), 1.0)
                    assert abs(ne["cost"] - want["cost"]) <= 1e-9 * max(1.0, want["cost"])
                    assert float(np.abs(ne["H"] - want["H"]).max()) <= 1e-9 * float(np.abs(want["H"]).max())
    finally:
        c.close()


def test_window_local_map_building_blocks(ctx, mla, orc, case16):
    """Estimator::buildLocalMap (estimator.cpp:1160-1203) from its parts: every window frame's cloud into the pivot frame
    (pcl::transformPointCloud with the float 4x4: bit for bit), the union thinned by pcl::VoxelGrid<PointXYZI> (same voxels in the
    same order, centroids summed in the member order std::sort leaves: bit for bit), indexed, matched."""
    rng = np.random.default_rng(17)
    base = np.zeros((len(case16["surf_map"]), 4), np.float32)
    base[:, :3] = case16["surf_map"][:, :3]
    frames, poses = [], []
    for i in range(4):
        sel = base[i::4].copy()
        sel[:, 3] = rng.uniform(0, 16, len(sel)).astype(np.float32)
        q = np.array([0.01 * i, -0.02, 0.03 * i, 1.0]); q /= np.linalg.norm(q)
        poses.append(np.concatenate([[0.3 * i, -0.1 * i, 0.02], q]))
        frames.append(sel)
    moved_g = [ctx.transform_point_cloud(f, p) for f, p in zip(frames, poses)]
    moved_r = [orc.transform_point_cloud(f, p) for f, p in zip(frames, poses)]
    for g, r in zip(moved_g, moved_r):
        np.testing.assert_array_equal(g, r)
    union = np.concatenate(moved_g)
    leaf = 0.4 * min(2.0, max(0.75, 1.0 / 192 * float(16 * 1 * 4)))          # the reference's ratio for N_SCANS 16, 1 LiDAR, window 4
    ds_g = ctx.voxel_grid(union, leaf)
    ds_r = orc.voxel_grid(union, leaf)
    assert ds_g.shape == ds_r.shape and len(ds_g) < len(union)
    np.testing.assert_array_equal(ds_g.view(np.uint32), ds_r.view(np.uint32))
    # the thinned cloud as the odometry local map: exact 5-NN through the same index the mapper uses
    ctx.map_set(mla.SURF, np.ascontiguousarray(ds_g[:, :3]))
    qm = ds_g[rng.integers(0, len(ds_g), 300), :3] + rng.normal(0, 0.1, (300, 3)).astype(np.float32)
    idx, d2 = ctx.knn(mla.SURF, qm.astype(np.float32))
    ridx, rd2 = orc.Map(np.ascontiguousarray(ds_g[:, :3])).knn(qm.astype(np.float32))
    within = rd2 < 1.0
    assert within.sum() > 100 and np.array_equal(idx[within], ridx[within])
    # degenerate inputs
    one = ctx.voxel_grid(union[:1], leaf)
    np.testing.assert_array_equal(one, union[:1])
    same = np.repeat(union[:1], 50, axis=0)
    np.testing.assert_allclose(ctx.voxel_grid(same, leaf), union[:1], rtol=1e-6)


def test_full_size_front_end_properties(mla, synth):
    """BASELINE config 2 sizes, front-end rows, properties that need no oracle: undistortion with the identity motion changes nothing
    (bit for bit); fusing with the identity extrinsic reproduces the extractor's clouds; the joint 128-ring scan gives the per-LiDAR
    lists back to back; pcl::VoxelGrid is idempotent (a thinned cloud has one point per voxel: thinning it again returns it bit for
    bit); the device-resident frame is deterministic."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = synth.make_scene(seed=42, **synth.SCENE_PRESETS["500k"])
        gt = synth.gt_body_pose()
        scans = [synth.simulate_scan(sc, gt, synth.HERCULES_BODY_T_LASER[i], 64, seed=7 + i) for i in range(2)]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    c = mla.Context(0)
    try:
        s = scans[0]
        c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run(); ex = c.extract_fetch(); lf = c.extract_voxel(0.2)
        c.scan_undistort(ident)                                    # s = frac(intensity) / period, identity motion: T^-1 T(s) = I
        c.fuse_reset(); c.fuse_add_scan(0, ident)
        for kind, ref in ((mla.SURF, lf), (mla.CORNER, s.points[ex["less_sharp"]])):
            dc = c.fused_cloud(kind)
            got = _device_to_host(dc.ptr, dc.n * 16).view(np.float32).reshape(-1, 4)
            np.testing.assert_array_equal(got[:, :3], ref[:, :3])
            assert not got[:, 3].any()
        # pcl::VoxelGrid idempotence on the thinned less-flat cloud of the whole scan (per ring it was thinned at 0.2; here at 0.4)
        once = c.voxel_grid(lf, 0.4)
        twice = c.voxel_grid(once, 0.4)
        assert len(once) < len(lf)
        np.testing.assert_array_equal(twice, once)
        # joint scan == per-LiDAR scans back to back, and the frame twice gives the same bits
        pts = np.concatenate([x.points for x in scans]); off = len(scans[0].points)
        ss = np.concatenate([scans[0].scan_start, scans[1].scan_start + off]).astype(np.int32)
        se = np.concatenate([scans[0].scan_end, scans[1].scan_end + off]).astype(np.int32)
        runs = []
        for _ in range(2):
            c.scan_upload(pts, ss, se); c.extract_run(); c.extract_voxel_run(0.2)
            c.fuse_reset()
            c.fuse_add_rings(0, 64, 0, ident); c.fuse_add_rings(64, 128, 1, ident)
            runs.append([_device_to_host(c.fused_cloud(k).ptr, c.fused_cloud(k).n * 16).copy() for k in (mla.SURF, mla.CORNER)])
        assert all(np.array_equal(a, b) for a, b in zip(*runs))
        exj = c.extract_fetch()
        c.scan_upload(scans[1].points, scans[1].scan_start, scans[1].scan_end); c.extract_run(); ex1 = c.extract_fetch()
        for key in ("sharp", "less_sharp", "flat"):
            assert np.array_equal(exj[key], np.concatenate([ex[key], ex1[key] + off])), key
    finally:
        c.close()


def _raw_cloud(synth, n_rings, seed, clutter, order="shuffled"):
    """a raw cloud: a simulated scan, part of the points pulled off their surfaces along the ray; shuffled (nothing may be assumed about a driver's order),
    ring by ring in azimuth order, or in firing order (azimuth step by azimuth step, every ring) starting somewhere inside the sweep"""
    scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    s = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[0], n_rings, seed=seed)
    rng = np.random.default_rng(seed)
    pts = s.points.copy()
    pts[:, 3] = rng.uniform(0.0, 0.9, len(pts)).astype(np.float32)          # some intensity payload in [0, 1)
    m = rng.random(len(pts)) < clutter
    pts[m, :3] *= rng.uniform(0.5, 1.3, (int(m.sum()), 1)).astype(np.float32)
    if order == "shuffled":
        return pts[rng.permutation(len(pts))]
    if order == "ring_major":
        return pts
    if order == "ring_major_rev":                                # the same rings, every ring walked the other way round (the other sense of rotation)
        return np.ascontiguousarray(np.concatenate([pts[a:b][::-1] for a, b in zip(s.scan_start - 5, s.scan_end + 6)]))
    if order == "zigzag":                                        # every ring in pieces of 37 points, every other piece reversed: runs of both directions in one row
        out = []
        for a, b in zip(s.scan_start - 5, s.scan_end + 6):
            for k, c in enumerate(range(a, b, 37)):
                piece = pts[c:min(c + 37, b)]
                out.append(piece[::-1] if k & 1 else piece)
        return np.ascontiguousarray(np.concatenate(out))
    az = np.arctan2(pts[:, 1], pts[:, 0])
    az = np.mod(az - az[len(pts) // 3], 2 * np.pi)               # the sweep starts at some point's azimuth, not at a column boundary
    o = np.argsort(az, kind="stable")
    return np.ascontiguousarray(pts[o[::-1] if order == "firing_rev" else o])


@pytest.mark.parametrize("vs", [16, 64])
def test_image_segmenter_points_on_bin_edges(mla, orc, synth, vs):
    """Rows and columns are decided by f32 atan / atan2, where the device's libm and glibc (what the reference runs) may differ in the last ulp: a point within an ulp of
    a bin edge could land in another pixel, a ground pair within an ulp of 10 degrees could flip. The kernels do not decide such points: they go to the host, which
    evaluates the reference's expression with its own libm (segment.hip: seg_pixel_host / seg_ground_host). Here a third of a scan is moved ONTO the edges -- azimuths
    at exact half-columns, elevations at exact row boundaries, vertical neighbours exactly 10 degrees apart -- and the result must still be the oracle's, bit for bit."""
    rng = np.random.default_rng(17)
    pts = _raw_cloud(synth, vs, 4, 0.05, "firing")
    n = len(pts)
    r = np.linalg.norm(pts[:, :3], axis=1).astype(np.float64)
    ha = np.degrees(np.arctan2(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)))
    va = np.degrees(np.arctan(pts[:, 2] / np.hypot(pts[:, 0], pts[:, 1]).astype(np.float64)))
    res = 360.0 / 1800
    pick = rng.random(n) < 0.35
    # azimuth onto the nearest half-column: (ha - 90) / res = k + 0.5
    k = np.floor((ha - 90.0) / res)
    ha_e = np.where(pick, 90.0 + (k + 0.5) * res, ha)
    # elevation onto a row boundary for half of those (16 rings: (va + 15.1) / 2 integer; 64 rings: (2 - va) * 3 + 0.5 integer above -8.83)
    pick_v = pick & (rng.random(n) < 0.5)
    if vs == 16:
        va_e = np.where(pick_v, np.round((va + 15.1) / 2.0) * 2.0 - 15.1, va)
    else:
        va_e = np.where(pick_v & (va > -8.0), 2.0 - (np.round((2.0 - va) * 3.0 + 0.5) - 0.5) / 3.0, va)
    x = r * np.cos(np.radians(va_e)) * np.sin(np.radians(ha_e))
    y = r * np.cos(np.radians(va_e)) * np.cos(np.radians(ha_e))
    z = r * np.sin(np.radians(va_e))
    edge = np.ascontiguousarray(np.stack([x, y, z, pts[:, 3]], 1).astype(np.float32))
    # ground pairs exactly 10 degrees apart: a column of points whose consecutive differences have atan2(dz, dxy) = 10 degrees
    col = []
    for c in range(40):
        az = np.radians(3.7 + c * 1.3)
        base = np.array([8.0 * np.sin(az), 8.0 * np.cos(az), -1.9])
        step = np.array([np.sin(az) * np.cos(np.radians(10.0)), np.cos(az) * np.cos(np.radians(10.0)), np.sin(np.radians(10.0))]) * 0.55
        for j in range(5):
            col.append(base + j * step)
    extra = np.concatenate([np.array(col), np.full((len(col), 1), 0.3)], 1).astype(np.float32)
    cloud = np.ascontiguousarray(np.concatenate([edge, extra]))
    prm = orc.seg_params(vertical_scans=vs, segment_flag=True)
    ref = orc.segment_cloud(cloud, prm)
    c = mla.Context(0)
    try:
        got = c.segment_cloud(cloud, vertical_scans=vs, segment_flag=1)
    finally:
        c.close()
    assert got["cloud"].shape == ref["cloud"].shape and np.array_equal(got["cloud"].view(np.uint32), ref["cloud"].view(np.uint32))
    assert np.array_equal(got["scan_start"], ref["scan_start"]) and np.array_equal(got["scan_end"], ref["scan_end"])
    assert got["outlier"].shape == ref["outlier"].shape and np.array_equal(got["outlier"].view(np.uint32), ref["outlier"].view(np.uint32))


@pytest.mark.parametrize("vs,rings,clutter,order", [(16, 16, 0.1, "ring_major"), (16, 16, 0.4, "firing"), (64, 64, 0.1, "firing"), (64, 64, 0.3, "ring_major"),
                                                    (16, 16, 0.4, "firing_rev"), (64, 64, 0.3, "ring_major_rev"), (64, 64, 0.3, "firing_rev"), (16, 16, 0.3, "zigzag"),
                                                    (64, 64, 0.2, "zigzag")])
def test_image_segmenter_on_ordered_clouds(mla, orc, synth, vs, rings, clutter, order):
    """clouds in the orders drivers really deliver: a ring's fill positions then grow with the column (one step down where the sweep starts) or fall with it (the
    other sense of rotation: `..._rev`), and the outlier erasure -- "erase what is NOW at the position recorded at fill time" (image_segmenter.hpp:374) -- applies a
    whole monotone run at once (segment.hip: seg_rows_kernel; ascending since round 5, descending since round 6) instead of one erasure at a time, which the
    shuffled clouds of the test above and the `zigzag` order here (runs of both directions in one row) come close to. Bit-equal to the oracle every way."""
    pts = _raw_cloud(synth, rings, 5, clutter, order)
    prm = orc.seg_params(vertical_scans=vs, segment_flag=True)
    ref = orc.segment_cloud(pts, prm)
    c = mla.Context(0)
    try:
        got = c.segment_cloud(pts, vertical_scans=vs, segment_flag=1)
    finally:
        c.close()
    assert len(ref["cloud"]) < len(pts) - 100
    assert got["cloud"].shape == ref["cloud"].shape and np.array_equal(got["cloud"].view(np.uint32), ref["cloud"].view(np.uint32))
    assert np.array_equal(got["scan_start"], ref["scan_start"]) and np.array_equal(got["scan_end"], ref["scan_end"])
    assert got["outlier"].shape == ref["outlier"].shape and np.array_equal(got["outlier"].view(np.uint32), ref["outlier"].view(np.uint32))


@pytest.mark.parametrize("vs,rings,clutter,flag", [(16, 16, 0.1, True), (16, 16, 0.4, True), (16, 16, 0.1, False), (32, 64, 0.1, True), (64, 64, 0.1, True)])
def test_image_segmenter_parity(mla, orc, synth, vs, rings, clutter, flag):
    """(f3) ImageSegmenter::segmentCloud: ring-major cloud, ScanInfo and outlier cloud bit-equal to the oracle (itself pinned against the reference's
    own lines), the scan staged on the device feeding extractCloud directly. Device atan / atan2 may differ from glibc in the last ulp: the test
    data keeps every sample away from the bin edges (asserted), which is where an ulp could flip a row, a column or a ground pair."""
    import torch
    torch.cuda.init()
    pts = _raw_cloud(synth, rings, 3, clutter)
    prm = orc.seg_params(vertical_scans=vs, segment_flag=flag)
    ref = orc.segment_cloud(pts, prm)
    # margins: column fractional part away from .5, i.e. (ha - 90) / res away from half-integers
    ha = np.degrees(np.arctan2(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)))
    t = (ha - 90.0) / (360.0 / 1800)
    assert np.min(np.abs((t - np.floor(t)) - 0.5)) > 1e-3
    c = mla.Context(0)
    got = c.segment_cloud(pts, vertical_scans=vs, segment_flag=int(flag))
    assert got["cloud"].shape == ref["cloud"].shape
    assert np.array_equal(got["cloud"].view(np.uint32), ref["cloud"].view(np.uint32))
    assert np.array_equal(got["scan_start"], ref["scan_start"]) and np.array_equal(got["scan_end"], ref["scan_end"])
    assert got["outlier"].shape == ref["outlier"].shape and np.array_equal(got["outlier"].view(np.uint32), ref["outlier"].view(np.uint32))
    if flag:
        assert len(ref["cloud"]) < len(pts) - 100          # clutter really was removed
    if vs == 16:
        # the staged scan feeds extractCloud without leaving the device: same labels as the oracle's extraction of the oracle's cloud
        c.extract_run()
        ex = c.extract_fetch()
        rex = orc.extract(ref["cloud"], ref["scan_start"], ref["scan_end"])
        assert rex["n_ties"] == 0
        for k in ("label", "sharp", "less_sharp", "flat", "less_flat_raw"):
            assert np.array_equal(ex[k], rex[k]), k
    # device-resident input: same answer
    d = torch.from_numpy(pts).cuda()
    torch.cuda.synchronize()
    got2 = c.segment_cloud(d, vertical_scans=vs, segment_flag=int(flag))
    assert np.array_equal(got2["cloud"].view(np.uint32), ref["cloud"].view(np.uint32)) and np.array_equal(got2["outlier"].view(np.uint32), ref["outlier"].view(np.uint32))
    c.close()


def test_scan_upload_ahead_equals_the_plain_upload(mla, orc, case16, track_case):
    """mlh_scan_upload_ahead (round 6): the NEXT scan's points sent to the device beside the current scan's kernels; the mlh_scan_upload that names the same buffer packs
    from what arrived. Same extraction bit for bit as the plain upload (labels, curvature bits, the four lists, the per-ring voxel cloud) over a sequence of frames
    that alternates two different scans; the counter says the look-ahead really served them; an upload of ANOTHER buffer in between drops it (and is correct); a second
    look-ahead replaces the first; a bad buffer is refused."""
    scans = [case16["scans"][0], track_case["scans"][1], case16["scans"][0], track_case["scans"][0]]
    pts = [np.ascontiguousarray(s.points, np.float32) for s in scans]

    def same(a, b):
        assert np.array_equal(a["curvature"].view(np.uint32), b["curvature"].view(np.uint32))
        for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw"):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a["less_flat_ds"].view(np.uint32), b["less_flat_ds"].view(np.uint32))

    plain = mla.Context(0)
    want = [plain.extract(p, s.scan_start, s.scan_end, voxel_leaf=0.2) for p, s in zip(pts, scans)]
    plain.close()
    ctx = mla.Context(0)
    try:
        assert ctx.info()["scan_uploads_from_ahead"] == 0
        ctx.scan_upload_ahead(pts[0])
        for k, s in enumerate(scans):
            ctx.scan_upload(pts[k], s.scan_start, s.scan_end)          # packs from the look-ahead
            ctx.extract_run()
            if k + 1 < len(scans):
                ctx.scan_upload_ahead(pts[k + 1])                       # beside this frame's extraction
            got = ctx.extract_fetch()
            got["less_flat_ds"] = ctx.extract_voxel(0.2)
            same(got, want[k])
        assert ctx.info()["scan_uploads_from_ahead"] == len(scans)
        # another buffer in between: the look-ahead is dropped, both uploads are what a plain upload gives
        ctx.scan_upload_ahead(pts[0])
        same(ctx.extract(pts[1], scans[1].scan_start, scans[1].scan_end, voxel_leaf=0.2), want[1])
        same(ctx.extract(pts[0], scans[0].scan_start, scans[0].scan_end, voxel_leaf=0.2), want[0])
        assert ctx.info()["scan_uploads_from_ahead"] == len(scans)
        # a second look-ahead replaces the first
        ctx.scan_upload_ahead(pts[1])
        ctx.scan_upload_ahead(pts[3])
        same(ctx.extract(pts[3], scans[3].scan_start, scans[3].scan_end, voxel_leaf=0.2), want[3])
        assert ctx.info()["scan_uploads_from_ahead"] == len(scans) + 1
        # a copy of the same data at another address is another buffer
        other = pts[0].copy()
        ctx.scan_upload_ahead(pts[0])
        same(ctx.extract(other, scans[0].scan_start, scans[0].scan_end, voxel_leaf=0.2), want[0])
        assert ctx.info()["scan_uploads_from_ahead"] == len(scans) + 1
        with pytest.raises(mla.MlhError):
            ctx.scan_upload_ahead(np.zeros((0, 4), np.float32))
    finally:
        ctx.close()


def test_fuse_from_other_contexts_equals_the_one_context_sequence(mla, synth, case16, track_case):
    """mlh_fuse_add_scan_from (round 6): every LiDAR segmented and extracted on a context of its own -- here, as the facade's front-end lanes do it, by two threads at
    once -- and one context gathers their mapping features device to device. The fused clouds, the thinned feature counts and the mapper's pose equal, bit for bit, what
    ONE context gives that segments / extracts / fuses the LiDARs one after the other; three frames in a row (the sources' scans are rewritten behind the appends,
    ordered by events, while the gathering context still thins and solves); the guards refuse a source that has nothing extracted."""
    import threading
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
    meas = np.diag([0.0025] * 3)
    rng = np.random.default_rng(5)
    raws = []
    for s in (case16["scans"][0], track_case["scans"][1]):
        p = np.ascontiguousarray(s.points[:, :4], np.float32).copy()
        p[:, 3] = 0.0
        raws.append(np.ascontiguousarray(p[rng.permutation(len(p))]))
    frames = [raws, raws[::-1], raws]                  # three frames; the second swaps the LiDARs' clouds (other sizes in every buffer)
    opts = mla.default_opts(flags=mla.FLAG_WITH_UA)

    def front(c, raw):
        c.segment_cloud(raw, fetch=False)
        c.extract_run()
        c.extract_voxel_run(0.2)

    def mapper(c):
        fs, fc = c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER)
        clouds = [_device_to_host(d.ptr, d.n * 16).view(np.float32).reshape(-1, 4).copy() for d in (fs, fc)]
        m = c.downsample_current_scan_pair(fs, fc, 0.4, 0.2, ext, covs, meas, True, 0.6)
        pose, _ = c.scan2map(case16["p0"], opts, want_stats=False)
        return clouds, list(m), pose

    one = mla.Context(0)
    want = []
    try:
        one.map_set(mla.SURF, case16["surf_map"]); one.map_set(mla.CORNER, case16["corner_map"])
        for fr in frames:
            one.fuse_reset()
            for i, raw in enumerate(fr):
                front(one, raw)
                one.fuse_add_scan(i, ext[i])
            want.append(mapper(one))
    finally:
        one.close()
    main, lanes = mla.Context(0), [mla.Context(0), mla.Context(0)]
    try:
        with pytest.raises(mla.MlhError):
            main.fuse_add_scan_from(lanes[0], 0, ext[0])          # nothing extracted there
        main.map_set(mla.SURF, case16["surf_map"]); main.map_set(mla.CORNER, case16["corner_map"])
        def start_lanes(fr):
            th = [threading.Thread(target=front, args=(lanes[i], fr[i])) for i in range(2)]
            for t in th: t.start()
            return th
        th = start_lanes(frames[0])
        for k, fr in enumerate(frames):
            for t in th: t.join()
            main.fuse_reset()
            for i in range(2):
                main.fuse_add_scan_from(lanes[i], i, ext[i])
            # the NEXT frame's front end starts at once, on the lanes, while the appends may still be reading the lanes' scans and the gathering context has not
            # even been asked for its fused clouds: ordered by the events alone
            th = start_lanes(frames[k + 1]) if k + 1 < len(frames) else []
            clouds, m, pose = mapper(main)
            assert m == want[k][1] and min(m) > 50, (k, m, want[k][1])
            for a, b in zip(clouds, want[k][0]):
                np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
            np.testing.assert_array_equal(pose, want[k][2])
    finally:
        main.close()
        for c in lanes:
            c.close()
