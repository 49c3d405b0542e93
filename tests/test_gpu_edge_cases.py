"""Edge cases of the HIP path against the oracle: skipped / ragged / very long rings, tiny and non-tile-multiple feature sets,
queries with no neighbours, maps too small to match, call-order and argument errors through the C-ABI status codes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx(mla):
    c = mla.Context(0)
    yield c
    c.close()


def _ring_scan(rng, lens, noise=0.02):
    """ragged synthetic rings: points on a noisy circle-ish polyline so that curvatures are generic (no ties)"""
    pts, starts, ends, off = [], [], [], 0
    for r, n in enumerate(lens):
        a = np.linspace(0, 2 * np.pi, n, endpoint=False)
        rad = 8.0 + 3.0 * np.sin(3 * a + r) + (np.abs(np.sin(5 * a)) > 0.95) * 2.0 + rng.normal(0, noise, n)
        p = np.stack([rad * np.cos(a), rad * np.sin(a), np.full(n, 0.3 * r) + rng.normal(0, noise, n), np.zeros(n)], axis=1)
        pts.append(p)
        starts.append(off + 5)
        ends.append(off + n - 6)
        off += n
    return np.concatenate(pts).astype(np.float32), np.array(starts, np.int32), np.array(ends, np.int32)


def _check_extract(ctx, orc, pts, ss, se):
    ref = orc.extract(pts, ss, se)
    assert ref["n_ties"] == 0
    got = ctx.extract(pts, ss, se, voxel_leaf=0.2)
    assert np.array_equal(got["curvature"].view(np.uint32), ref["curvature"].view(np.uint32))
    assert np.array_equal(got["label"], ref["label"])
    assert np.array_equal(got["picked"], ref["picked"])
    for k in ("sharp", "less_sharp", "flat", "less_flat_raw"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["less_flat_ds"].shape == ref["less_flat_ds"].shape
    # the per-ring VoxelGrid sums a voxel's members in the order std::sort leaves them (the default member order): the reference's centroids, every bit
    assert np.array_equal(got["less_flat_ds"].view(np.uint32), ref["less_flat_ds"].view(np.uint32))


def test_extract_ragged_and_skipped_rings(ctx, orc):
    rng = np.random.default_rng(0)
    # ring lengths: a skipped one (end - start < 6 <=> n < 17), the minimum processed one, odd sizes, one long ring
    _check_extract(ctx, orc, *_ring_scan(rng, [16, 17, 23, 64, 301, 1800, 40, 777]))


def test_extract_4000_column_rings(ctx, orc):
    """config_realvehicle_kitti.yaml: horizon_scan 4000 -> ~130 KB of LDS per ring workgroup"""
    rng = np.random.default_rng(1)
    _check_extract(ctx, orc, *_ring_scan(rng, [4000, 3990, 4000]))


@pytest.mark.parametrize("lens", [[900, 850, 1000], [2049, 2500], [4500, 4400]])
def test_extract_ring_lengths_across_sort_widths(ctx, orc, lens):
    """the per-ring voxel sort holds 1, 2, 4 or 8 keys per thread (rings of up to 1024, 2048, 4096, 8192 less-flat points)"""
    rng = np.random.default_rng(len(lens) + lens[0])
    _check_extract(ctx, orc, *_ring_scan(rng, lens))


def test_extract_rejects_bad_ring_tables(ctx, mla):
    rng = np.random.default_rng(2)
    pts, ss, se = _ring_scan(rng, [100, 100])
    with pytest.raises(mla.MlhError):          # not inset from the cloud ends
        ctx.scan_upload(pts, np.array([0, 105], np.int32), se)
    with pytest.raises(mla.MlhError):          # rings closer than the +5/-6 insets: suppression of one could reach the other
        ctx.scan_upload(pts, np.array([5, 96], np.int32), np.array([94, 194], np.int32))


@pytest.mark.parametrize("m", [1, 7, 33, 255, 257, 1000])
def test_small_and_ragged_feature_counts(ctx, mla, orc, case16, feats16, m):
    surf = np.ascontiguousarray(feats16[0][:m])
    ctx.map_set(mla.SURF, case16["surf_map"])
    ctx.features_set(mla.SURF, surf)
    got = ctx.match_linearize(mla.SURF, case16["p0"])
    valid, coeffs = orc.Map(case16["surf_map"]).match("s", surf, case16["p0"])
    assert np.array_equal(got["valid"], valid)
    ref = orc.linearize("s", surf, np.full(m, 0.0075), case16["p0"], valid, coeffs)
    np.testing.assert_allclose(got["H"], ref["H"], rtol=1e-9, atol=1e-8)
    assert got["count"] == ref["count"]


def test_queries_without_neighbours_and_tiny_maps(ctx, mla, orc, case16, feats16):
    # features far outside the map extent: no candidate cell at all
    far = feats16[0][:300].copy()
    far[:, :3] += 5000.0
    ctx.map_set(mla.SURF, case16["surf_map"])
    ctx.features_set(mla.SURF, far)
    out = ctx.match_linearize(mla.SURF, case16["p0"])
    assert out["count"] == 0 and not out["valid"].any() and np.all(out["H"] == 0)
    idx, d2 = ctx.knn(mla.SURF, far[:10, :3])
    assert np.all(idx == -1) and np.all(np.isinf(d2))
    # a map with fewer than 5 points can never produce a correspondence
    tiny = case16["surf_map"][:4].copy()
    ctx.map_set(mla.SURF, tiny)
    ctx.features_set(mla.SURF, feats16[0][:500])
    out = ctx.match_linearize(mla.SURF, case16["p0"])
    assert out["count"] == 0
    # exactly 5 coplanar points: the one query in their middle matches, as in the oracle
    five = np.array([[0, 0, 0], [0.3, 0, 0], [0, 0.3, 0], [0.3, 0.3, 0.001], [0.15, 0.15, 0]], np.float32) + np.float32(10.0)
    q = np.array([[10.1, 10.1, 10.05, 0]], np.float32)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    ctx.map_set(mla.SURF, five)
    ctx.features_set(mla.SURF, q)
    out = ctx.match_linearize(mla.SURF, ident)
    v, c = orc.Map(five).match("s", q, ident)
    assert np.array_equal(out["valid"], v) and v[0] == 1
    assert np.array_equal(out["coeffs"].astype(np.float32).view(np.uint32), c.astype(np.float32).view(np.uint32))


def test_scan2map_needs_a_minimum_map(mla, case16, feats16):
    """scan2MapOptimization does nothing unless the map holds > 50 surf and > 10 corner points (lidar_mapper_keyframe.cpp:429)"""
    c = mla.Context(0)
    c.map_set(mla.SURF, case16["surf_map"][:40])
    c.map_set(mla.CORNER, case16["corner_map"])
    c.features_set(mla.SURF, feats16[0])
    c.features_set(mla.CORNER, feats16[1])
    pose, stats = c.scan2map(case16["p0"])
    assert np.array_equal(pose, case16["p0"])
    assert all(s["n_surf"] == 0 for s in stats)
    c.close()


def test_status_codes(mla, case16, feats16):
    c = mla.Context(0)
    with pytest.raises(mla.MlhError, match="map_set"):
        c.features_set(mla.SURF, feats16[0])
        c.match_linearize(mla.SURF, case16["p0"])
    c.map_set(mla.SURF, case16["surf_map"])
    with pytest.raises(mla.MlhError, match="exceeds"):      # acceptance radius larger than the grid was built for
        c.match_linearize(mla.SURF, case16["p0"], min_match_sq_dis=4.0)
    with pytest.raises(mla.MlhError):                        # stride not a multiple of 4 / too small
        c._ck(c.lib.mlh_map_set(c.h, mla.SURF, case16["surf_map"].ctypes.data, 10, 100, 1.0, mla.MEM_HOST))
    with pytest.raises(mla.MlhError, match="extract"):
        c._scan_n = 10
        c.extract_fetch()
    c.close()


def test_voxel_filter_edge_cases(ctx, orc):
    """VoxelGridCovarianceMLOAM: one point, all points in one voxel, every member over the trace threshold, leaf far too small for
    the extent (the reference returns the input unchanged), bad arguments."""
    one = np.array([[1.0, 2.0, 3.0, 7.0]], np.float32)
    np.testing.assert_array_equal(ctx.voxel_filter(one, 0.4), one)
    rng = np.random.default_rng(0)
    blob = np.zeros((500, 11), np.float32)
    blob[:, :3] = rng.uniform(10.0, 10.3, (500, 3))
    blob[:, 3] = rng.integers(0, 3, 500)
    blob[:, 4] = blob[:, 7] = blob[:, 9] = rng.uniform(0.005, 0.02, 500)     # distinct weights: with equal ones the reference's pick
    blob[:, 10] = 3 * blob[:, 4]                                                # of the "heaviest" member depends on its unstable sort
    got, ref = ctx.voxel_filter(blob, 0.4, 1.0), orc.voxel_grid_cov(blob, 0.4, 1.0)
    assert len(got) == len(ref) <= 8
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    far = np.zeros((4, 4), np.float32)
    far[1, 0] = 3.0e3; far[2, 1] = 3.0e3                            # (3e3 / 0.001)^2 voxels: the voxel index would overflow int32
    np.testing.assert_array_equal(ctx.voxel_filter(far, 0.001), far)
    with pytest.raises(Exception):
        ctx.voxel_filter(np.zeros((0, 4), np.float32), 0.4)
    with pytest.raises(Exception):
        ctx.voxel_filter(one, 0.0)


def test_association_and_odom_edge_cases(ctx, orc, synth):
    """cloudUCTAssociateToMap with a threshold that drops everything / nothing; pure-odom table errors."""
    rng = np.random.default_rng(2)
    pts = np.zeros((300, 11), np.float32)
    pts[:, :3] = rng.uniform(-20, 20, (300, 3))
    pts[:, 3] = rng.integers(0, 2, 300)
    ext = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0.2, -0.4, 0.0, 0, 0, 0.0998334166, 0.9950041653]])
    ext_cov = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.0003] * 3)])
    pg, cg, meas = np.array([1.0, 2.0, 0.5, 0, 0, 0, 1.0]), np.eye(6) * 1e-6, np.diag([0.0025] * 3)
    assert len(ctx.cloud_uct_associate_to_map(pts, pg, cg, ext, ext_cov, meas, True, 1e-9)) == 0
    allp = ctx.cloud_uct_associate_to_map(pts, pg, cg, ext, ext_cov, meas, True, 1e9)
    assert len(allp) == 300
    np.testing.assert_allclose(allp[:, :3], pts[:, :3] + pg[:3].astype(np.float32), atol=1e-5)
    with pytest.raises(Exception):
        ctx.pure_odom_set(np.array([2], np.int32), np.zeros((1, 3)), np.zeros((1, 6)), np.array([0], np.int32), np.array([0], np.int32))
    with pytest.raises(Exception):
        ctx.pure_odom_set(np.array([0], np.int32), np.zeros((1, 3)), np.zeros((1, 6)), np.array([-1], np.int32), np.array([0], np.int32))


def test_tracker_edge_cases(ctx, mla, orc, track_case):
    """trackCloud with too few correspondences (threshold so small that nothing matches): both rounds are skipped and the pose
    comes back unchanged, as the reference's `continue`; single-point clouds; a previous cloud confined to one ring."""
    tc = track_case
    ctx.track_set_prev(mla.CORNER, tc["corner_last"]); ctx.track_set_prev(mla.SURF, tc["surf_last"])   # index built for 25 m^2, queried with less
    ctx.track_set_cur(mla.CORNER, tc["corner_sharp"]); ctx.track_set_cur(mla.SURF, tc["surf_flat"])
    p0 = np.array([0.1, 0.2, 0.3, 0, 0, 0, 1.0])
    pose, stats = ctx.track_cloud(p0, mla.default_track_opts(distance_sq_threshold=1e-6))
    np.testing.assert_array_equal(pose, p0)
    assert all(s["lm_iterations"] == 0 and s["n_corner"] + s["n_surf"] < 10 for s in stats)
    # one ring only: the corner walk has no other ring to offer -> no corner correspondences; the surf walk needs a lower/higher ring too
    ring0 = tc["corner_last"][tc["corner_last"][:, 3] == tc["corner_last"][0, 3]]
    ctx.track_set_prev(mla.CORNER, ring0)
    valid, _ = ctx.track_match(mla.CORNER, np.array([0, 0, 0, 0, 0, 0, 1.0]))
    rv, _ = orc.track_match("c", ring0, tc["corner_sharp"], np.array([0, 0, 0, 0, 0, 0, 1.0]))
    assert not valid.any() and not rv.any()
    ctx.track_set_prev(mla.CORNER, tc["corner_last"][:1])
    ctx.track_set_cur(mla.CORNER, tc["corner_sharp"][:1])
    valid, _ = ctx.track_match(mla.CORNER, np.array([0, 0, 0, 0, 0, 0, 1.0]))
    assert not valid.any()


def test_map_staging_paths_agree(mla, synth, case16):
    """mlh_map_set / mlh_map_set_pair: first call lays the grid box out (bounds pass), later calls reuse it while the cloud fits, a cloud
    that outgrows it is detected on the device and re-laid -- the k-NN answers never depend on which way a call went."""
    rng = np.random.default_rng(5)
    surf, corner = case16["surf_map"], case16["corner_map"]
    q = (surf[rng.choice(len(surf), 4000, replace=False)] + rng.normal(0, 0.15, (4000, 3))).astype(np.float32)
    ref_ctx = mla.Context(0)
    ref_ctx.map_set(mla.SURF, surf)
    ref = ref_ctx.knn(mla.SURF, q)
    c = mla.Context(0)
    c.map_set_pair(surf, corner)                       # bounds pass for both
    a = c.knn(mla.SURF, q)
    assert np.array_equal(a[0], ref[0]) and np.array_equal(a[1].view(np.uint32), ref[1].view(np.uint32))
    sub = np.ascontiguousarray(surf[: len(surf) // 2])
    c.map_set_pair(sub, corner)                        # fits the existing box: no host round trip for the bounds
    ref_ctx.map_set(mla.SURF, sub)                     # (this one reuses its own box too)
    a, r = c.knn(mla.SURF, q), ref_ctx.knn(mla.SURF, q)
    assert np.array_equal(a[0], r[0]) and np.array_equal(a[1].view(np.uint32), r[1].view(np.uint32))
    fresh = mla.Context(0)
    fresh.map_set(mla.SURF, sub)                       # a context that never saw the larger cloud: different box, same answers
    f = fresh.knn(mla.SURF, q)
    assert np.array_equal(a[0], f[0]) and np.array_equal(a[1].view(np.uint32), f[1].view(np.uint32))
    grown = np.concatenate([surf, surf[:500] + np.array([40.0, -35.0, 6.0], np.float32)]).astype(np.float32)
    c.map_set_pair(grown, corner)                      # outgrows the box: detected on the device, box laid out again
    fresh.map_set(mla.SURF, grown)
    qq = np.concatenate([q, grown[-200:] + 0.05]).astype(np.float32)
    a, f = c.knn(mla.SURF, qq), fresh.knn(mla.SURF, qq)
    assert np.array_equal(a[0], f[0]) and np.array_equal(a[1].view(np.uint32), f[1].view(np.uint32))
    assert np.all(a[1][-200:, 0] < 0.1)                # the added points are found
    bad = grown.copy()
    bad[17, 1] = np.nan
    with pytest.raises(mla.MlhError):
        c.map_set(mla.SURF, bad)                       # non-finite coordinates are still rejected (through the bounds pass)
    for x in (ref_ctx, c, fresh):
        x.close()


def _quantised_scan(case16):
    """coordinates snapped to a 1/32 m lattice: sums of such values are exact in f32, so equal geometry gives EXACTLY equal curvatures --
    flat walls and ground arcs produce hundreds of ties, and zero curvature on perfectly collinear runs"""
    sc = case16["scans"][0]
    pts = sc.points.copy()
    pts[:, :3] = np.round(pts[:, :3] * 32.0) / 32.0
    return pts, sc.scan_start, sc.scan_end


def test_extract_with_exact_curvature_ties(mla, orc, case16):
    """extractCloud on quantised data ("feature labels bit-exact" includes the reference's order among EQUAL curvatures). The reference's
    comparator sees the curvature only, so tied points are met in the order libstdc++'s introsort leaves them (feature_extract.cpp:152-162). Default:
    the HIP path re-orders every sector that has a tie with that algorithm on the device and equals the oracle's literal std::sort call (tie_rule=0) AND
    extractCloud compiled from the reference's own lines. Opt-out (mlh_set_extract_tie_order(ctx, 0)): the (curvature, index) rule, oracle tie_rule=1."""
    pts, ss, se = _quantised_scan(case16)
    ref0 = orc.extract(pts, ss, se, tie_rule=0)
    ref1 = orc.extract(pts, ss, se, tie_rule=1)
    assert ref0["n_ties"] > 500, ref0["n_ties"]
    keys = ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw")
    assert not all(np.array_equal(ref0[k], ref1[k]) for k in keys)          # the two rules really differ on this scan
    c = mla.Context(0)
    try:
        got = c.extract(pts, ss, se, voxel_leaf=0.2)
        assert np.array_equal(got["curvature"].view(np.uint32), ref0["curvature"].view(np.uint32))
        for k in keys:
            assert np.array_equal(got[k], ref0[k]), k
        np.testing.assert_array_equal(got["less_flat_ds"].view(np.uint32), ref0["less_flat_ds"].view(np.uint32))
        if orc.ref_lib() is not None:                                        # ... and the reference's own lines (oracle/_ref), point for point
            want = orc.ref_extract(pts, ss, se)
            for k in ("sharp", "less_sharp", "flat"):
                assert np.array_equal(want[k].view(np.uint32), np.ascontiguousarray(pts[got[k]]).view(np.uint32)), k
            np.testing.assert_array_equal(got["less_flat_ds"].view(np.uint32), want["less_flat_ds"].view(np.uint32))
        again = c.extract(pts, ss, se)
        assert all(np.array_equal(again[k], got[k]) for k in keys)          # deterministic
        c.set_extract_tie_order(False)
        alt = c.extract(pts, ss, se)
        for k in keys:
            assert np.array_equal(alt[k], ref1[k]), k
        c.set_extract_tie_order(True)
        # both rules refine the reference's comparator: on tie-free data they change nothing
        sc = case16["scans"][0]
        a, b = orc.extract(sc.points, sc.scan_start, sc.scan_end, tie_rule=0), orc.extract(sc.points, sc.scan_start, sc.scan_end, tie_rule=1)
        assert a["n_ties"] == 0 and all(np.array_equal(a[k], b[k]) for k in keys)
        g = c.extract(sc.points, sc.scan_start, sc.scan_end)
        assert all(np.array_equal(g[k], a[k]) for k in keys)
    finally:
        c.close()


def test_extract_ties_on_long_and_short_rings(mla, orc, synth):
    """the reference-order pass at the sizes where its scratch moves: a 4000-column ring (sectors of ~660 points, scratch in HBM), 64 quantised rings
    (scratch in LDS, every sector tied), and rings so short that one wavefront walks the sectors in sequence"""
    scn = synth.make_scene(seed=9, **synth.SCENE_PRESETS["50k"])
    keys = ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw")
    c = mla.Context(0)
    try:
        for rings, cols, q in ((16, 4000, 32.0), (64, 1800, 32.0), (16, 40, 1.0)):
            sc = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[0], rings, seed=31, n_cols=cols)
            pts = sc.points.copy()
            pts[:, :3] = np.round(pts[:, :3] * q) / q
            ref = orc.extract(pts, sc.scan_start, sc.scan_end, tie_rule=0)
            assert ref["n_ties"] > 5, (rings, cols, ref["n_ties"])
            got = c.extract(pts, sc.scan_start, sc.scan_end)
            for k in keys:
                assert np.array_equal(got[k], ref[k]), (rings, cols, k)
    finally:
        c.close()


def test_extract_with_non_finite_points(mla, orc, case16):
    """NaN / inf input points (a driver's "no return" marker): their curvature and that of their 10 neighbours is NaN; NaN is neither
    > 0.1 nor < 0.1, so those points are never labelled, sort above every number (by index among themselves) and a NaN gap never breaks a
    suppression run (NaN > 0.05 is false) -- the same decisions in the oracle (tie_rule=1, where NaN has a defined place) and on the GPU."""
    sc = case16["scans"][0]
    pts = sc.points.copy()
    rng = np.random.default_rng(3)
    bad = rng.choice(len(pts), 40, replace=False)
    pts[bad[:30], :3] = np.nan
    pts[bad[30:], 0] = np.inf
    ref = orc.extract(pts, sc.scan_start, sc.scan_end, tie_rule=1)
    c = mla.Context(0)
    c.set_extract_tie_order(False)
    got = c.extract(pts, sc.scan_start, sc.scan_end)
    c.set_extract_tie_order(True)
    got_ref_order = c.extract(pts, sc.scan_start, sc.scan_end)
    c.close()
    # default order: the reference's std::sort with a comparator that NaN makes inconsistent -- the same comparison sequence gives the same (not even
    # sorted) list, and the walks go through all of it as the reference's loops do (oracle tie_rule=0 = the literal call)
    ref0 = orc.extract(pts, sc.scan_start, sc.scan_end, tie_rule=0)
    for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw"):
        assert np.array_equal(got_ref_order[k], ref0[k]), k
    nan_ref, nan_got = np.isnan(ref["curvature"]), np.isnan(got["curvature"])
    assert nan_ref.sum() >= 40 and np.array_equal(nan_ref, nan_got)
    assert np.array_equal(got["curvature"][~nan_got].view(np.uint32), ref["curvature"][~nan_ref].view(np.uint32))
    for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw"):
        assert np.array_equal(got[k], ref[k]), k
    assert not np.any(got["label"][nan_got] != 0)          # a NaN curvature is never an edge nor a flat point


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", ["8", "16", "32"])
def test_bounded_search_of_later_iterations_on_tie_heavy_maps(mla, orc, lanes):
    """Iterations >= 1 of a solve bound the 5-NN search by the previous iteration's neighbours (knn_group_bounded). On a map of lattice points, duplicated points and a
    wall of coplanar samples -- squared distances tie exactly, also AT the bound -- and with a pose that moves centimetres between iterations, the neighbours (hence the
    fits, counts and poses) must be what the cold search finds: per-iteration counts == the oracle's, and the pose bit-equal between the schedules with and without the
    bound (and with and without the finish in the consumer)."""
    rng = np.random.default_rng(9)
    g = np.arange(-16, 17, dtype=np.float32) * 0.25
    yy, zz = np.meshgrid(g, g[:17] + 4.0, indexing="ij")
    wall = np.stack([np.full(yy.size, 3.0, np.float32), yy.ravel(), zz.ravel()], 1)                    # a lattice wall x = 3
    xx, yy2 = np.meshgrid(g, g, indexing="ij")
    floor = np.stack([xx.ravel(), yy2.ravel(), np.full(xx.size, -1.5, np.float32)], 1)                # a lattice floor z = -1.5 (the reference's plane fit solves n.p = -1: not through the origin)
    wall2 = np.stack([yy.ravel(), np.full(yy.size, -3.5, np.float32), zz.ravel()], 1)                  # a second wall y = -3.5
    surf_map = np.concatenate([wall, floor, wall2, floor[rng.choice(len(floor), 300, replace=False)], wall[rng.choice(len(wall), 200, replace=False)]]).astype(np.float32)
    surf_map = np.ascontiguousarray(surf_map[rng.permutation(len(surf_map))])
    edge = np.stack([np.full(60, 3.0, np.float32), np.full(60, -3.5, np.float32), (np.arange(60, dtype=np.float32) * 0.125 + 4.0)], 1)     # the walls' common edge, lattice spaced
    corner_map = np.ascontiguousarray(np.concatenate([edge, edge[::3], edge + np.array([0, 7.5, 0], np.float32)]), np.float32)
    # features: points ON the surfaces at cell centres / edge midpoints of the lattices (many equidistant neighbours), seen from a slightly wrong pose
    fs = np.concatenate([np.stack([np.full(900, 3.0), rng.integers(-12, 12, 900) * 0.25 + 0.125, rng.integers(1, 15, 900) * 0.25 + 0.125], 1),
                         np.stack([rng.integers(-12, 12, 1500) * 0.25 + 0.125, rng.integers(-12, 12, 1500) * 0.25 + 0.125, np.full(1500, -1.5)], 1),
                         np.stack([rng.integers(-12, 12, 900) * 0.25 + 0.125, np.full(900, -3.5), rng.integers(1, 15, 900) * 0.25], 1)]).astype(np.float32)
    fc = np.stack([np.full(40, 3.0), np.full(40, -3.5), rng.integers(34, 88, 40) * 0.0625 + 2.0], 1).astype(np.float32)
    f4s = np.concatenate([fs, np.zeros((len(fs), 1), np.float32)], 1)
    f4c = np.concatenate([fc, np.zeros((len(fc), 1), np.float32)], 1)
    p0 = np.array([0.04, -0.03, 0.05, 0.004, -0.003, 0.002, 1.0]); p0[3:] /= np.linalg.norm(p0[3:])
    n_it = 4
    ref = orc.gn_iterations(orc.Map(surf_map), orc.Map(corner_map), f4s, f4c, p0, orc.mapper_params(), n_it)
    assert ref["iters"][0]["n_surf"] > 1000 and ref["iters"][0]["n_corner"] > 10 and not ref["iters"][-1]["is_degenerate"]
    poses = {}
    for sched in ((0, 0, 0), (0, 1, 0), (1, 1, 1)):
        os.environ["MLH_KNN_LANES"] = lanes
        try:
            c = mla.Context(0)
        finally:
            os.environ.pop("MLH_KNN_LANES", None)
        try:
            c.set_gn_schedule(*sched)
            c.map_set_pair(surf_map, corner_map)
            c.features_set(mla.SURF, f4s)
            c.features_set(mla.CORNER, f4c)
            poses[sched] = c.gn_solve(p0, n_it, want_stats=False)[0]
            _, st = c.gn_solve(p0, n_it, want_stats=True)
            c.gn_solve_begin(p0, n_it)
            assert np.array_equal(c.gn_solve_end(), poses[sched])
        finally:
            c.close()
        assert [(x["n_surf"], x["n_corner"]) for x in st] == [(r["n_surf"], r["n_corner"]) for r in ref["iters"]], sched
    assert np.array_equal(poses[(0, 1, 0)], poses[(0, 0, 0)]) and np.array_equal(poses[(1, 1, 1)], poses[(0, 0, 0)])
    assert float(np.abs(poses[(1, 1, 1)] - ref["pose"]).max()) < 1e-7


@pytest.mark.gpu
def test_knn_with_exact_distance_ties(mla, orc):
    """Exact ties in the squared distance -- duplicated map points, and queries at the centres of a lattice whose points are exactly
    representable -- are where a k-NN's order is a convention: FLANN's result order among equal distances is implementation-defined, the
    oracle's kd-tree and the HIP search both order by (distance, map index). This pins the two on that rule (indices AND order, bit-equal
    distances), with ties straddling the K-th place, for both lane widths of the search and through the matching path."""
    rng = np.random.default_rng(3)
    g = np.arange(-6, 7, dtype=np.float32) * 0.25                       # 13^3 lattice, spacing exactly representable
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    cloud = np.concatenate([lat, lat[rng.choice(len(lat), 400, replace=False)], rng.uniform(-1.5, 1.5, (500, 3)).astype(np.float32)])
    cloud = np.ascontiguousarray(cloud[rng.permutation(len(cloud))], np.float32)          # duplicates end up at unrelated indices
    m4 = np.zeros((len(cloud), 4), np.float32)
    m4[:, :3] = cloud
    # queries: cell centres (8 lattice points at the same distance), edge midpoints (2 or 4 ties), lattice points themselves (0 + 6 ties)
    c = lat[rng.choice(len(lat), 300, replace=False)]
    q = np.concatenate([c + np.float32(0.125), c + np.array([0.125, 0, 0], np.float32), c]).astype(np.float32)
    q = q[np.all(np.abs(q) < 1.4, axis=1)]
    om = orc.Map(m4)
    ridx, rd2 = om.knn(q, 5)
    n_tied = int(np.sum(rd2[:, 3] == rd2[:, 4])) + int(np.sum(np.any(np.diff(rd2, axis=1) == 0, axis=1)))
    assert n_tied > 200                                                                    # the case is about ties
    for lanes in ("8", "16", "32"):
        os.environ["MLH_KNN_LANES"] = lanes
        try:
            c_ = mla.Context(0)
        finally:
            os.environ.pop("MLH_KNN_LANES", None)
        try:
            c_.map_set(mla.SURF, m4)
            idx, d2 = c_.knn(mla.SURF, q)
            assert np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))
            assert np.array_equal(idx, ridx), f"{int(np.sum(np.any(idx != ridx, axis=1)))} queries order tied neighbours differently ({lanes} lanes)"
            # the matching path (pruned search inside the correspondence kernel) on the same tie-heavy data
            f4 = np.zeros((len(q), 4), np.float32)
            f4[:, :3] = q
            c_.features_set(mla.SURF, f4)
            ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
            got = c_.match_linearize(mla.SURF, ident, dense=False)
            v, co = om.match("s", f4, ident)
            assert np.array_equal(got["valid"], v)
            mm = v.astype(bool)
            assert np.array_equal(got["coeffs"][mm].astype(np.float32).view(np.uint32), co[mm].astype(np.float32).view(np.uint32))
        finally:
            c_.close()


@pytest.mark.gpu
def test_map_info_and_lane_choice(mla, case16, feats16):
    """mlh_map_info: the occupancy statistics the index build collects (non-empty cells, population of the cell an average map point lives in)
    are there after the first staging call (bounds pass, read back directly) and after a repeated one (sticky grid box: carried over by the
    pinned mirror), equal a numpy count on the same grid rule, and the lanes-per-query choice follows them: 16 for a small launch whatever the
    density, 8 for a sparse kind / 16 for a dense kind once the launch is large."""
    c = mla.Context(0)
    try:
        surf, corner = case16["surf_map"], case16["corner_map"]
        c.map_set_pair(surf, corner)
        first = (c.map_info(mla.SURF), c.map_info(mla.CORNER))
        c.map_set_pair(surf, corner)                       # second call: geometry reused, no bounds pass
        c.map_set_pair(surf, corner)
        again = (c.map_info(mla.SURF), c.map_info(mla.CORNER))
        for a, b, cloud in zip(first, again, (surf, corner)):
            assert a["n"] == b["n"] == len(cloud)
            assert 0 < a["occupied_cells"] <= len(cloud) and a["occupied_cells"] == b["occupied_cells"]
            assert a["mean_cell_population"] >= 1.0 and a["mean_cell_population"] == b["mean_cell_population"]
            # the same statistics by numpy: cells of edge 1.001 m (any origin gives the same ORDER of magnitude; the exact box is internal)
            ijk = np.floor(cloud[:, :3] / 1.001).astype(np.int64)
            _, cnt = np.unique(ijk, axis=0, return_counts=True)
            assert 0.5 < a["occupied_cells"] / len(cnt) < 2.0
            assert 0.5 < a["mean_cell_population"] / (float((cnt.astype(np.float64) ** 2).sum()) / len(cloud)) < 2.0
        # small launch (the 16-ring frame's ~5 k features): 16 lanes for both kinds
        c.features_set(mla.SURF, feats16[0]); c.features_set(mla.CORNER, feats16[1])
        assert c.map_info(mla.SURF)["knn_lanes"] == 16 and c.map_info(mla.CORNER)["knn_lanes"] == 16
        # large launch: the density decides
        big = np.ascontiguousarray(np.tile(feats16[0], (4, 1)))
        c.features_set(mla.SURF, big)
        for kind in (mla.SURF, mla.CORNER):
            info = c.map_info(kind)
            expect = 8 if 9.0 * info["mean_cell_population"] < 128 else 16
            assert info["knn_lanes"] == expect
    finally:
        c.close()


def test_std_sort_permutation_edges(mla, orc):
    """mlh_std_sort_permutation: empty first / second cloud, a single element, ranges right at the 16-element insertion threshold and at
    the 64- and 2048-element boundaries the device path switches strategy on; invalid arguments are refused, not guessed at."""
    rng = np.random.default_rng(3)
    c = mla.Context(0)
    try:
        for n in (1, 2, 16, 17, 63, 64, 65, 2047, 2048, 2049, 4097):
            keys = rng.integers(0, max(2, n // 3), n).astype(np.int32)
            want = orc.std_sort_permutation(keys)
            for n0 in (0, n):
                for mode in (1, 2):
                    np.testing.assert_array_equal(c.std_sort_permutation(keys, n0=n0, mode=mode), want)
            h = n // 2
            both = np.concatenate([orc.std_sort_permutation(keys[:h]), h + orc.std_sort_permutation(keys[h:])])
            np.testing.assert_array_equal(c.std_sort_permutation(keys, n0=h, mode=1), both)
        with pytest.raises(mla.MlhError):
            c.std_sort_permutation(np.zeros(4, np.int32), n0=5)
        with pytest.raises(mla.MlhError):
            c.std_sort_permutation(np.zeros(4, np.int32), mode=3)
        with pytest.raises(mla.MlhError):
            c.set_voxel_member_order(3)
    finally:
        c.close()



def test_segmenter_outlier_cloud_of_an_azimuth_decimated_scan(mla, orc, synth):
    """laser_cloud_outlier holds one point per infeasible-cluster pixel whose column is a multiple of 5 (image_segmenter.hpp:
    366-378) -- on a cloud whose points ALL sit in such columns (a sensor with 1 degree azimuth steps at HORIZON_SCAN 1800) that is far more
    than n / 5. The call takes the rows the caller has room for: with the bound the header gives nothing is truncated and the result equals the
    oracle's; with a short buffer the writes stop at the capacity and n_outlier still reports the true count."""
    scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
    s = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[0], 16, seed=5)
    rng = np.random.default_rng(5)
    pts = s.points.copy()
    pts[:, 3] = 0.25
    ha = np.degrees(np.arctan2(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)))
    t = (ha - 90.0) / (360.0 / 1800)
    col = (-np.round(t) + 900).astype(np.int64)
    col[col >= 1800] -= 1800
    keep = (col % 5 == 0) & (np.abs((t - np.floor(t)) - 0.5) > 1e-3)
    pts = pts[keep]
    pts[:, :3] *= rng.uniform(0.4, 1.6, (len(pts), 1)).astype(np.float32)      # every point pulled off its surface: isolated pixels -> small clusters
    pts = pts[rng.permutation(len(pts))]
    prm = orc.seg_params(vertical_scans=16, segment_flag=True)
    ref = orc.segment_cloud(pts, prm)
    n = len(pts)
    assert len(ref["outlier"]) > n // 5 + 3, (len(ref["outlier"]), n)           # the old "n / 5 + 2" bound really is exceeded
    assert len(ref["outlier"]) <= min(n, 16 * 360) + 1
    c = mla.Context(0)
    try:
        got = c.segment_cloud(pts, vertical_scans=16, segment_flag=1)
        assert got["n_outlier"] == len(ref["outlier"])
        assert np.array_equal(got["outlier"].view(np.uint32), ref["outlier"].view(np.uint32))
        assert np.array_equal(got["cloud"].view(np.uint32), ref["cloud"].view(np.uint32))
        short = c.segment_cloud(pts, vertical_scans=16, segment_flag=1, outlier_capacity=7)
        assert short["n_outlier"] == len(ref["outlier"]) and len(short["outlier"]) == 7
        assert np.array_equal(short["outlier"].view(np.uint32), ref["outlier"][:7].view(np.uint32))
    finally:
        c.close()


def test_gn_solve_submitted_and_collected_separately(mla, case16, feats16):
    """mlh_gn_solve_begin / _end: the same iterations as mlh_gn_solve, bit for bit -- also with the NEXT frame's map staging enqueued (and its hand-shake
    answered) between submission and collection, which is how bench.py keeps the GPU busy across frame boundaries; misuse is refused, not guessed at."""
    c = mla.Context(0)
    try:
        c.map_set_pair(case16["surf_map"], case16["corner_map"])
        c.features_set(mla.SURF, feats16[0]); c.features_set(mla.CORNER, feats16[1])
        want, _ = c.gn_solve(case16["p0"], 4, want_stats=False)
        with pytest.raises(mla.MlhError):
            c.gn_solve_end()                                         # nothing in flight
        c.gn_solve_begin(case16["p0"], 4)
        c.gn_solve_begin(case16["p0"], 4)                            # two in flight are allowed (frame k + 1 submitted before frame k is collected) ...
        with pytest.raises(mla.MlhError):
            c.gn_solve_begin(case16["p0"], 4)                        # ... a third is not
        assert np.array_equal(c.gn_solve_end(), want) and np.array_equal(c.gn_solve_end(), want)
        for _ in range(3):                                           # frames back to back: stage k + 1 behind solve k, then read pose k
            c.gn_solve_begin(case16["p0"], 4)
            c.map_set_pair(case16["surf_map"], case16["corner_map"])
            assert np.array_equal(c.gn_solve_end(), want)
        again, _ = c.gn_solve(case16["p0"], 4, want_stats=False)     # the synchronous call still works afterwards
        assert np.array_equal(again, want)
        # overlapped staging: the next frame's maps go to the other map set on a second stream while the solve runs. Alternate two DIFFERENT map pairs
        # (the second one shifted by 3 cm, so its solve has a different answer): every solve must see exactly the maps staged for it.
        moved_s, moved_c = case16["surf_map"].copy(), case16["corner_map"].copy()
        moved_s[:, 0] += 0.03; moved_c[:, 0] += 0.03
        c.map_set_pair(moved_s, moved_c)
        want_moved, _ = c.gn_solve(case16["p0"], 4, want_stats=False)
        assert np.abs(want_moved - want).max() > 1e-3
        c.map_set_pair(case16["surf_map"], case16["corner_map"])
        c.gn_solve_begin(case16["p0"], 4)                            # frame 0 on the original maps
        got = []
        for k in range(1, 7):
            maps = (moved_s, moved_c) if k % 2 else (case16["surf_map"], case16["corner_map"])
            c.map_set_pair_overlapped(*maps)                         # frame k's maps, while frame k - 1 is being solved
            c.gn_solve_begin(case16["p0"], 4)                        # frame k submitted before frame k - 1 is collected (bench.py's order)
            got.append(c.gn_solve_end())
        got.append(c.gn_solve_end())
        for k, pose in enumerate(got):
            assert np.array_equal(pose, want_moved if k % 2 else want), k
        info = c.map_info(mla.SURF)
        assert info["n"] == len(case16["surf_map"])
        # a map that OUTGROWS the other set's grid box while a solve is in flight: the bounds pass and the new layout run on the staging stream too
        far = np.zeros((2, case16["surf_map"].shape[1]), case16["surf_map"].dtype)
        far[0, :2] = np.abs(case16["surf_map"][:, :2]).max() + 35.0
        far[1, :2] = -far[0, :2]
        grown_s = np.concatenate([case16["surf_map"], far]); grown_c = np.concatenate([case16["corner_map"], far[:, :case16["corner_map"].shape[1]]])
        c.map_set_pair(grown_s, grown_c)
        want_grown, _ = c.gn_solve(case16["p0"], 4, want_stats=False)
        c2 = mla.Context(0)
        try:
            c2.map_set_pair(case16["surf_map"], case16["corner_map"])
            c2.features_set(mla.SURF, feats16[0]); c2.features_set(mla.CORNER, feats16[1])
            c2.gn_solve_begin(case16["p0"], 4)
            got = []
            for k in range(1, 5):
                maps = (grown_s, grown_c) if k % 2 else (case16["surf_map"], case16["corner_map"])
                c2.map_set_pair_overlapped(*maps)
                c2.gn_solve_begin(case16["p0"], 4)
                got.append(c2.gn_solve_end())
            got.append(c2.gn_solve_end())
            for k, pose in enumerate(got):
                assert np.array_equal(pose, want_grown if k % 2 else want), k
        finally:
            c2.close()
    finally:
        c.close()


@pytest.mark.gpu
def test_maps_staged_beside_a_frames_front_end(mla, case16, feats16):
    """mlh_map_set_pair_overlapped with NO solve in flight and device-resident clouds: the index is built on the second stream into the other map set while
    the main stream runs launches that read no map (here: a scan's upload and extraction), and the solve that follows reads exactly the maps staged for it.
    Alternates two different map pairs, with the synchronous entry points (map_set_pair, map_rebuild, scan2map) mixed in; host-resident clouds take the
    synchronous path and give the same answers."""
    import torch
    torch.cuda.init()
    scan = case16["scans"][0]
    moved_s, moved_c = case16["surf_map"].copy(), case16["corner_map"].copy()
    moved_s[:, 0] += 0.03; moved_c[:, 0] += 0.03
    dev = {False: (torch.from_numpy(np.ascontiguousarray(case16["surf_map"], np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(case16["corner_map"], np.float32)).cuda()),
           True: (torch.from_numpy(np.ascontiguousarray(moved_s, np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(moved_c, np.float32)).cuda())}
    host = {False: (case16["surf_map"], case16["corner_map"]), True: (moved_s, moved_c)}
    opts = mla.default_opts()
    c = mla.Context(0)
    try:
        c.features_set(mla.SURF, feats16[0]); c.features_set(mla.CORNER, feats16[1])
        want = {}
        for moved in (False, True):
            c.map_set_pair(*host[moved])
            want[moved], _ = c.scan2map(case16["p0"], opts, want_stats=False)
        assert np.abs(want[True] - want[False]).max() > 1e-3
        for k in range(8):
            moved = bool(k % 2) if k < 6 else True                   # (the last two stage the SAME maps twice in a row: both sets then hold them)
            c.scan_upload(scan.points, scan.scan_start, scan.scan_end); c.extract_run()      # something map-free in flight on the main stream
            c.map_set_pair_overlapped(*dev[moved])
            if k == 3: c.map_rebuild(mla.ALL_KINDS)                  # re-indexing the set that has just become current changes nothing
            pose, _ = c.scan2map(case16["p0"], opts, want_stats=False)
            assert np.array_equal(pose, want[moved]), k
            if k == 4:                                               # a synchronous staging in between lands in the current set; the next overlapped one in the other
                c.map_set_pair(*host[False])
                pose, _ = c.scan2map(case16["p0"], opts, want_stats=False)
                assert np.array_equal(pose, want[False])
        c.map_set_pair_overlapped(*host[False])                      # host-resident, no solve in flight: the synchronous path
        pose, _ = c.scan2map(case16["p0"], opts, want_stats=False)
        assert np.array_equal(pose, want[False])
        assert c.map_info(mla.SURF)["n"] == len(case16["surf_map"])
    finally:
        c.close()


@pytest.mark.gpu
def test_next_frames_start_pose_is_chained_on_the_device(mla, orc, case16, feats16):
    """mlh_gn_solve_begin_chained: frame k + 1 is submitted before frame k's pose has reached the host; its start pose -- transformUpdate with frame k's result,
    transformAssociateToMap with the next odometry pose (lidar_mapper_keyframe.cpp:145-160) -- is computed where that result lives. Against the host-side
    sequence: solve, chain through the oracle (which is pinned to the reference's own lines), solve again with mlh_gn_solve."""
    rng = np.random.default_rng(11)

    def odom(scale):
        q = np.array([0.0, 0.0, 0.0, 1.0]) + rng.normal(size=4) * 0.01 * scale
        return np.concatenate([rng.normal(size=3) * scale, q / np.linalg.norm(q)])

    odoms = [odom(1.0)]
    for _ in range(3):                                               # consecutive odometry poses a few centimetres apart
        step = odom(0.02)
        odoms.append(orc.pose_chain(odoms[-1], np.array([0, 0, 0, 0, 0, 0, 1.0]), step))
    c = mla.Context(0)
    try:
        c.map_set_pair(case16["surf_map"], case16["corner_map"])
        c.features_set(mla.SURF, feats16[0]); c.features_set(mla.CORNER, feats16[1])
        with pytest.raises(mla.MlhError):
            c.gn_solve_begin_chained(odoms[0], odoms[1], 3)          # nothing to continue from
        # host-side sequence
        want = []
        pose, _ = c.gn_solve(case16["p0"], 3, want_stats=False)
        want.append(pose)
        for k in range(1, 4):
            start = orc.pose_chain(want[-1], odoms[k - 1], odoms[k])
            pose, _ = c.gn_solve(start, 3, want_stats=False)
            want.append(pose)
        # device-side chain, two solves in flight
        c.gn_solve_begin(case16["p0"], 3)
        got = []
        for k in range(1, 4):
            c.gn_solve_begin_chained(odoms[k - 1], odoms[k], 3)
            got.append(c.gn_solve_end())
        got.append(c.gn_solve_end())
        for k in range(4):
            # same arithmetic, same order: the device's f64 sqrt / division are correctly rounded, so the start poses -- and with them everything after -- are equal
            assert np.array_equal(got[k], want[k]), (k, got[k] - want[k])
        assert np.abs(want[1] - want[0]).max() > 0                   # the chain really moved the start pose
    finally:
        c.close()


@pytest.mark.gpu
def test_features_handed_from_one_context_to_another(mla, case16, feats16):
    """mlh_features_copy: the staged feature sets of an estimator-side context become those of a mapper-side context on the same GPU, device to device; the
    mapper side then solves exactly what it would have solved had the features been staged from the host. The source may restage at once."""
    a, b, ref = mla.Context(0), mla.Context(0), mla.Context(0)
    try:
        for c in (b, ref):
            c.map_set_pair(case16["surf_map"], case16["corner_map"])
        ref.features_set(mla.SURF, feats16[0]); ref.features_set(mla.CORNER, feats16[1])
        want, _ = ref.scan2map(case16["p0"], want_stats=False)
        with pytest.raises(mla.MlhError):
            b.features_copy_from(a, mla.SURF)                         # nothing staged on the source yet
        a.features_set(mla.SURF, feats16[0]); a.features_set(mla.CORNER, feats16[1])
        b.features_copy_from(a, mla.SURF); b.features_copy_from(a, mla.CORNER)
        a.features_set(mla.SURF, feats16[0][::-1].copy())             # the source moves on: the copy must not care
        got, _ = b.scan2map(case16["p0"], want_stats=False)
        assert np.array_equal(got, want)
        with pytest.raises(mla.MlhError):
            b.features_copy_from(b, mla.SURF)                         # a context cannot hand over to itself
    finally:
        for c in (a, b, ref):
            c.close()


@pytest.mark.parametrize("name", ["all_invalid", "corner_far", "surf_far", "three_surf", "one_each", "nan_sprinkled", "inf_in_surf", "one_feature_2000_times",
                                  "start_3m_off", "unnormalised_quaternion"])
def test_solver_on_degenerate_inputs(mla, orc, synth, case16, feats16, name):
    """What the reference's loop does with inputs that are not a frame: no feature matches (the pose comes back untouched, LM terminates at once), one kind missing,
    three surf features, one feature of each kind, NaN / inf coordinates among the features (never valid, never poison the sums), one feature repeated 2 000 times, a
    start 3 m / 10 deg off, a start quaternion of norm 1.7. Gauss-Newton: per-iteration counts and degeneracy verdicts equal to the oracle's, pose 1e-7, classic ==
    deferred == split submission to the bit. scan2MapOptimization: LM iteration counts and terminations equal, pose 1e-7, split == synchronous to the bit."""
    fs, fc = feats16
    p = case16["p0"]

    def far(x):
        y = x.copy(); y[:, :3] += 5000.0
        return y

    def nan_some(x, k):
        y = x.copy(); y[::k, 0] = np.nan
        return y
    if name == "all_invalid": s, c = far(fs), far(fc)
    elif name == "corner_far": s, c = fs, far(fc)
    elif name == "surf_far": s, c = far(fs), fc
    elif name == "three_surf": s, c = np.ascontiguousarray(fs[:3]), fc
    elif name == "one_each": s, c = np.ascontiguousarray(fs[:1]), np.ascontiguousarray(fc[:1])
    elif name == "nan_sprinkled": s, c = nan_some(fs, 7), nan_some(fc, 5)
    elif name == "inf_in_surf": s, c = np.where(np.arange(len(fs))[:, None] % 11 == 0, np.float32(np.inf), fs).astype(np.float32), fc
    elif name == "one_feature_2000_times": s, c = np.ascontiguousarray(np.tile(fs[100:101], (2000, 1))), np.ascontiguousarray(np.tile(fc[50:51], (500, 1)))
    elif name == "start_3m_off": s, c, p = fs, fc, synth.perturbed_pose(case16["gt"], seed=9, dt=3.0, drot_deg=10.0)
    else: s, c, p = fs, fc, np.concatenate([p[:3], p[3:] * 1.7])
    ms, mc = orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"])

    def perr(a, b):
        return max(float(np.linalg.norm(a[:3] - b[:3])), 2 * min(float(np.linalg.norm(a[3:] - b[3:])), float(np.linalg.norm(a[3:] + b[3:]))))
    ctx = mla.Context(0)
    try:
        ctx.map_set_pair(case16["surf_map"], case16["corner_map"])
        ctx.features_set(mla.SURF, s); ctx.features_set(mla.CORNER, c)
        ref = orc.gn_iterations(ms, mc, s, c, p, orc.mapper_params(), 4)
        ctx.set_gn_schedule(0, 0, 0)
        pose_c, st = ctx.gn_solve(p, 4)
        ctx.set_gn_schedule(1, 1, 1)
        pose_f = ctx.gn_solve(p, 4, want_stats=False)[0]
        ctx.gn_solve_begin(p, 4)
        pose_s = ctx.gn_solve_end()
        for a, b in zip(st, ref["iters"]):
            assert (a["n_surf"], a["n_corner"], a["is_degenerate"]) == (b["n_surf"], b["n_corner"], b["is_degenerate"])
        assert np.isfinite(pose_c).all() and perr(pose_c, ref["pose"]) < 1e-7
        assert np.array_equal(pose_c, pose_f) and np.array_equal(pose_c, pose_s)
        if name in ("all_invalid", "one_each"):
            assert np.array_equal(pose_c, p)                      # nothing to solve: the pose is handed back untouched
        ref2 = orc.scan2map(ms, mc, s, c, p, orc.mapper_params())
        pose2, st2 = ctx.scan2map(p)
        ctx.scan2map_begin(p)
        pose3, status = ctx.scan2map_end()[:2]
        for a, b in zip(st2, ref2["outer"]):
            assert (a["lm_iterations"], a["termination"]) == (b["lm_iterations"], b["termination"])
        assert perr(pose2, ref2["pose"]) < 1e-7 and status in (0, 2) and np.array_equal(pose2, pose3)
    finally:
        ctx.close()


def test_zero_arguments_with_a_live_context():
    """scripts/fuzz_zero_args.py: every context-taking entry point called with a valid context (staged, and fresh) and zero / null for everything else returns -- an
    error, or success where zero is a legal value -- and the context solves to the same bits afterwards. A child process: a crash would name its function."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_zero_args.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "context still solves to the same bits: True" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-1500:])


def test_launch_checks_name_the_failing_kernel(tmp_path):
    """MLH_CHECK_LAUNCH=1: hipGetLastError() behind every launch (a bad launch configuration used to surface at the next synchronisation under another
    call's name). In a child process with the switch on, a deliberately impossible launch is reported by the call that made it, with the kernel's name, and an
    ordinary solve (a few dozen launches, every one checked) is unaffected and gives the bits of an unchecked run."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import importlib, sys, numpy as np
        sys.path.insert(0, %r)
        mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
        sc = synth.make_scene(seed=42, **synth.SCENE_PRESETS["50k"])
        surf_map, corner_map = synth.sample_maps(sc, kf_rings=16, kf_lidars=1)
        gt = synth.gt_body_pose()
        scan = synth.simulate_scan(sc, gt, synth.HERCULES_BODY_T_LASER[0], 16)
        c = mla.Context(0)
        ex = c.extract(scan.points, scan.scan_start, scan.scan_end)
        c.map_set(mla.SURF, surf_map); c.map_set(mla.CORNER, corner_map)
        c.features_set(mla.SURF, synth.voxel_mean(scan.points[ex["less_flat_raw"]], 0.4)); c.features_set(mla.CORNER, synth.voxel_mean(scan.points[ex["less_sharp"]], 0.2))
        pose, _ = c.gn_solve(synth.perturbed_pose(gt), 3, want_stats=False)
        print("POSE", pose.tobytes().hex())
        try:
            c.debug_bad_launch()
            print("BAD_LAUNCH returned")
        except mla.MlhError as e:
            print("BAD_LAUNCH raised:", e)
        pose2, _ = c.gn_solve(synth.perturbed_pose(gt), 3, want_stats=False)
        print("POSE2", pose2.tobytes().hex())
    ''') % ROOT
    out = {}
    for flag in ("1", "0"):
        env = dict(os.environ, MLH_CHECK_LAUNCH=flag)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out[flag] = r.stdout
    assert "BAD_LAUNCH raised:" in out["1"] and "debug_noop_kernel" in out["1"], out["1"]
    assert "BAD_LAUNCH returned" in out["0"], out["0"]
    poses = [line.split()[1] for flag in ("1", "0") for line in out[flag].splitlines() if line.startswith("POSE")]
    assert len(poses) == 4 and len(set(poses)) == 1, poses           # checked and unchecked, before and after the provoked error: the same bits


def test_pageable_scan_buffer_is_the_callers_again_when_upload_returns(mla, synth, case16, feats16):
    """mlh_scan_upload of a PAGEABLE host buffer relies on hipMemcpyAsync having consumed the source when it returns (capi.hip, mlh_scan_upload). Pinned here: with ~5 ms
    of kernels queued AHEAD of the copy on the context's stream (a 200-iteration solve submitted with mlh_gn_solve_begin), the caller overwrites its buffer the
    moment the call returns; the extraction that follows must see the ORIGINAL points. If the runtime ever defers reading pageable sources, this fails and the
    pinned staging path (MLH_SCAN_STAGE_PINNED=1) has to become the default."""
    sc = case16["scans"][0]
    c = mla.Context(0)
    try:
        want = c.extract(sc.points, sc.scan_start, sc.scan_end)["label"].copy()
        c.map_set(mla.SURF, case16["surf_map"]); c.map_set(mla.CORNER, case16["corner_map"])
        big_s = np.ascontiguousarray(np.tile(feats16[0], (12, 1))); big_c = np.ascontiguousarray(np.tile(feats16[1], (12, 1)))
        c.features_set(mla.SURF, big_s); c.features_set(mla.CORNER, big_c)
        for trial in range(6):
            pts = np.ascontiguousarray(sc.points, np.float32).copy()          # an ordinary (pageable) numpy buffer of the caller's
            c.gn_solve_begin(case16["p0"], 200)                               # ~5 ms of launches in front of the upload
            c.scan_upload(pts, sc.scan_start, sc.scan_end)
            pts[:] = np.float32(777.0 + trial)                                # the caller reuses its buffer at once
            c.extract_run()
            got = c.extract_fetch()["label"]
            c.gn_solve_end()
            assert np.array_equal(got, want), f"trial {trial}: the upload read the caller's buffer after mlh_scan_upload had returned"
    finally:
        c.close()


@pytest.mark.parametrize("n_rings", [16, 64])
def test_rough_scans_extract_thin_and_match(mla, synth, orc, case16, n_rings):
    """Scans with a real sensor's artefacts (synth.roughen_scan: azimuth SECTORS missing, near-range returns below a metre, rings with fewer than twelve points, an
    empty ring -- VERDICT r04: every test cloud so far was a clean ray-cast): extractCloud's labels / lists / per-ring voxel centroids are the oracle's bits, and
    the correspondences of the features they give (validity, coefficient bits) too."""
    sc = case16["scene"]
    gt = case16["gt"]
    c = mla.Context(0)
    try:
        c.map_set(mla.SURF, case16["surf_map"]); c.map_set(mla.CORNER, case16["corner_map"])
        ms, mc = orc.Map(case16["surf_map"]), orc.Map(case16["corner_map"])
        for seed in range(4):
            clean = synth.simulate_scan(sc, gt, synth.HERCULES_BODY_T_LASER[seed % 2], n_rings, seed=20 + seed)
            scan = synth.roughen_scan(clean, seed=seed, n_sectors=2 + seed % 3, near_fraction=0.005 * (1 + seed), short_rings=1 + seed % 3)
            ref = orc.extract(scan.points, scan.scan_start, scan.scan_end)
            got = c.extract(scan.points, scan.scan_start, scan.scan_end, voxel_leaf=0.2)
            assert np.array_equal(got["curvature"].view(np.uint32), ref["curvature"].view(np.uint32))
            assert np.array_equal(got["label"], ref["label"]) and np.array_equal(got["picked"], ref["picked"])
            for k in ("sharp", "less_sharp", "flat", "less_flat_raw"):
                assert np.array_equal(got[k], ref[k]), (seed, k)
            assert np.array_equal(got["less_flat_ds"].view(np.uint32), ref["less_flat_ds"].view(np.uint32))
            # the features the mapper would take from this scan (the near-range points among them), matched at a perturbed pose
            surf = np.ascontiguousarray(ref["less_flat_ds"], np.float32); corner = np.ascontiguousarray(scan.points[ref["less_sharp"]], np.float32)
            if seed % 2:           # the second LiDAR's points, moved into the body frame as the fusion does
                T = np.eye(4); T[:3, :3] = synth.quat_to_rot(synth.HERCULES_BODY_T_LASER[1][:4]); T[:3, 3] = synth.HERCULES_BODY_T_LASER[1][4:7]
                surf[:, :3] = synth.transform_points(surf[:, :3], T); corner[:, :3] = synth.transform_points(corner[:, :3], T)
            p0 = synth.perturbed_pose(gt, seed=50 + seed)
            for kind, feats, m_, name in ((mla.SURF, surf, ms, "s"), (mla.CORNER, corner, mc, "c")):
                c.features_set(kind, feats)
                r = c.match_linearize(kind, p0, dense=False)
                v, co = m_.match(name, feats, p0)
                assert np.array_equal(r["valid"], v), (seed, name)
                nc = 4 if name == "s" else 6
                assert np.array_equal(r["coeffs"][v.astype(bool), :nc].astype(np.float32).view(np.uint32), np.asarray(co)[v.astype(bool), :nc].astype(np.float32).view(np.uint32)), (seed, name)
    finally:
        c.close()


@pytest.mark.gpu
def test_lm_loop_kernel_with_a_missing_workgroup_neither_hangs_nor_loses_the_frame(mla, case16, feats16, monkeypatch):
    """scan2map's LM loop runs as one launch whose workgroups meet at a counter barrier once per iteration. A workgroup that never arrives (MLH_DEBUG_LOOP_STALL=1 makes
    one skip its second arrival) must not hang the stream: it says so in the barrier's "given up" word, the others leave at their next look, the launch publishes the
    failure bit -- and the call solves the frame again through the launch-per-iteration form: the same pose, a counted fallback (mlh_get_info), a context that goes on
    working (the barrier's counters are re-armed by the last workgroup to leave). tests/test_gpu_residency.py holds the same at BASELINE config 2's size."""
    import time
    c = mla.Context(0)
    try:
        c.map_set(mla.SURF, case16["surf_map"]); c.map_set(mla.CORNER, case16["corner_map"])
        c.features_set(mla.SURF, feats16[0]); c.features_set(mla.CORNER, feats16[1])
        ref = c.scan2map(case16["p0"], want_stats=False)[0]
        monkeypatch.setenv("MLH_DEBUG_LOOP_STALL", "1")
        t0 = time.time()
        got = c.scan2map(case16["p0"], want_stats=False)[0]
        assert time.time() - t0 < 5.0
        assert np.array_equal(got, ref)
        di = c.info()
        assert di["loop_timeouts"] == 1 and di["loop_fallbacks"] == 1
        monkeypatch.delenv("MLH_DEBUG_LOOP_STALL")
        again = c.scan2map(case16["p0"], want_stats=False)[0]
        assert np.array_equal(again, ref)
        c.scan2map_begin(case16["p0"])
        pose, status = c.scan2map_end()
        assert status == 0 and np.array_equal(pose, ref)
    finally:
        c.close()
