"""The one-launch Levenberg-Marquardt loops (match.hip: lm_loop_kernel, track.hip: track_lm_loop_kernel) synchronise their workgroups inside the launch, so every
workgroup must be resident at once. The host gates them on what the device -- and the part of it the solver's stream may use -- admits (mlh_get_info: occupancy query x
compute units, asked at mlh_create), takes the launch-per-iteration forms beyond the gate, and solves a frame whose barrier was nevertheless given up on again through
those forms inside the same call. Held here, at BASELINE config 2's size (83 tiles): the same pose bits under 16- and 32-CU solver streams with the loop kernel chosen
only where it fits; four contexts in four threads running whole frames at once without a single barrier given up on; the fused thinning + solve call at its tile bound;
a barrier that IS given up on (debug hook; a too-short time limit) costing a slow frame, not a lost one.
The call these launches stand for: ceres::Solve inside scan2MapOptimization, lidar_mapper_keyframe.cpp:586-596."""
import os
import threading
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


@pytest.fixture(scope="module")
def cfg2(synth, orc):
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
    ex = [orc.extract(s.points, s.scan_start, s.scan_end) for s in scans]
    surf, corner = bench.fuse_features(synth, scans, ex)
    ext = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:len(scans)]
    for e in ext:
        e[3:] /= np.linalg.norm(e[3:])
    covs = np.stack([np.zeros((6, 6))] + [np.diag([0.0025] * 3 + [0.00030461] * 3)] * (len(scans) - 1))
    return dict(surf_map=surf_map, corner_map=corner_map, scans=scans, surf=surf, corner=corner, p0=synth.perturbed_pose(gt, seed=43), ext=ext, covs=covs,
                meas=np.diag([0.0025] * 3))


def _stage(c, mla, cfg2):
    c.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
    c.features_set(mla.SURF, cfg2["surf"])
    c.features_set(mla.CORNER, cfg2["corner"])
    return (len(cfg2["surf"]) + 255) // 256 + (len(cfg2["corner"]) + 255) // 256


def _mask_words(n_cu):
    words, left = [], n_cu
    while left > 0:
        words.append("%x" % ((1 << min(left, 32)) - 1))
        left -= 32
    return ",".join(words)


def _gate(occ, n_cu, cap):
    """capi.hip: set_loop_gates, for a context without a masked staging stream"""
    return min(cap, max(0, min(occ, 6) * n_cu - max(n_cu // 8, 8)))


def test_device_info_is_asked_not_assumed(mla):
    c = mla.Context(0)
    try:
        di = c.info()
        assert di["cu_count"] > 0 and di["cu_solver"] == di["cu_count"]
        for occ, gate, cap in zip(di["loop_blocks_per_cu"], di["loop_max_tiles"], (160, 512, 160)):
            assert 1 <= occ <= 16
            assert gate == _gate(occ, di["cu_count"], cap)
        assert di["loop_launches"] == di["loop_timeouts"] == di["loop_fallbacks"] == 0
    finally:
        c.close()


@pytest.mark.parametrize("n_cu", [16, 32, 64, 96])
def test_cu_masked_solver_stream_same_pose_bits_and_the_loop_only_where_it_fits(mla, cfg2, n_cu):
    """scan2map (synchronous and split submission) and the 5-GN solve on a solver stream confined to n_cu compute units: the poses of the whole device, bit for bit; the
    one-launch loop is taken exactly when the frame's tiles fit the gate computed for n_cu compute units; no barrier is ever given up on."""
    full = mla.Context(0)
    try:
        tiles = _stage(full, mla, cfg2)
        want = full.scan2map(cfg2["p0"], want_stats=False)[0]
        want_gn = full.gn_solve(cfg2["p0"], 5, want_stats=False)[0]
        di_full = full.info()
        assert di_full["loop_launches"] == 1 and di_full["loop_timeouts"] == 0
    finally:
        full.close()
    c = _ctx_with_env(mla, MLH_SOLVER_CU_MASK=_mask_words(n_cu))
    try:
        di = c.info()
        assert di["cu_solver"] == n_cu and di["cu_count"] == di_full["cu_count"]
        gate = di["loop_max_tiles"][0]
        assert gate == _gate(di["loop_blocks_per_cu"][0], n_cu, 160)
        assert _stage(c, mla, cfg2) == tiles
        for _ in range(20):
            got = c.scan2map(cfg2["p0"], want_stats=False)[0]
            assert np.array_equal(got, want)
        for _ in range(5):
            c.scan2map_begin(cfg2["p0"])
            pose, status = c.scan2map_end()
            assert status == 0 and np.array_equal(pose, want)
        assert np.array_equal(c.gn_solve(cfg2["p0"], 5, want_stats=False)[0], want_gn)
        di = c.info()
        assert di["loop_timeouts"] == 0 and di["loop_fallbacks"] == 0
        assert di["loop_launches"] == (25 if tiles <= gate else 0), (tiles, gate, di)
    finally:
        c.close()


def test_gate_by_hand_and_disabled(mla, cfg2):
    """MLH_LOOP_MAX_TILES lowers the gates (0: never): at exactly the frame's tile count the loop launch is taken, one below it is not; the pose bits do not move."""
    ref = mla.Context(0)
    try:
        tiles = _stage(ref, mla, cfg2)
        want = ref.scan2map(cfg2["p0"], want_stats=False)[0]
    finally:
        ref.close()
    for limit, expect_loop in ((tiles, True), (tiles - 1, False), (0, False)):
        c = _ctx_with_env(mla, MLH_LOOP_MAX_TILES=limit)
        try:
            _stage(c, mla, cfg2)
            assert c.info()["loop_max_tiles"][0] == limit
            assert np.array_equal(c.scan2map(cfg2["p0"], want_stats=False)[0], want)
            assert c.info()["loop_launches"] == (1 if expect_loop else 0)
        finally:
            c.close()


def _frame(c, mla, cfg2, opts, fused_call):
    c.fuse_reset()
    for i, s in enumerate(cfg2["scans"]):
        c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run(); c.extract_voxel_run(0.2)
        c.fuse_add_scan(i, cfg2["ext"][i])
    if fused_call:
        return c.downsample_scan2map(c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER), 0.4, 0.2, cfg2["ext"], cfg2["covs"], cfg2["meas"], cfg2["p0"], opts)
    cnt = c.downsample_current_scan_pair(c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER), 0.4, 0.2, cfg2["ext"], cfg2["covs"], cfg2["meas"], True, 0.6)
    return c.scan2map(cfg2["p0"], opts, want_stats=False)[0], cnt


def test_fused_thinning_plus_solve_call_at_its_tile_bound(mla, cfg2):
    """mlh_downsample_scan2map sizes its loop launch for the UN-thinned clouds (the bound's tiles, ~300 here). On a solver stream of just enough compute units for that
    bound the call takes its fused form (every bound tile's workgroup is launched; the ones that exist synchronise) and returns the two calls' pose and counts; one
    compute unit fewer and it takes the two calls by itself."""
    opts = mla.default_opts(flags=mla.FLAG_WITH_UA)
    ref = mla.Context(0)
    try:
        ref.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
        want, want_cnt = _frame(ref, mla, cfg2, opts, False)
        bound = (ref.fused_cloud(mla.SURF).n + 255) // 256 + (ref.fused_cloud(mla.CORNER).n + 255) // 256
        occ1 = ref.info()["loop_blocks_per_cu"][1]
        n_total = ref.info()["cu_count"]
    finally:
        ref.close()
    assert 160 < bound <= 512
    n_cu = next((n for n in range(1, n_total + 1) if _gate(occ1, n, 512) >= bound), n_total + 1)      # the fewest compute units whose gate admits the bound
    if n_cu > n_total:
        pytest.skip("the device has fewer compute units than the bound needs")
    for cus, fused_form in ((n_cu, True), (n_cu - 1, False)):
        c = _ctx_with_env(mla, MLH_SOLVER_CU_MASK=_mask_words(cus))
        try:
            c.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
            assert (c.info()["loop_max_tiles"][1] >= bound) == fused_form
            for _ in range(10):
                got, cnt = _frame(c, mla, cfg2, opts, True)
                assert np.array_equal(got, want) and tuple(cnt) == tuple(want_cnt)
            di = c.info()
            assert di["loop_timeouts"] == 0 and di["loop_fallbacks"] == 0
            assert di["loop_launches"] == (10 if fused_form or (len(cfg2["surf"]) + 255) // 256 + (len(cfg2["corner"]) + 255) // 256 <= di["loop_max_tiles"][0] else 0)
        finally:
            c.close()


def test_four_contexts_in_four_threads_whole_frames_no_barrier_given_up(mla, cfg2):
    """The facade's shape: one context per thread, all on one device, each running whole frames (upload, extract, fuse, thinning -- 1 024-thread sort workgroups that
    fill every wave slot for tens of microseconds -- and the solve with its in-kernel barrier) at the same time. Every frame of every thread returns the single
    context's pose bits, and no barrier is given up on (MLOAM_RESIDENCY_FRAMES=2000 for the soak: profiles/r06_soak.txt)."""
    opts = mla.default_opts(flags=mla.FLAG_WITH_UA)
    ref = mla.Context(0)
    try:
        ref.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
        want, want_cnt = _frame(ref, mla, cfg2, opts, False)
    finally:
        ref.close()
    n_frames = int(os.environ.get("MLOAM_RESIDENCY_FRAMES", "60"))
    errors, infos = [], []

    def work(tid):
        try:
            c = mla.Context(0)
            try:
                c.map_set_pair(cfg2["surf_map"], cfg2["corner_map"])
                for f in range(n_frames):
                    got, cnt = _frame(c, mla, cfg2, opts, fused_call=(tid + f) % 2 == 1)
                    if not (np.array_equal(got, want) and tuple(cnt) == tuple(want_cnt)):
                        errors.append((tid, f, "pose / counts differ"))
                        return
                infos.append(c.info())
            finally:
                c.close()
        except Exception as ex:      # noqa: BLE001 -- reported below
            errors.append((tid, -1, repr(ex)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(infos) == 4
    for di in infos:
        assert di["loop_launches"] == n_frames
        assert di["loop_timeouts"] == 0 and di["loop_fallbacks"] == 0, di


@pytest.mark.parametrize("loop_form", ["tagged_records", "grid_barrier"])
def test_a_barrier_given_up_on_is_a_slow_frame_not_a_lost_one(mla, cfg2, monkeypatch, loop_form):
    """(Both forms of the one-launch loop's exchange: tagged records summed by polling -- a record that never arrives --, and records behind a grid barrier -- an
    arrival that never comes; MLH_LOOP_TAGGED.) MLH_DEBUG_LOOP_STALL=1 makes one workgroup skip an arrival: the launch publishes the failure, the call solves the frame again through the launch-per-iteration
    form and returns the SAME pose; the context counts it, halves its gate (the next frames take the other form by themselves) and goes on working. The same through
    the split submission (status 2: solved again at collection) and for a frame chained behind the failed one (status 3: began from a pose that was not a result)."""
    import time
    monkeypatch.setenv("MLH_LOOP_TAGGED", "1" if loop_form == "tagged_records" else "0")
    c = mla.Context(0)
    try:
        tiles = _stage(c, mla, cfg2)
        want = c.scan2map(cfg2["p0"], want_stats=False)[0]
        gate0 = c.info()["loop_max_tiles"][0]
        monkeypatch.setenv("MLH_DEBUG_LOOP_STALL", "1")
        t0 = time.time()
        got = c.scan2map(cfg2["p0"], want_stats=False)[0]
        assert time.time() - t0 < 5.0
        assert np.array_equal(got, want)
        di = c.info()
        assert di["loop_timeouts"] == 1 and di["loop_fallbacks"] == 1
        assert di["loop_max_tiles"][0] == min(gate0, tiles) // 2 < tiles
        launches = di["loop_launches"]
        assert np.array_equal(c.scan2map(cfg2["p0"], want_stats=False)[0], want)        # (the gate is below the frame now: no loop launch, nothing to stall)
        assert c.info()["loop_launches"] == launches and c.info()["loop_timeouts"] == 1
    finally:
        c.close()
    c = mla.Context(0)
    try:
        _stage(c, mla, cfg2)
        monkeypatch.setenv("MLH_DEBUG_LOOP_STALL", "1")
        c.scan2map_begin(cfg2["p0"])
        pose, status = c.scan2map_end()
        assert status == 2 and np.array_equal(pose, want)
        di = c.info()
        assert di["loop_timeouts"] == 1 and di["loop_fallbacks"] == 1
        monkeypatch.delenv("MLH_DEBUG_LOOP_STALL")
        c.scan2map_begin(cfg2["p0"])
        pose, status = c.scan2map_end()
        assert status == 0 and np.array_equal(pose, want)
    finally:
        c.close()
    # a frame chained behind the failed one: reported as such (status 3), and the failed one -- which cannot be solved again with a successor in flight -- as status 1
    c = mla.Context(0)
    try:
        _stage(c, mla, cfg2)
        monkeypatch.setenv("MLH_DEBUG_LOOP_STALL", "1")
        c.scan2map_begin(cfg2["p0"])
        monkeypatch.delenv("MLH_DEBUG_LOOP_STALL")
        ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
        c.scan2map_begin_chained(ident, ident)
        pose1, status1 = c.scan2map_end()
        pose2, status2 = c.scan2map_end()
        assert status1 == 1 and status2 == 3
        assert np.array_equal(pose1, cfg2["p0"])
        c.scan2map_begin(cfg2["p0"])
        pose, status = c.scan2map_end()
        assert status == 0 and np.array_equal(pose, want)
    finally:
        c.close()


def test_a_time_limit_shorter_than_a_barrier_still_returns_the_pose(mla, cfg2):
    """MLH_LOOP_TIMEOUT_US=1 with another context's kernels in the way: whether or not a wait outlasts one microsecond, every call returns the pose bits."""
    ref = mla.Context(0)
    try:
        _stage(ref, mla, cfg2)
        want = ref.scan2map(cfg2["p0"], want_stats=False)[0]
    finally:
        ref.close()
    c = _ctx_with_env(mla, MLH_LOOP_TIMEOUT_US=1)
    try:
        _stage(c, mla, cfg2)
        for _ in range(10):
            assert np.array_equal(c.scan2map(cfg2["p0"], want_stats=False)[0], want)
        di = c.info()
        assert di["loop_fallbacks"] == di["loop_timeouts"] <= 10
    finally:
        c.close()


def test_tracker_loop_gate_and_fallback(mla, track_case, monkeypatch):
    """trackCloud's rounds (lidar_tracker.cpp:42-121) behind the same gate: a solver stream of 4 compute units (gate below the frame's tiles) and MLH_LOOP_MAX_TILES=0
    return the whole device's pose bits through the launch-per-iteration rounds."""
    tc = track_case
    p0 = np.array([0, 0, 0, 0, 0, 0, 1.0])

    def run(c):
        c.track_set_prev(mla.CORNER, tc["corner_last"]); c.track_set_prev(mla.SURF, tc["surf_last"])
        c.track_set_cur(mla.CORNER, tc["corner_sharp"]); c.track_set_cur(mla.SURF, tc["surf_flat"])
        return c.track_cloud(p0, want_stats=False)[0]
    c = mla.Context(0)
    try:
        want = run(c)
        assert c.info()["loop_launches"] == 2 and c.info()["loop_timeouts"] == 0       # (two rounds, a launch each)
    finally:
        c.close()
    for env in (dict(MLH_LOOP_MAX_TILES=0), dict(MLH_SOLVER_CU_MASK="1")):
        c = _ctx_with_env(mla, **env)
        try:
            assert np.array_equal(run(c), want)
            di = c.info()
            assert di["loop_timeouts"] == 0
            if "MLH_LOOP_MAX_TILES" in env:
                assert di["loop_launches"] == 0
        finally:
            c.close()
