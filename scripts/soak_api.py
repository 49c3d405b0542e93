"""History-independence soak of the C-ABI (not part of the test suite): every job below is a deterministic function of its inputs; its expected output is taken
ONCE from a fresh context, then a long-lived context runs the jobs in random order -- maps of different sizes staged five ways (or found staged and reused),
feature sets of different sizes (host / device memory, pose blocks), solves synchronous / split / split with the maps re-staged beside them, the front end
(extractCloud, segmentCloud, fusion, thinning, tracking) in between -- and every output must be the expected bits. What this finds: stale state, buffers that grow
or are reused wrongly, launches that are not ordered behind their producers.
usage: python scripts/soak_api.py [seconds] [seed] [threads]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest, oracle as O
import torch
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O.build()
ext_fn = lambda s: O.extract(s.points, s.scan_start, s.scan_end)
caseA = conftest._make_case(synth, "50k", 16, 1)
caseB = conftest._make_case(synth, "50k", 16, 2, seed=5)
featA = conftest.features_from_extraction(synth, caseA["scans"], ext_fn)
featB = conftest.features_from_extraction(synth, caseB["scans"], ext_fn)
MAPS = {"A": (caseA["surf_map"], caseA["corner_map"]), "B": (caseB["surf_map"], caseB["corner_map"]),
        "C": (np.ascontiguousarray(caseA["surf_map"][::5]), np.ascontiguousarray(caseA["corner_map"][::3]))}
MAPS_DEV = {k: (torch.from_numpy(v[0]).cuda(), torch.from_numpy(v[1]).cuda()) for k, v in MAPS.items()}
FEATS = {"A": featA, "B": featB, "A5": (np.ascontiguousarray(np.tile(featA[0], (5, 1))), np.ascontiguousarray(np.tile(featA[1], (5, 1)))),
         "As": (np.ascontiguousarray(featA[0][:300]), np.ascontiguousarray(featA[1][:90]))}
FEATS_DEV = {k: (torch.from_numpy(v[0]).cuda(), torch.from_numpy(v[1]).cuda()) for k, v in FEATS.items()}
P0 = {"A": caseA["p0"], "B": caseB["p0"], "C": caseA["p0"]}
SCANS = [caseA["scans"][0], caseB["scans"][0], caseB["scans"][1]]
GN_FEATS = ["A", "B", "A5", "As"]
if os.environ.get("SOAK_BIG"):
    # BASELINE config 2's sizes beside the small ones: the 500 k map, its thinned (21 k) and un-thinned (230 k: more fit tiles than the deferred finish takes)
    # feature sets, two 64-ring scans -- buffers grow and shrink by two orders of magnitude between jobs
    import bench
    scL, surfL, cornerL, gtL, scansL = bench.build_workload(synth, "500k")
    exL = [O.extract(s_.points, s_.scan_start, s_.scan_end) for s_ in scansL]
    MAPS["L"] = (surfL, cornerL)
    MAPS_DEV["L"] = (torch.from_numpy(surfL).cuda(), torch.from_numpy(cornerL).cuda())
    for nm, thin in (("L", True), ("Ld", False)):
        FEATS[nm] = bench.fuse_features(synth, scansL, exL, thin=thin)
        FEATS_DEV[nm] = (torch.from_numpy(FEATS[nm][0]).cuda(), torch.from_numpy(FEATS[nm][1]).cuda())
    P0["L"] = synth.perturbed_pose(gtL, seed=43)
    SCANS += list(scansL)
    GN_FEATS += ["L", "Ld"]
sc32 = synth.simulate_scan(caseA["scene"], caseA["gt"], synth.HERCULES_BODY_T_LASER[0], 32, seed=21, n_cols=900)
SCANS.append(sc32)
track = conftest._track_case(synth, O)
ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
EXT = np.array([np.concatenate([r[4:7], r[:4]]) for r in synth.HERCULES_BODY_T_LASER])[:2]
for e in EXT:
    e[3:] /= np.linalg.norm(e[3:])
COVS = np.stack([np.zeros((6, 6)), np.diag([0.0025] * 3 + [0.00030461] * 3)])
MEAS = np.diag([0.0025] * 3)
vg_pts = [np.ascontiguousarray(s.points[:, :4]) for s in SCANS[:2]]
seg_pts = []
for s in (SCANS[0], SCANS[3]):
    p = np.ascontiguousarray(s.points[:, :4]).copy()
    seg_pts.append(np.ascontiguousarray(p[np.random.default_rng(3).permutation(len(p))]))


class St:
    map = None
    feat = None


def ensure_map(c, st, m, r, force=False):
    how = "reuse"
    if st.map != m or force or r.random() < 0.25:
        how = ["pair_host", "two_host", "pair_dev", "pair_overlapped_dev", "pair_host"][int(r.integers(5))]
        s_, c_ = MAPS[m]
        if how == "pair_host":
            c.map_set_pair(s_, c_)
        elif how == "two_host":
            c.map_set(mla.SURF, s_); c.map_set(mla.CORNER, c_)
        elif how == "pair_dev":
            c.map_set_pair(*MAPS_DEV[m])
        else:
            c.map_set_pair_overlapped(*MAPS_DEV[m])
        st.map = m
    elif r.random() < 0.3:
        how = "rebuild"
        c.map_rebuild(mla.ALL_KINDS)
    return how


def ensure_feat(c, st, f, r):
    how = "reuse"
    if st.feat != f or r.random() < 0.25:
        how = "host" if r.random() < 0.5 else "dev"
        src = FEATS[f] if how == "host" else FEATS_DEV[f]
        c.features_set(mla.SURF, src[0]); c.features_set(mla.CORNER, src[1])
        st.feat = f
    return how


def j_extract(c, st, r, i):
    s = SCANS[i]
    # mlh_scan_upload_ahead in every position a caller can put it: this scan sent ahead (the upload packs from it), another scan sent ahead (dropped by this upload),
    # a look-ahead left behind for whatever job comes next (its upload -- of this scan, of another, through the frame jobs -- finds it)
    how = int(r.integers(4))
    if how == 1:
        c.scan_upload_ahead(s.points)
    elif how == 2:
        c.scan_upload_ahead(SCANS[(i + 1) % len(SCANS)].points)
    ex = c.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
    if how == 3:
        c.scan_upload_ahead(SCANS[int(r.integers(len(SCANS)))].points)
    return [ex[k] for k in ("label", "picked", "sharp", "less_sharp", "flat", "less_flat_raw", "less_flat_ds")], ("", "ahead", "ahead-other", "ahead-left")[how]


def j_knn(c, st, r, m, kind):
    how = ensure_map(c, st, m, r)
    q = FEATS["L" if m == "L" else "A"][kind][:2000]
    idx, d2 = c.knn(kind, q)
    # the raw query is exact inside the acceptance radius only (include/mloam_hip.h: mlh_knn); beyond it the answer depends on where the grid box of this map set
    # happens to start, i.e. on the maps staged before -- which is history, and allowed
    far = d2 >= 1.0
    idx, d2 = idx.copy(), d2.copy()
    idx[far] = -1; d2[far] = np.inf
    return [idx, d2], how


def j_gn(c, st, r, m, f, n, variant):
    how = ensure_map(c, st, m, r) + "/" + ensure_feat(c, st, f, r)
    if variant == "sync":
        return [c.gn_solve(P0[m], n, want_stats=False)[0]], how
    if variant == "stats":
        return [c.gn_solve(P0[m], n, want_stats=True)[0]], how
    c.gn_solve_begin(P0[m], n)
    if variant == "split_restage":
        c.map_set_pair_overlapped(*MAPS_DEV[m])
        if r.random() < 0.5:
            c.map_set_pair_overlapped(*MAPS_DEV[m])
    return [c.gn_solve_end()], how


def j_s2m(c, st, r, m, f, variant):
    how = ensure_map(c, st, m, r) + "/" + ensure_feat(c, st, f, r)
    if variant == "sync":
        return [c.scan2map(P0[m], want_stats=False)[0]], how
    c.scan2map_begin(P0[m], lm_lookahead=12)
    if variant == "split_restage":
        c.map_set_pair_overlapped(*MAPS_DEV[m])
    pose, status = c.scan2map_end()[:2]
    if status == 1:        # the LM loop outran the look-ahead and the maps were restaged beside it: the contract hands the frame back (include/mloam_hip.h)
        pose = c.scan2map(P0[m], want_stats=False)[0]
    return [pose], how + f"/status{status}"


def j_ml(c, st, r, m, f, kind):
    how = ensure_map(c, st, m, r) + "/" + ensure_feat(c, st, f, r)
    o = c.match_linearize(kind, P0[m], dense=True)
    return [o["valid"], o["coeffs"], o["r"], o["J"], o["H"], o["g"], np.array([o["cost"], o["count"]])], how


def j_gf(c, st, r, m, f, kind, method):
    how = ensure_map(c, st, m, r) + "/" + ensure_feat(c, st, f, r)
    o = c.good_feature_matching(kind, P0[m], gf_method=method, gf_ratio=0.2, seed=3)
    return [o["sel"], o["H"], o["matched"]], how


def j_s2m_gf(c, st, r, m, f, method):
    # scan2MapOptimization behind a good-feature selection: 'fps' runs the two kinds' farthest-point loops side by side in one launch (select.hip)
    how = ensure_map(c, st, m, r) + "/" + ensure_feat(c, st, f, r)
    pose, stt = c.scan2map(P0[m], mla.default_opts(gf_method=mla.GF_METHODS[method], gf_ratio=0.3, gf_seed=5), want_stats=True)
    return [pose, np.array([[s_["n_surf"], s_["n_corner"]] for s_ in stt])], how


def j_voxel_grid(c, st, r, i, leaf):
    return [c.voxel_grid(vg_pts[i], leaf)], ""


def j_voxel_filter(c, st, r, i, leaf):
    return [c.voxel_filter(vg_pts[i], leaf)], ""


def j_track(c, st, r):
    c.track_set_prev(mla.CORNER, track["corner_last"]); c.track_set_prev(mla.SURF, track["surf_last"])
    c.track_set_cur(mla.CORNER, track["corner_sharp"]); c.track_set_cur(mla.SURF, track["surf_flat"])
    return [c.track_cloud(ident, want_stats=False)[0]], ""


def j_segment(c, st, r, i):
    vs = 16 if i == 0 else 32
    o = c.segment_cloud(seg_pts[i], vertical_scans=vs)
    return [o["cloud"], o["outlier"], o["scan_start"], o["scan_end"]], ""


def j_downsample(c, st, r, kind):
    src = FEATS["B"][kind]
    out = c.downsample_current_scan(kind, src, 0.4 if kind == mla.SURF else 0.2, EXT, COVS, MEAS, True, 0.6)
    st.feat = None
    return [out], ""


def j_blocks(c, st, r, m, subset):
    how = ensure_map(c, st, m, r)
    fs = [FEATS["A"], FEATS["B"], FEATS["As"], FEATS["A"]]
    c.features_set_blocks(mla.SURF, [fs[b][0] for b in subset]); c.features_set_blocks(mla.CORNER, [fs[b][1] for b in subset])
    st.feat = None
    kk, th, fz = [5, 10, 10, 10], [100.0, 70.0, 70.0, 70.0], [0, 1, 1, 1]
    p = c.gn_solve_blocks(np.array([P0[m]] * len(subset)), 4, [kk[b] for b in subset], [th[b] for b in subset], [fz[b] for b in subset],
                          mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=1.0), want_stats=False)[0]
    return [p], how


def j_frame(c, st, r, m):
    """two scans -> extractCloud -> fusion -> thinning -> scan2MapOptimization, device-resident hand-overs"""
    how = ensure_map(c, st, m, r)
    c.fuse_reset()
    for i, s in enumerate((SCANS[-2], SCANS[-1]) if m == "L" else (SCANS[1], SCANS[2])):
        c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run(); c.extract_voxel_run(0.2)
        c.fuse_add_scan(i, EXT[i])
    n = c.downsample_current_scan_pair(c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER), 0.4, 0.2, EXT, COVS, MEAS, True, 0.6)
    st.feat = None
    pose = c.scan2map(P0[m], mla.default_opts(flags=mla.FLAG_WITH_UA), want_stats=False)[0]
    return [np.array(n), pose], how


def j_frame_one_call(c, st, r, m):
    """the same frame with thinning + solve as ONE call (mlh_downsample_scan2map: the thinned counts stay on the device until the pose comes back): it must leave the
    counts, the pose and the staged feature sets of the two calls (the match that follows reads those sets)"""
    how = ensure_map(c, st, m, r)
    c.fuse_reset()
    for i, s in enumerate((SCANS[-2], SCANS[-1]) if m == "L" else (SCANS[1], SCANS[2])):
        c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run(); c.extract_voxel_run(0.2)
        c.fuse_add_scan(i, EXT[i])
    pose, n = c.downsample_scan2map(c.fused_cloud(mla.SURF), c.fused_cloud(mla.CORNER), 0.4, 0.2, EXT, COVS, MEAS, P0[m], mla.default_opts(flags=mla.FLAG_WITH_UA))
    st.feat = None
    ml = c.match_linearize(mla.SURF, P0[m], flags=mla.FLAG_WITH_UA)
    return [np.array(n), pose, ml["valid"], ml["H"]], how


WIN = [conftest.make_window_case(synth, O, 1, 2), conftest.make_window_case(synth, O, 2, 2, seed=4)]


def j_window(c, st, r, i, what):
    """Estimator::optimizeMap's coupled window: factor table staged, normal equations / device Gauss-Newton"""
    w = WIN[i]
    c.pure_odom_set(w["types"], w["points"], w["coeffs"], w["fi"], w["ei"])
    if what == "ne":
        o = c.pure_odom_normal_eq(w["pivot"], w["frames"], w["exts"], huber_delta=1.0)
        return [o["H"], o["g"], np.array([o["cost"], o["count"]])], ""
    o = c.pure_odom_gn_solve(w["pivot"], w["frames"], w["exts"], n_iters=3, huber_delta=1.0)
    return [o["frames"], o["exts"], np.array([o["cost"], o["count"], o["status"]])], ""


def j_window_device_table(c, st, r, m):
    """the factor table built on the device from matches against the resident map (mlh_pure_odom_begin / _add_matches), then its normal equations"""
    how = ensure_map(c, st, m, r)
    c.pure_odom_begin()
    for fi_, (f, rel) in enumerate((("A", P0[m]), ("As", P0[m]))):
        for kind in (mla.SURF, mla.CORNER):
            c.features_set(kind, FEATS[f][kind])
            c.pure_odom_add_matches(kind, rel, 0, fi_ % 2)
    st.feat = None
    o = c.pure_odom_normal_eq(ident, np.array([ident]), np.array([ident, ident]), huber_delta=1.0)
    return [o["H"], o["g"], np.array([o["cost"], o["count"]])], how


def j_window_device_table_gf(c, st, r, m, ratio):
    """the same table behind the odometry's good-feature selection (mlh_pure_odom_add_matches_gf): selections and the normal equations of the selected factors"""
    how = ensure_map(c, st, m, r)
    c.pure_odom_begin()
    outs = []
    for kind in (mla.SURF, mla.CORNER):
        c.features_set(kind, FEATS["A"][kind])
        outs.append(c.pure_odom_add_matches_gf(kind, P0[m], ident, P0[m], ident, 0, 0, gf_ratio=ratio, seed=17 + kind))
    st.feat = None
    o = c.pure_odom_normal_eq(ident, np.array([P0[m]]), np.array([ident]), huber_delta=1.0)
    return outs + [o["H"], o["g"], np.array([o["cost"], o["count"]])], how


def j_uncertainty(c, st, r, f):
    cov, keep = c.point_uncertainty(FEATS[f][0], EXT, COVS, MEAS, 0.6)
    return [cov, keep], ""


SORT_KEYS = [np.random.default_rng(k).integers(0, 200 * (k + 1), 3000 * (k + 1)).astype(np.int32) for k in range(3)]


def j_std_sort(c, st, r, i):
    return [c.std_sort_permutation(SORT_KEYS[i])], ""


JOBS = []
for i in range(2):
    for what in ("ne", "gn"):
        JOBS.append((("window", i, what), j_window, (i, what)))
for m in ("A", "C"):
    JOBS.append((("window_dev", m), j_window_device_table, (m,)))
    for ratio in (0.8, 0.3):
        JOBS.append((("window_dev_gf", m, ratio), j_window_device_table_gf, (m, ratio)))
for f in ("A", "B"):
    JOBS.append((("uncertainty", f), j_uncertainty, (f,)))
for i in range(3):
    JOBS.append((("std_sort", i), j_std_sort, (i,)))
for i in range(len(SCANS)):
    JOBS.append((("extract", i), j_extract, (i,)))
for m in MAPS:
    for kind in (mla.SURF, mla.CORNER):
        JOBS.append((("knn", m, kind), j_knn, (m, kind)))
    for f in GN_FEATS:
        if (f in ("L", "Ld")) != (m == "L"):        # the big feature sets against the big map only (and only they against it)
            continue
        for n in (1, 2, 5):
            for variant in ("sync", "stats", "split", "split_restage"):
                JOBS.append((("gn", m, f, n), j_gn, (m, f, n, variant)))
        for variant in ("sync", "split", "split_restage"):
            JOBS.append((("s2m", m, f), j_s2m, (m, f, variant)))
    for f in (("L",) if m == "L" else ("A", "As")):
        for kind in (mla.SURF, mla.CORNER):
            JOBS.append((("ml", m, f, kind), j_ml, (m, f, kind)))
    if m != "L":
        JOBS.append((("gf", m, "A", mla.SURF, "gd_fix"), j_gf, (m, "A", mla.SURF, "gd_fix")))
        JOBS.append((("gf", m, "A", mla.CORNER, "rnd"), j_gf, (m, "A", mla.CORNER, "rnd")))
        JOBS.append((("gf", m, "A", mla.SURF, "fps"), j_gf, (m, "A", mla.SURF, "fps")))
        JOBS.append((("gf", m, "B", mla.CORNER, "fps"), j_gf, (m, "B", mla.CORNER, "fps")))
        JOBS.append((("s2m_gf", m, "A", "fps"), j_s2m_gf, (m, "A", "fps")))
        for subset in ((0, 1, 2, 3), (1,), (0, 2)):
            JOBS.append((("blocks", m, subset), j_blocks, (m, subset)))
    JOBS.append((("frame", m), j_frame, (m,)))
    JOBS.append((("frame_one_call", m), j_frame_one_call, (m,)))
for i in range(2):
    for leaf in (0.2, 0.4):
        JOBS.append((("voxel_grid", i, leaf), j_voxel_grid, (i, leaf)))
        JOBS.append((("voxel_filter", i, leaf), j_voxel_filter, (i, leaf)))
    JOBS.append((("segment", i), j_segment, (i,)))
JOBS.append((("track",), j_track, ()))
for kind in (mla.SURF, mla.CORNER):
    JOBS.append((("downsample", kind), j_downsample, (kind,)))

t0 = time.time()
expected = {}
r0 = np.random.default_rng(0)
for key, fn, args in JOBS:
    if key in expected:
        continue
    c = mla.Context(0)
    try:
        out, _ = fn(c, St(), r0, *args)
    finally:
        c.close()
    expected[key] = [np.asarray(a).copy() for a in out]
print(f"{len(expected)} distinct jobs ({len(JOBS)} with their variants), expected outputs from fresh contexts in {time.time() - t0:.1f} s", flush=True)

n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import threading
results, failed = {}, []


def run(tid):
    """one long-lived context of its own per thread (the reference's estimator runs extractCloud from NUM_OF_LASER OpenMP threads, the mapper in its own node):
    contexts must not see each other"""
    rng_t = np.random.default_rng(seed + 1000 * tid)
    ctx = mla.Context(0)
    st = St()
    n_ops, trace, counts = 0, [], {}
    t0 = time.time()
    while time.time() - t0 < budget and not failed:
        key, fn, args = JOBS[int(rng_t.integers(len(JOBS)))]
        if rng_t.random() < 0.02:
            ctx.set_gn_schedule(int(rng_t.integers(2)), int(rng_t.integers(2)), int(rng_t.integers(2)))
            trace.append("schedule")
        out, how = fn(ctx, st, rng_t, *args)
        n_ops += 1
        counts[key[0]] = counts.get(key[0], 0) + 1
        trace.append(f"{args}:{how}")
        bad = [i for i, (a, b) in enumerate(zip(out, expected[key])) if np.asarray(a).tobytes() != b.tobytes()]
        if bad or len(out) != len(expected[key]):
            msg = [f"MISMATCH in thread {tid} after {n_ops} jobs in {key} {args} (outputs {bad} differ; staging: {how})"]
            for i in bad:
                a, b = np.asarray(out[i]), expected[key][i]
                if a.shape != b.shape:
                    msg.append(f"  output {i}: shape {a.shape} vs expected {b.shape}")
                    continue
                w = np.nonzero((a != b).ravel() & ~((a != a) & (b != b)).ravel())[0]
                msg.append(f"  output {i}: {len(w)} of {a.size} elements differ; first at flat index {w[:4]}: got {a.ravel()[w[:4]]} expected {b.ravel()[w[:4]]}")
            msg.append("last jobs: " + " | ".join(trace[-25:]))
            failed.append("\n".join(msg))
            return
    ctx.close()
    results[tid] = (n_ops, counts)


t0 = time.time()
threads = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
for t in threads:
    t.start()
for t in threads:
    t.join()
if failed:
    print(failed[0])
    sys.exit(1)
tot = sum(r[0] for r in results.values())
counts = {}
for r in results.values():
    for k, v in r[1].items():
        counts[k] = counts.get(k, 0) + v
print(f"API soak: {tot} jobs on {n_threads} long-lived context(s){' in concurrent threads' if n_threads > 1 else ''}, every output equal to a fresh context's, bit for bit; "
      f"seed {seed}, {time.time() - t0:.0f} s; per family: {counts}")
