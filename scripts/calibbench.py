"""BASELINE config 4 on ONE GPU: 4 x 64-ring scan, one pose block per LiDAR (body pose + 3 extrinsics; N_NEIGH 5 / 10 / 10 / 10, CHECK_FOV,
freeze-on-degenerate), 5 GN iterations with re-matching, against the 4 M-point map (a 288 GB GPU holds it whole) and the 500 k map.
GPU (mlh_gn_solve_blocks) next to the CPU oracle's per-block iterations."""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench, oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
k_neigh, thre, freeze = [5, 10, 10, 10], [100.0, 70.0, 70.0, 70.0], [0, 1, 1, 1]
for preset in os.environ.get("CALIBBENCH_PRESETS", "500k,4M").split(","):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, preset, n_lidars=4)
    ctx = mla.Context(0)
    surf_b, corner_b, poses0 = [], [], []
    for i, s in enumerate(scans):
        ex = ctx.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
        c = np.zeros((len(ex["less_sharp"]), 4), np.float32); c[:, :3] = s.points[ex["less_sharp"]][:, :3]
        surf_b.append(np.ascontiguousarray(synth.voxel_mean(ex["less_flat_ds"].copy(), 0.4)))
        corner_b.append(np.ascontiguousarray(synth.voxel_mean(c, 0.2)))
        bl = synth.HERCULES_BODY_T_LASER[i]
        T = synth.pose_to_mat(gt) @ np.block([[synth.quat_to_rot(bl[:4]), bl[4:7, None]], [np.zeros((1, 3)), np.ones((1, 1))]])
        from scipy.spatial.transform import Rotation as Rot
        gt_i = np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
        poses0.append(synth.perturbed_pose(gt_i, seed=50 + i, dt=0.1, drot_deg=1.0))
    poses0 = np.array(poses0)
    t0 = time.perf_counter(); ctx.map_set(mla.SURF, surf_map); ctx.map_set(mla.CORNER, corner_map); ctx.synchronize(); t_set = 1e3 * (time.perf_counter() - t0)
    ctx.features_set_blocks(mla.SURF, surf_b); ctx.features_set_blocks(mla.CORNER, corner_b)
    opts = mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=1.0)
    n_it = 5
    def frame():
        ctx.map_rebuild(mla.ALL_KINDS)
        return ctx.gn_solve_blocks(poses0, n_it, k_neigh, thre, freeze, opts, want_stats=False)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.15: out = frame()      # busy GPU before the timed region (see bench.py: short regions after an idle phase read 3-12x slow twice)
    ctx.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): out = frame()
    ctx.synchronize(); gpu_ms = 1e3 * (time.perf_counter() - t0) / n
    poses = out[0]
    nf = sum(len(x) for x in surf_b) + sum(len(x) for x in corner_b)
    print("  map index:", ctx.map_info(mla.SURF), ctx.map_info(mla.CORNER), "features surf/corner per block:", [len(x) for x in surf_b], [len(x) for x in corner_b])
    ms, mc = O.Map(surf_map), O.Map(corner_map)
    t0 = time.perf_counter(); dts = []
    for b in range(4):
        prm = O.mapper_params(huber_delta=1.0, map_eig_thre=thre[b], n_neigh=k_neigh[b], check_fov=True, freeze_when_degenerate=bool(freeze[b]))
        ref = O.gn_iterations(ms, mc, surf_b[b], corner_b[b], poses0[b], prm, n_it)
        dts.append(np.linalg.norm(poses[b][:3] - ref["pose"][:3]))
    cpu_ms = 1e3 * (time.perf_counter() - t0)
    tk = 1e3 * (ms.rebuild_seconds() + mc.rebuild_seconds())
    print(f"config 4 on one GPU, {preset} map ({len(surf_map) + len(corner_map)} points), {nf} features in 4 pose blocks: index rebuild + {n_it} GN iterations "
          f"GPU {gpu_ms:.3f} ms ({nf * n_it / gpu_ms * 1e3:.3g} features/s); CPU oracle {cpu_ms:.0f} ms + kd-tree build {tk:.0f} ms; max |dt| over blocks {max(dts):.1e} m; first map_set {t_set:.1f} ms")
    # ---- each block ALONE on this GPU (what one of four GPUs runs when bench.py deals the pose blocks over the ranks: whole map, one block, no exchange): the
    # slowest block is the frame time four GPUs would have, measured here on one -- a projection, not a multi-GPU measurement
    alone = []
    for b in range(4):
        ctx.features_set_blocks(mla.SURF, [surf_b[b]]); ctx.features_set_blocks(mla.CORNER, [corner_b[b]])
        def frame1():
            ctx.map_rebuild(mla.ALL_KINDS)
            return ctx.gn_solve_blocks(poses0[b:b + 1].copy(), n_it, k_neigh[b:b + 1], thre[b:b + 1], freeze[b:b + 1], opts, want_stats=False)
        for _ in range(20): o1 = frame1()
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(40): o1 = frame1()
        ctx.synchronize(); alone.append(1e3 * (time.perf_counter() - t0) / 40)
        assert np.array_equal(np.asarray(o1[0])[0], np.asarray(poses)[b]), b
    print(f"  every block by itself on this GPU (index rebuild + {n_it} GN iterations, poses bit-equal to the four-block solve): " + " / ".join(f"{a:.3f}" for a in alone)
          + f" ms -> with the blocks dealt over 4 GPUs (bench.py config4.blocks_over_ranks; no exchange) a frame would take {max(alone):.3f} ms, {gpu_ms / max(alone):.2f}x "
          f"(projected from one GPU; never run on four)")
    ctx.features_set_blocks(mla.SURF, surf_b); ctx.features_set_blocks(mla.CORNER, corner_b)
    # ---- the COUPLED window problem of Estimator::optimizeMap (estimator.cpp:687-848) on the same data: parameter blocks [pivot | 1 frame | 4
    # extrinsics] = 36 local parameters, one LidarPureOdom{PlaneNorm,Edge}Factor per matched feature of every LiDAR; pivot and the reference
    # LiDAR's extrinsic held constant (estimator.cpp:636, 642). Matching on the GPU against the resident map (pivot frame = map frame here),
    # the factor table staged once, then per Gauss-Newton iteration ONE device pass for the 36 x 36 normal equations + a host solve.
    from scipy.spatial.transform import Rotation as Rot
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    frame0 = synth.perturbed_pose(gt, seed=70, dt=0.05, drot_deg=0.5)
    exts0 = []
    for i in range(4):
        bl = synth.HERCULES_BODY_T_LASER[i]
        e = np.concatenate([bl[4:7], bl[:4] / np.linalg.norm(bl[:4])])
        exts0.append(e if i == 0 else synth.perturbed_pose(e, seed=80 + i, dt=0.03, drot_deg=0.3))
    exts0 = np.array(exts0)
    to_pose = lambda T: np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()])
    t0 = time.perf_counter()
    types, points, coeffs, fi, ei = [], [], [], [], []
    for i in range(4):
        rel = to_pose(synth.pose_to_mat(frame0) @ synth.pose_to_mat(exts0[i]))          # T_pivot^-1 T_frame T_ext with T_pivot = I
        for kind, feats, ty in ((mla.SURF, surf_b[i], 0), (mla.CORNER, corner_b[i], 1)):
            ctx.features_set(kind, feats)
            out_m = ctx.match_linearize(kind, rel, flags=mla.FLAG_CHECK_FOV, huber_delta=1.0, dense=False)
            m = out_m["valid"].astype(bool)
            types.append(np.full(m.sum(), ty, np.int32)); points.append(feats[m, :3].astype(np.float64)); coeffs.append(out_m["coeffs"][m])
            fi.append(np.zeros(m.sum(), np.int32)); ei.append(np.full(m.sum(), i, np.int32))
    tab = [np.concatenate(a) for a in (types, points, coeffs, fi, ei)]
    t_match = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter(); ctx.pure_odom_set(*tab); ctx.synchronize(); t_stage = 1e3 * (time.perf_counter() - t0)
    D = 36
    free = np.r_[6:12, 18:36]                                                             # the frame + extrinsics 1..3
    def window_gn(neq, iters=5):
        fr, ex = frame0.copy()[None, :], exts0.copy()
        for _ in range(iters):
            ne = neq(fr, ex)
            step = np.zeros(D); step[free] = np.linalg.solve(ne["H"][np.ix_(free, free)], -ne["g"][free])
            fr[0] = mla.pose_plus(fr[0], step[6:12])
            for k in range(1, 4): ex[k] = mla.pose_plus(ex[k], step[12 + 6 * k:18 + 6 * k])
        return fr, ex, ne
    gpu_neq = lambda fr, ex: ctx.pure_odom_normal_eq(ident, fr, ex, huber_delta=1.0)
    for _ in range(3): window_gn(gpu_neq)
    ctx.synchronize(); t0 = time.perf_counter(); nrep = 20
    for _ in range(nrep): fr_g, ex_g, ne_g = window_gn(gpu_neq)
    ctx.synchronize(); t_gn = 1e3 * (time.perf_counter() - t0) / nrep
    t0 = time.perf_counter()
    for _ in range(200): ctx.pure_odom_normal_eq(ident, fr_g, ex_g, huber_delta=1.0)
    t_ne = 1e3 * (time.perf_counter() - t0) / 200
    # the same five iterations device-resident (mlh_pure_odom_gn_solve: 3 launches per iteration, the poses never leave HBM)
    for _ in range(3): sol = ctx.pure_odom_gn_solve(ident, frame0[None, :], exts0, n_iters=5, huber_delta=1.0)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(nrep): sol = ctx.pure_odom_gn_solve(ident, frame0[None, :], exts0, n_iters=5, huber_delta=1.0)
    ctx.synchronize(); t_dev_gn = 1e3 * (time.perf_counter() - t0) / nrep
    d_dev = max(float(np.linalg.norm(sol["frames"][0][:3] - fr_g[0][:3])), max(float(np.linalg.norm(sol["exts"][k][:3] - ex_g[k][:3])) for k in range(1, 4)))
    print(f"  coupled window solved ON THE DEVICE (mlh_pure_odom_gn_solve, 5 Gauss-Newton iterations, status {sol['status']}): {t_dev_gn:.3f} ms per solve; vs the host-in-the-loop iterations above: {d_dev:.1e} m")
    cpu_neq = lambda fr, ex: O.pure_odom_normal_eq(tab[0], tab[1], tab[2], None, tab[3], tab[4], ident, fr, ex, 1.0)
    t0 = time.perf_counter(); fr_c, ex_c, ne_c = window_gn(cpu_neq); t_cpu = 1e3 * (time.perf_counter() - t0)
    dH = float(np.abs(ne_g["H"] - ne_c["H"]).max() / np.abs(ne_c["H"]).max())
    dpose = max(float(np.linalg.norm(fr_g[0][:3] - fr_c[0][:3])), max(float(np.linalg.norm(ex_g[k][:3] - ex_c[k][:3])) for k in range(1, 4)))
    # the same window with the factor table built ON THE DEVICE (mlh_pure_odom_begin / _add_matches): no validity / coefficient copies, no staging
    def build_on_device():
        ctx.pure_odom_begin()
        for i in range(4):
            rel = to_pose(synth.pose_to_mat(frame0) @ synth.pose_to_mat(exts0[i]))
            for kind, feats_k in ((mla.SURF, surf_b[i]), (mla.CORNER, corner_b[i])):
                ctx.features_set(kind, feats_k)
                ctx.pure_odom_add_matches(kind, rel, 0, i, k_neigh=5, flags=mla.FLAG_CHECK_FOV)
    for _ in range(3): build_on_device()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10): build_on_device()
    ctx.synchronize(); t_dev_build = 1e3 * (time.perf_counter() - t0) / 10
    ne_d = ctx.pure_odom_normal_eq(ident, frame0[None, :], exts0, huber_delta=1.0)
    ctx.pure_odom_set(*tab)
    ne_h = ctx.pure_odom_normal_eq(ident, frame0[None, :], exts0, huber_delta=1.0)
    dev_vs_host = float(np.abs(ne_d["H"] - ne_h["H"]).max() / np.abs(ne_h["H"]).max())
    print(f"  factor table built on the device ({ne_d['count']} factors = {ne_h['count']} host-staged): 8 x (features_set + match pass + append) {t_dev_build:.2f} ms "
          f"(of which the host->device feature uploads; was {t_match + t_stage:.2f} ms through the host), normal equations vs the host-staged table: max rel |dH| {dev_vs_host:.1e}")
    # the same table behind the ODOMETRY's good-feature selection (Estimator::goodFeatureMatching, ODOM_GF_RATIO = 0.8): 8 x (features_set + match + scored rows +
    # the host draw loop + append of the selected); CPU: the oracle's restatement of the same calls
    def build_selected(ratio):
        ctx.pure_odom_begin()
        n_sel = 0
        for i in range(4):
            rel = to_pose(synth.pose_to_mat(frame0) @ synth.pose_to_mat(exts0[i]))
            for kind, feats_k in ((mla.SURF, surf_b[i]), (mla.CORNER, corner_b[i])):
                ctx.features_set(kind, feats_k)
                n_sel += len(ctx.pure_odom_add_matches_gf(kind, rel, ident, frame0, exts0[i], 0, i, gf_ratio=ratio, seed=3 + i))
        return n_sel
    for _ in range(3): n_sel = build_selected(0.8)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10): n_sel = build_selected(0.8)
    ctx.synchronize(); t_sel = 1e3 * (time.perf_counter() - t0) / 10
    t0 = time.perf_counter(); n_cpu = 0
    for i in range(4):
        rel = to_pose(synth.pose_to_mat(frame0) @ synth.pose_to_mat(exts0[i]))
        for ch, om, feats_k in (("s", ms, surf_b[i]), ("c", mc, corner_b[i])):
            n_cpu += len(O.odom_good_feature_matching(om, ch, feats_k, rel, ident, frame0, exts0[i], 0.8, 3 + i)["sel"])
    t_sel_cpu = 1e3 * (time.perf_counter() - t0)
    print(f"  the odometry's good-feature selection in front of the table (Estimator::goodFeatureMatching, ODOM_GF_RATIO 0.8): 8 x (features_set + match + scored rows + host draw "
          f"loop + append) {t_sel:.2f} ms, {n_sel} factors selected (CPU oracle, same calls: {t_sel_cpu:.1f} ms, {n_cpu} selected)")
    print(f"  coupled window system on the {preset} map: {len(tab[0])} factors, D = {D} (pivot | 1 frame | 4 extrinsics): GPU matching of 4 LiDARs x 2 kinds "
          f"{t_match:.2f} ms (8 launches, host copies of validity + coefficients), table staging {t_stage:.2f} ms, one normal-equation pass {t_ne:.3f} ms, "
          f"5 coupled GN iterations (device J^T J / J^T r + host 24-dim solve + Plus) {t_gn:.3f} ms vs CPU oracle accumulation {t_cpu:.1f} ms; "
          f"max rel |dH| {dH:.1e}, pose agreement after 5 iterations {dpose:.1e} m")
    ctx.close()
