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
for preset in ("500k", "4M"):
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
    for _ in range(3): out = frame()
    ctx.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): out = frame()
    ctx.synchronize(); gpu_ms = 1e3 * (time.perf_counter() - t0) / n
    poses = out[0]
    nf = sum(len(x) for x in surf_b) + sum(len(x) for x in corner_b)
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
    ctx.close()
