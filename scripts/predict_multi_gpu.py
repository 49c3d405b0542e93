"""What the first multi-GPU run must show ON THE CONFIGURATIONS THE NORTH STAR QUOTES ITS TARGET ON, stated before such a run exists (no multi-GPU box has been
reachable in any round; SCALE_r0N.json are `skipped` records). Same method as bench.py's `multi_gpu_predicted` (its docstring: every prospective rank's share alone on
this GPU -- index build of its maps + one classic-schedule iteration by the dispatches' own timestamps, degeneracy test off -- predicted step = the slowest rank's index
build + 5 x (correspondence + fit + an ASSUMED 5 us exchange)), for
  (a) BASELINE config 4 -- 4 x 64 rings, one pose block per LiDAR -- on the 4 M-point map ("4 M-point map, 8 x MI355X"; "at least 6x strong scaling at 8 GPUs on a
      4 M-point map"): angular wedges of the map at N = 2 / 4 / 8, and the exchange-free split of the pose BLOCKS over the ranks (whole map on every rank) at N = 2 / 4;
  (b) the un-thinned 2 x 64-ring frame (228 k queries: where the kernels leave the latency regime, bench.py roofline.saturated) on the 500 k map, wedges and
      round-robin features at N = 2 / 4 / 8.
Writes one JSON object (argv[1], default profiles/r06_multi_gpu_predicted.json) with `meets_target` against the 6x. usage: python scripts/predict_multi_gpu.py [out.json]"""
import importlib, json, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth"); shard = importlib.import_module("m-loam_amd.shard")
import torch
torch.cuda.init()
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_multi_gpu_predicted.json")
EX_US, GN = 5.0, 5
res = dict(method="bench.py predicted_scaling: each prospective rank's share alone on ONE MI355X; predicted step = slowest rank's index build + 5 x (correspondence + fit + "
                  "assumed exchange)", assumed_exchange_us=EX_US, north_star_target="at least 6x strong scaling at 8 GPUs on a 4 M-point map")

# ---------------------------------------------------------------- (b) the un-thinned frame on the 500 k map
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
ctx = mla.Context(0)
ex = [ctx.extract(s.points, s.scan_start, s.scan_end) for s in scans]
ctx.close()
surf_d, corner_d = bench.fuse_features(synth, scans, ex, thin=False)
center = np.asarray(gt[:3], np.float64)
p0 = synth.perturbed_pose(gt, seed=43)
c = mla.Context(0)
c.map_set_pair(surf_map, corner_map); c.features_set(mla.SURF, surf_d); c.features_set(mla.CORNER, corner_d)
p_conv = c.gn_solve(p0, 5, want_stats=False)[0]
c.close()
b = bench.predicted_scaling(mla, torch, shard, 0, surf_map, corner_map, surf_d, corner_d, np.asarray(p_conv, np.float64), center, reps=20)
res["config2_unthinned_500k"] = dict(features=dict(surf=int(len(surf_d)), corner=int(len(corner_d))), n1_ms_per_step=b["n1_same_method_ms_per_step"],
                                     splits={k: dict(predicted_ms_per_step=v["predicted_ms_per_step"], predicted_speedup_vs_n1=v["predicted_speedup_vs_n1"]) for k, v in b["splits"].items()})
print("config 2, un-thinned features, 500k map:", json.dumps(res["config2_unthinned_500k"]), flush=True)

# ---------------------------------------------------------------- (a) config 4 on the 4 M map
t0 = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "4M", n_lidars=4)
print(f"4M workload built in {time.time() - t0:.0f} s: map {len(surf_map)} + {len(corner_map)} points", flush=True)
k_neigh, thre, freeze = [5, 10, 10, 10], [-1.0] * 4, [0, 0, 0, 0]          # (degeneracy test off: a wedge's LOCAL sums are degenerate; the real iteration solves on the exchanged sums)
ctx = mla.Context(0)
surf_b, corner_b, poses0 = [], [], []
from scipy.spatial.transform import Rotation as Rot
for i, s in enumerate(scans):
    e = ctx.extract(s.points, s.scan_start, s.scan_end, voxel_leaf=0.2)
    cc = np.zeros((len(e["less_sharp"]), 4), np.float32); cc[:, :3] = s.points[e["less_sharp"]][:, :3]
    surf_b.append(np.ascontiguousarray(synth.voxel_mean(e["less_flat_ds"].copy(), 0.4)))
    corner_b.append(np.ascontiguousarray(synth.voxel_mean(cc, 0.2)))
    bl = synth.HERCULES_BODY_T_LASER[i]
    T = synth.pose_to_mat(gt) @ np.block([[synth.quat_to_rot(bl[:4]), bl[4:7, None]], [np.zeros((1, 3)), np.ones((1, 1))]])
    poses0.append(np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_quat()]))       # at the ground truth: the converged regime
poses0 = np.array(poses0)
center = np.asarray(gt[:3], np.float64)
opts = mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=1.0)
ctx.set_gn_schedule(0, 0, 0)
far = np.full((1, 3), 1.0e6, np.float32)


def share(lsm, lcm, blocks, planes=None, reps=20):
    """index build + one classic iteration of `blocks` against (lsm, lcm), alone on this GPU"""
    d_sm, d_cm = torch.from_numpy(np.ascontiguousarray(lsm)).cuda(), torch.from_numpy(np.ascontiguousarray(lcm)).cuda()
    torch.cuda.synchronize()
    ctx.shard_set(None, None)
    if planes is not None:
        ctx.shard_set(*planes)
    ctx.map_set_pair(d_sm, d_cm)
    ctx.features_set_blocks(mla.SURF, [surf_b[i] for i in blocks]); ctx.features_set_blocks(mla.CORNER, [corner_b[i] for i in blocks])
    kn, th, fr, ps = [k_neigh[i] for i in blocks], [thre[i] for i in blocks], [freeze[i] for i in blocks], poses0[blocks]
    for _ in range(5):
        ctx.map_set_pair(d_sm, d_cm); ctx.gn_solve_blocks(ps, 1, kn, th, fr, opts, want_stats=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.map_set_pair(d_sm, d_cm)
    ctx.synchronize()
    build_us = 1e6 * (time.perf_counter() - t0) / reps
    ctx.profile_enable((1 << mla.K_KNN) | (1 << mla.K_FIT)); ctx.profile_reset()
    for _ in range(reps):
        ctx.gn_solve_blocks(ps, 1, kn, th, fr, opts, want_stats=False)
    knn_ms, knn_n = ctx.profile_get(mla.K_KNN); fit_ms, fit_n = ctx.profile_get(mla.K_FIT)
    ctx.profile_enable(0)
    return dict(index_build_us=round(build_us, 1), knn_us=round(1e3 * knn_ms / max(knn_n, 1), 2), fit_us=round(1e3 * fit_ms / max(fit_n, 1), 2))


def step_ms(ranks, exchange):
    return max(r["index_build_us"] + GN * (r["knn_us"] + r["fit_us"] + (EX_US if exchange else 0.0)) for r in ranks) * 1e-3


all_blocks = [0, 1, 2, 3]
n1 = [share(surf_map, corner_map, all_blocks)]
cfg4 = dict(map_points=int(len(surf_map) + len(corner_map)), features=int(sum(len(x) for x in surf_b) + sum(len(x) for x in corner_b)), n1_per_rank_alone=n1,
            n1_ms_per_step=round(step_ms(n1, False), 4), splits={})
for n in (2, 4, 8):
    ranks = []
    for r in range(n):
        ms_ = shard.shard_points_mask(surf_map, center, n, r); mc_ = shard.shard_points_mask(corner_map, center, n, r)
        lsm, lcm = surf_map[ms_], corner_map[mc_]
        ranks.append(share(lsm if len(lsm) else far, lcm if len(lcm) else far, all_blocks, shard.wedge_planes(center, n, r)))
    st = step_ms(ranks, True)
    cfg4["splits"][f"wedges_n{n}"] = dict(per_rank_alone=ranks, predicted_ms_per_step=round(st, 4), predicted_speedup_vs_n1=round(cfg4["n1_ms_per_step"] / st, 3))
    print(f"config 4 / 4M, wedges N = {n}: {st:.4f} ms per step, {cfg4['n1_ms_per_step'] / st:.2f}x", flush=True)
for n, deal in ((2, [[0, 1], [2, 3]]), (4, [[0], [1], [2], [3]])):
    ranks = [share(surf_map, corner_map, blk) for blk in deal]
    st = step_ms(ranks, False)                                   # (independent blocks: no exchange)
    cfg4["splits"][f"blocks_over_ranks_n{n}"] = dict(per_rank_alone=ranks, predicted_ms_per_step=round(st, 4), predicted_speedup_vs_n1=round(cfg4["n1_ms_per_step"] / st, 3))
    print(f"config 4 / 4M, blocks over {n} ranks: {st:.4f} ms per step, {cfg4['n1_ms_per_step'] / st:.2f}x", flush=True)
ctx.close()
res["config4_4M"] = cfg4
best8 = max([v["predicted_speedup_vs_n1"] for k, v in cfg4["splits"].items() if k.endswith("n8")] + [cfg4["splits"]["blocks_over_ranks_n4"]["predicted_speedup_vs_n1"]])
res["predicted_best_speedup_at_8_gpus_config4_4M"] = best8
res["meets_target"] = bool(best8 >= 6.0)
res["verdict"] = ("the north star's 6x at 8 GPUs on the 4 M map is NOT what a run will show: a frame of this size is a chain of short launches whatever a rank's share, and the only "
                  "term that shrinks with N is the index build (and, un-thinned, the correspondence launch); the first cross-GPU SCALE run is to be read against these figures, "
                  "not against 6x")
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print("wrote", out_path, "meets_target:", res["meets_target"], "best at 8 GPUs:", best8)
