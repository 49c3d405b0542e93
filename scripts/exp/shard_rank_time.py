"""Why are some wedges slower than the whole map (bench.py: multi_gpu_predicted)? Per prospective rank of an N-way wedge split: local map sizes, owned features,
ms of map_set_pair alone and of the 5-iteration solve alone, the grid's geometry and lane choice."""
import importlib, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth"); shard = importlib.import_module("m-loam_amd.shard")
import torch
torch.cuda.init()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
p0 = synth.perturbed_pose(gt, seed=43)
c = mla.Context(0)
ex = []
for s in scans:
    c.scan_upload(s.points, s.scan_start, s.scan_end); c.extract_run(); ex.append(c.extract_fetch())
surf, corner = bench.fuse_features(synth, scans, ex, thin=True)
center = p0[:2]
Tm = synth.pose_to_mat(p0)
d_s, d_c = torch.from_numpy(surf).cuda(), torch.from_numpy(corner).cuda()
opts = mla.default_opts()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sched = tuple(int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "000"))
for r in range(N):
    ms_ = shard.shard_points_mask(surf_map, center, N, r); mc_ = shard.shard_points_mask(corner_map, center, N, r)
    lsm, lcm = np.ascontiguousarray(surf_map[ms_]), np.ascontiguousarray(corner_map[mc_])
    d_sm, d_cm = torch.from_numpy(lsm).cuda(), torch.from_numpy(lcm).cuda()
    torch.cuda.synchronize()
    lo, hi = shard.wedge_planes(center, N, r)
    c.set_gn_schedule(*sched)
    c.shard_set(lo, hi)
    c.map_set_pair(d_sm, d_cm); c.features_set(mla.SURF, d_s); c.features_set(mla.CORNER, d_c)
    own_s = shard.owned_mask(synth.transform_points(surf[:, :3], Tm), lo, hi).sum(); own_c = shard.owned_mask(synth.transform_points(corner[:, :3], Tm), lo, hi).sum()
    for _ in range(8):
        c.map_set_pair(d_sm, d_cm); c.gn_solve(p0, 5, opts, want_stats=False)
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): c.map_set_pair(d_sm, d_cm)
    c.synchronize(); t_map = (time.perf_counter() - t0) / 40
    t0 = time.perf_counter()
    for _ in range(40): c.gn_solve(p0, 5, opts, want_stats=False)
    c.synchronize(); t_solve = (time.perf_counter() - t0) / 40
    print(f"rank {r}/{N}: map {len(lsm)}/{len(lcm)} owned {own_s}/{own_c}  map_set_pair {1e3 * t_map:.3f} ms  solve {1e3 * t_solve:.3f} ms  info {c.map_info(mla.SURF)} {c.map_info(mla.CORNER)}", flush=True)
