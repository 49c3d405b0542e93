"""host-side cost of the per-frame calls (wall time of each call in the pipelined loop)"""
import importlib, os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, torch
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sc, surf_map, corner_map, gt, scans = bench.build_workload(synth, "500k")
p0 = synth.perturbed_pose(gt, seed=43)
ctx = mla.Context(0)
ex = []
for s in scans:
    ctx.scan_upload(s.points, s.scan_start, s.scan_end); ctx.extract_run(); ex.append(ctx.extract_fetch())
surf, corner = bench.fuse_features(synth, scans, ex)
dsm, dcm = torch.from_numpy(surf_map).cuda(), torch.from_numpy(corner_map).cuda()
torch.cuda.synchronize()
ctx.map_set_pair(dsm, dcm)
ctx.features_set(mla.SURF, surf); ctx.features_set(mla.CORNER, corner)
opts = mla.default_opts()
for overlapped in (True, False):
    ctx.gn_solve_begin(p0, 5, opts)
    t = {"stage": 0.0, "end": 0.0, "begin": 0.0}
    n = 400
    for i in range(n + 50):
        if i == 50:
            t = {k: 0.0 for k in t}; t_all = time.perf_counter()
        a = time.perf_counter()
        (ctx.map_set_pair_overlapped if overlapped else ctx.map_set_pair)(dsm, dcm)
        b = time.perf_counter()
        ctx.gn_solve_end()
        c = time.perf_counter()
        ctx.gn_solve_begin(p0, 5, opts)
        d = time.perf_counter()
        t["stage"] += b - a; t["end"] += c - b; t["begin"] += d - c
    ctx.gn_solve_end(); ctx.synchronize()
    tot = time.perf_counter() - t_all
    print("overlapped" if overlapped else "same stream", "per frame us: ", {k: round(1e6 * v / n, 1) for k, v in t.items()}, "loop", round(1e6 * tot / n, 1))
