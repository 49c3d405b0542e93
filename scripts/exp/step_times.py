"""Wall time of every step of bench.py's pipelined loop in a short run (the driver's form: 5 warm-up + 20 timed): where filling and draining the pipeline go."""
import importlib, os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
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
conv, _ = ctx.gn_solve(p0, 5, opts, want_stats=False)
a_ = torch.randn(2048, 2048, device="cuda"); t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    for _ in range(8): a_ = (a_ @ a_) * 1e-3
    torch.cuda.synchronize()
def run(n):
    ts = []
    infl = False
    torch.cuda.synchronize(); ctx.synchronize()
    t_begin = time.perf_counter()
    for k in range(n):
        ctx.map_set_pair_overlapped(dsm, dcm)
        if infl:
            ctx.gn_solve_begin_chained(conv, p0, 5, opts); ctx.gn_solve_end()
        else:
            ctx.gn_solve_begin(p0, 5, opts)
        infl = True
        ts.append(time.perf_counter())
    ctx.gn_solve_end(); t_drain = time.perf_counter()
    torch.cuda.synchronize(); ctx.synchronize(); t_sync = time.perf_counter()
    d = np.diff([t_begin] + ts) * 1e6
    return d, (t_drain - ts[-1]) * 1e6, (t_sync - t_drain) * 1e6, (t_sync - t_begin) * 1e6 / n
run(5)
for rep in range(3):
    d, drain, sync, per = run(20)
    print("steps (us):", " ".join(f"{x:.0f}" for x in d), "| drain %.0f | final sync %.0f | per step %.1f" % (drain, sync, per))
