"""ImageSegmenter::segmentCloud on one 64-ring scan: wall time of mlh_segment_cloud (device-resident input, nothing fetched) and of its phases
(MLH_SEG_TIMING=1: the library prints them), next to the CPU oracle. VERDICT r02 item 8: what does the host hop of the cluster search cost?"""
import importlib, os, sys, time
import numpy as np
os.environ["MLH_SEG_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["500k"])
for rings, vs in ((64, 64), (16, 16)):
    s = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[0], rings, seed=3)
    rng = np.random.default_rng(3)
    pts = s.points.copy(); pts[:, 3] = 0.5
    m = rng.random(len(pts)) < 0.1
    pts[m, :3] *= rng.uniform(0.5, 1.3, (int(m.sum()), 1)).astype(np.float32)
    if os.environ.get("SEGBENCH_ORDER", "firing") == "shuffled":
        pts = pts[rng.permutation(len(pts))]
    else:      # firing order (a driver's): azimuth step by azimuth step, starting inside the sweep; SEGBENCH_ORDER=firing_rev: the other sense of rotation
        az = np.mod(np.arctan2(pts[:, 1], pts[:, 0]) - 1.0, 2 * np.pi)
        o = np.argsort(az, kind="stable")
        pts = np.ascontiguousarray(pts[o[::-1] if os.environ.get("SEGBENCH_ORDER") == "firing_rev" else o])
    ctx = mla.Context(0)
    d = torch.from_numpy(pts).cuda(); torch.cuda.synchronize()
    for _ in range(3):
        ctx.segment_cloud(d, fetch=False, vertical_scans=vs)
    sys.stderr.flush()
    print(f"--- {rings} rings, {len(pts)} points", file=sys.stderr)
    t0 = time.perf_counter(); n = 10
    for _ in range(n):
        ctx.segment_cloud(d, fetch=False, vertical_scans=vs)
    ctx.synchronize(); gpu_ms = 1e3 * (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); O.segment_cloud(pts, O.seg_params(vertical_scans=vs)); cpu_ms = 1e3 * (time.perf_counter() - t0)
    print(f"{rings} rings, {len(pts)} points: mlh_segment_cloud (device-resident in, scan staged on the device, nothing fetched) {gpu_ms:.3f} ms per call; CPU oracle {cpu_ms:.2f} ms")
    ctx.close()
    # As the reference calls it: NUM_OF_LASER OpenMP threads, one segmentCloud each (estimator.cpp:249-263) -- the host-side cluster search of one LiDAR runs
    # beside the other LiDAR's (the facade's ImageSegmenter takes a context per thread). Per scan of caller-visible time = the pair's wall time / 2.
    import threading
    sys.stderr.flush()
    _saved_err = os.dup(2)                      # the library's phase lines (MLH_SEG_TIMING is read once, at the first call) would flood the threaded legs: stderr away
    _null = os.open(os.devnull, os.O_WRONLY); os.dup2(_null, 2); os.close(_null)
    for n_thr in (2, 4):
        ctxs = [mla.Context(0) for _ in range(n_thr)]
        for c in ctxs:
            c.segment_cloud(d, fetch=False, vertical_scans=vs)
        walls = []
        for rep in range(3):
            bar = threading.Barrier(n_thr + 1)
            def work(c):
                bar.wait()
                for _ in range(40):
                    c.segment_cloud(d, fetch=False, vertical_scans=vs)
                c.synchronize()
            th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
            for t in th: t.start()
            bar.wait(); t0 = time.perf_counter()
            for t in th: t.join()
            walls.append(1e3 * (time.perf_counter() - t0) / 40)
        wall = sorted(walls)[1]          # median of three runs of 40 sets
        print(f"    {n_thr} LiDARs segmented by {n_thr} threads at once (a context each): {wall:.3f} ms per set = {wall / n_thr:.3f} ms per scan of caller-visible time")
        for c in ctxs: c.close()
    sys.stderr.flush(); os.dup2(_saved_err, 2); os.close(_saved_err)
