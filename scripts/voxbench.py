import sys, time, numpy as np
sys.path.insert(0, '.')
import importlib
mla = importlib.import_module('m-loam_amd')
ctx = mla.Context(0)
rng = np.random.default_rng(0)
for n in (200_000, 2_000_000):
    pts = np.zeros((n, 11), np.float32)
    pts[:, :2] = rng.uniform(-150, 150, (n, 2)); pts[:, 2] = rng.uniform(-5, 20, n)
    pts[:, 4] = pts[:, 7] = pts[:, 9] = 0.01
    for leaf in (0.4, 0.2):
        ctx.voxel_filter(pts, leaf, 1.0)
        t = time.perf_counter(); out = ctx.voxel_filter(pts, leaf, 1.0); dt = time.perf_counter() - t
        print(f"n={n} leaf={leaf} out={len(out)} host-to-host {dt*1e3:.2f} ms")
