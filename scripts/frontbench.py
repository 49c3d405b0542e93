"""The estimator's front end through the C++ facade, as a caller sees it: writes NUM_OF_LASER raw 64-ring clouds (the bench scene's two LiDARs,
firing order, 10 % clutter as scripts/segbench.py) and runs m-loam_amd/host/frontbench on them: one thread one LiDAR after the other | one calling thread + the
facade's own lanes (FrontEndLanes) | the reference's OpenMP loop; per LiDAR the clouds must be equal bit for bit.   python scripts/frontbench.py [frames]"""
import importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("m-loam_amd.synth")
frames = sys.argv[1] if len(sys.argv) > 1 else "40"
exe = os.path.join(ROOT, "m-loam_amd", "host", "frontbench")
if not os.path.exists(exe):
    subprocess.run(["make", "-C", os.path.dirname(exe), "-s", "frontbench"], check=True)
scn = synth.make_scene(seed=42, **synth.SCENE_PRESETS["500k"])
with tempfile.TemporaryDirectory() as d:
    for rings in (64, 16):
        for l in range(2):
            s = synth.simulate_scan(scn, synth.gt_body_pose(), synth.HERCULES_BODY_T_LASER[l], rings, seed=3 + l)
            rng = np.random.default_rng(3 + l)
            pts = s.points.copy(); pts[:, 3] = 0.0
            m = rng.random(len(pts)) < 0.1
            pts[m, :3] *= rng.uniform(0.5, 1.3, (int(m.sum()), 1)).astype(np.float32)
            az = np.mod(np.arctan2(pts[:, 1], pts[:, 0]) - 1.0, 2 * np.pi)          # firing order, starting inside the sweep
            np.ascontiguousarray(pts[np.argsort(az, kind="stable")], np.float32).tofile(os.path.join(d, f"raw_{l}.f32"))
        r = subprocess.run([exe, d, "2", str(rings), frames], capture_output=True, text=True, timeout=600)
        sys.stderr.write(r.stderr[-2000:])
        print(r.stdout.strip())
        if r.returncode != 0:
            sys.exit(f"frontbench failed ({r.returncode})")
