import importlib, os, sys
import numpy as np
os.environ["MLOAM_SCENE_FAMILY"] = "hard"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest, oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
O.build()
sseed, n_rings, n_lidars = 128839, 32, 2
case = conftest._make_case(synth, "50k", n_rings, n_lidars, seed=sseed)
feats = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
p0 = synth.perturbed_pose(case["gt"], seed=sseed + 1, dt=1.5, drot_deg=0.2)
msd, huber, fov = 0.64, 0.1, True
ctx = mla.Context(0)
ctx.map_set(mla.SURF, case["surf_map"], min_match_sq_dis=msd); ctx.map_set(mla.CORNER, case["corner_map"], min_match_sq_dis=msd)
ctx.features_set(mla.SURF, feats[0]); ctx.features_set(mla.CORNER, feats[1])
maps = (O.Map(case["surf_map"]), O.Map(case["corner_map"]))
prm = O.mapper_params(huber_delta=huber, n_neigh=5, check_fov=fov, min_match_sq_dis=msd)
opts = mla.default_opts(flags=mla.FLAG_CHECK_FOV, huber_delta=huber, min_match_sq_dis=msd)
pose2, st2 = ctx.scan2map(p0, opts)
ref2 = O.scan2map(maps[0], maps[1], feats[0], feats[1], p0, prm)
print("ref outer keys:", list(ref2["outer"][0].keys()))
for o_, (s, r) in enumerate(zip(st2, ref2["outer"])):
    print("outer", o_, "HIP n", s["n_surf"], s["n_corner"], "lm", s["lm_iterations"], s["successful_steps"], s["termination"], "cost", s["cost"], s["final_cost"])
    print("        ORC", {k: r[k] for k in r if k not in ("H", "g", "eigval", "H0", "pose_after")})
    print("   HIP eig", np.asarray(s["eigval"]))
    if "pose_after" in r:
        print("   pose_after diff", np.abs(np.asarray(s["pose_after"]) - np.asarray(r["pose_after"])).max())
print("final pose diff", np.abs(pose2 - ref2["pose"]).max())
# decisions at HIP's own first-outer pose: HIP vs oracle at the SAME pose
pa = np.asarray(st2[0]["pose_after"])
for kind, ch in ((mla.SURF, "s"), (mla.CORNER, "c")):
    got = ctx.match_linearize(kind, pa, flags=mla.FLAG_CHECK_FOV, min_match_sq_dis=msd, huber_delta=huber)
    valid, coeffs = maps[kind].match(ch, feats[kind], pa, n_neigh=5, check_fov=fov, min_match_sq_dis=msd)
    print(ch, "at HIP's pose after outer 0: flips HIP vs oracle", int((got["valid"] != valid).sum()), "valid", int(valid.sum()))
    if "pose_after" in ref2["outer"][0]:
        pb = np.asarray(ref2["outer"][0]["pose_after"])
        v2, c2 = maps[kind].match(ch, feats[kind], pb, n_neigh=5, check_fov=fov, min_match_sq_dis=msd)
        print(ch, "   oracle at ITS pose vs at HIP's pose: flips", int((v2 != valid).sum()), " |pose diff|", np.abs(pa - pb).max())
print("---- counts at p0")
for kind, ch in ((mla.SURF, "s"), (mla.CORNER, "c")):
    got = ctx.match_linearize(kind, p0, flags=mla.FLAG_CHECK_FOV, min_match_sq_dis=msd, huber_delta=huber)
    valid, coeffs = maps[kind].match(ch, feats[kind], p0, n_neigh=5, check_fov=fov, min_match_sq_dis=msd)
    print(ch, "match_linearize valid", int(got["valid"].sum()), "oracle", int(valid.sum()), "flips", int((got["valid"] != valid).sum()))
pg, sg = ctx.gn_solve(p0, 1, opts)
print("gn_solve iteration 0 counts", sg[0]["n_surf"], sg[0]["n_corner"])
rg = O.gn_iterations(maps[0], maps[1], feats[0], feats[1], p0, prm, 1)
print("oracle gn iteration 0 counts", rg["iters"][0]["n_surf"], rg["iters"][0]["n_corner"])
for lanes in (8, 16):
    os.environ["MLH_KNN_LANES"] = str(lanes)
    c2 = mla.Context(0)
    del os.environ["MLH_KNN_LANES"]
    c2.map_set(mla.SURF, case["surf_map"], min_match_sq_dis=msd); c2.map_set(mla.CORNER, case["corner_map"], min_match_sq_dis=msd)
    c2.features_set(mla.SURF, feats[0]); c2.features_set(mla.CORNER, feats[1])
    p_, s_ = c2.scan2map(p0, opts)
    g_ = c2.match_linearize(mla.CORNER, p0, flags=mla.FLAG_CHECK_FOV, min_match_sq_dis=msd, huber_delta=huber)
    print("lanes", lanes, "scan2map outer0 counts", s_[0]["n_surf"], s_[0]["n_corner"], "match_linearize corner", int(g_["valid"].sum()))
    c2.close()
