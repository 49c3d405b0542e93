"""Randomised parity soak, HIP path against the oracle (not part of the test suite): random scenes (seed, 16 / 32 rings, 1 or 2 LiDARs), random start-pose errors
from centimetres to 1.5 m / 6 degrees (so that many features sit at the acceptance gates), random options (N_NEIGH 5 / 10, CHECK_FOV, Huber deltas,
match radii). Per trial: the validity flags and the f32 coefficients of both kinds bit for bit (a single decision flip fails), residuals / Jacobians / normal
equations to 1e-9, five Gauss-Newton iterations (counts per iteration equal, pose 1e-7), scan2MapOptimization (LM iteration counts, terminations, pose).
usage: python scripts/soak_parity.py [trials] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest, oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O.build()
ctx = mla.Context(0)
t0 = time.time()
tot = dict(features=0, valid=0, gate_close=0, lm=0)


def pose_err(a, b):
    dt = float(np.linalg.norm(a[:3] - b[:3]))
    # (rotation: 2 |q_a -+ q_b| -- the arccos of a dot product one ulp below 1 already reads 3e-8)
    return dt, 2 * min(float(np.linalg.norm(a[3:] - b[3:])), float(np.linalg.norm(a[3:] + b[3:])))


for trial in range(trials):
    n_rings = int(rng.choice([16, 32]))
    n_lidars = int(rng.choice([1, 2]))
    sseed = int(rng.integers(1, 10 ** 6))
    case = conftest._make_case(synth, "50k", n_rings, n_lidars, seed=sseed)
    feats = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
    # (up to 1.5 m / 6 deg: starts from which Levenberg-Marquardt needs tens of iterations with rejected steps -- the bookkeeping must still be the oracle's)
    dt_mag, dr_mag = float(rng.choice([0.02, 0.1, 0.3, 0.5, 1.0, 1.5])), float(rng.choice([0.2, 1.0, 3.0, 6.0]))
    p0 = synth.perturbed_pose(case["gt"], seed=sseed + 1, dt=dt_mag, drot_deg=dr_mag)
    k_neigh = int(rng.choice([5, 10]))
    fov = bool(rng.integers(2))
    huber = float(rng.choice([0.1, 1.0]))
    msd = float(rng.choice([1.0, 0.64]))
    what = f"trial {trial}: scene {sseed}, {n_lidars} x {n_rings} rings, start error {dt_mag} m / {dr_mag} deg, N_NEIGH {k_neigh}, fov {fov}, huber {huber}, min_match_sq_dis {msd}"
    ctx.map_set(mla.SURF, case["surf_map"], min_match_sq_dis=msd); ctx.map_set(mla.CORNER, case["corner_map"], min_match_sq_dis=msd)
    ctx.features_set(mla.SURF, feats[0]); ctx.features_set(mla.CORNER, feats[1])
    maps = (O.Map(case["surf_map"]), O.Map(case["corner_map"]))
    prm = O.mapper_params(huber_delta=huber, n_neigh=k_neigh, check_fov=fov, min_match_sq_dis=msd)
    for kind, ch in ((mla.SURF, "s"), (mla.CORNER, "c")):
        got = ctx.match_linearize(kind, p0, flags=(mla.FLAG_CHECK_FOV if fov else 0), min_match_sq_dis=msd, huber_delta=huber, k_neigh=k_neigh)
        valid, coeffs = maps[kind].match(ch, feats[kind], p0, n_neigh=k_neigh, check_fov=fov, min_match_sq_dis=msd)
        if not np.array_equal(got["valid"], valid):
            w = np.nonzero(got["valid"] != valid)[0]
            raise SystemExit(f"DECISION FLIP {what}: kind {ch}, {len(w)} features, first {w[:5]}")
        if not np.array_equal(got["coeffs"].astype(np.float32).view(np.uint32), coeffs.astype(np.float32).view(np.uint32)):
            raise SystemExit(f"COEFFICIENT BITS {what}: kind {ch}")
        ref = O.linearize(ch, feats[kind], np.full(len(feats[kind]), 0.0075), p0, valid, coeffs, huber)
        if True:
            for k_, tol in (("r", 1e-9), ("J", 1e-9), ("H", 1e-9), ("g", 1e-8)):
                sc = max(1e-12, float(np.abs(ref[k_]).max()))
                if float(np.abs(got[k_] - ref[k_]).max()) > tol * max(sc, 1.0) + 1e-9 * sc:
                    raise SystemExit(f"{k_} {what}: kind {ch}, max |d| {np.abs(got[k_] - ref[k_]).max():.3e} of {sc:.3e}")
        tot["features"] += len(valid); tot["valid"] += int(valid.sum())
    opts = mla.default_opts(flags=(mla.FLAG_CHECK_FOV if fov else 0), huber_delta=huber, min_match_sq_dis=msd)
    if k_neigh == 5:
        pose, stats = ctx.gn_solve(p0, 5, opts)
        ref = O.gn_iterations(maps[0], maps[1], feats[0], feats[1], p0, prm, 5)
        for it, (s, r) in enumerate(zip(stats, ref["iters"])):
            if (s["n_surf"], s["n_corner"]) != (r["n_surf"], r["n_corner"]):
                raise SystemExit(f"GN COUNTS {what}: iteration {it}: {(s['n_surf'], s['n_corner'])} vs {(r['n_surf'], r['n_corner'])}")
            if s["is_degenerate"] != r["is_degenerate"]:
                raise SystemExit(f"GN DEGENERACY {what}: iteration {it}")
        dt, dr = pose_err(pose, ref["pose"])
        if dt > 1e-7 or dr > 1e-7:
            raise SystemExit(f"GN POSE {what}: {dt:.2e} m {dr:.2e} rad")
        fast = ctx.gn_solve(p0, 5, opts, want_stats=False)[0]
        if not np.array_equal(fast, pose):
            raise SystemExit(f"GN SCHEDULE {what}: the deferred schedule differs from the statistics path")
        pose2, st2 = ctx.scan2map(p0, opts)
        ref2 = O.scan2map(maps[0], maps[1], feats[0], feats[1], p0, prm)
        for o_, (s, r) in enumerate(zip(st2, ref2["outer"])):
            if (s["lm_iterations"], s["successful_steps"], s["termination"], s["is_degenerate"]) != (r["lm_iterations"], r["successful_steps"], r["termination"], r["is_degenerate"]):
                raise SystemExit(f"LM {what}: outer {o_}: {(s['lm_iterations'], s['successful_steps'], s['termination'])} vs {(r['lm_iterations'], r['successful_steps'], r['termination'])}")
            tot["lm"] += s["lm_iterations"]
        dt, dr = pose_err(pose2, ref2["pose"])
        if dt > 1e-7 or dr > 1e-7:
            raise SystemExit(f"SCAN2MAP POSE {what}: {dt:.2e} m {dr:.2e} rad")
        # the mapper's other modes: uncertainty-weighted residuals on features that carry covariances, with a good-feature selection at a random ratio
        gm = str(rng.choice(["wo_gf", "gd_fix", "rnd", "fps"]))
        gr = 1.0 if gm == "wo_gf" else float(rng.choice([0.2, 0.4, 0.9]))
        gseed = int(rng.integers(1, 1000))

        def f11(f):
            out = np.zeros((len(f), 11), np.float32); out[:, :4] = f[:, :4]
            d = rng.uniform(0.2, 1.0, (len(f), 3)) * 0.01
            out[:, 4] = d[:, 0]; out[:, 7] = d[:, 1]; out[:, 9] = d[:, 2]; out[:, 10] = out[:, 4] + out[:, 7] + out[:, 9]
            return out
        fs11, fc11 = f11(feats[0]), f11(feats[1])
        ctx.features_set(mla.SURF, fs11); ctx.features_set(mla.CORNER, fc11)
        o3 = mla.default_opts(flags=(mla.FLAG_CHECK_FOV if fov else 0) | mla.FLAG_WITH_UA, huber_delta=huber, min_match_sq_dis=msd, gf_method=mla.GF_METHODS[gm], gf_ratio=gr, gf_seed=gseed)
        pose3, st3 = ctx.scan2map(p0, o3)
        ref3 = O.scan2map(maps[0], maps[1], fs11, fc11, p0, O.mapper_params(huber_delta=huber, n_neigh=k_neigh, check_fov=fov, min_match_sq_dis=msd, with_ua=True, gf_method=gm, gf_ratio=gr, seed=gseed))
        for o_, (s, r) in enumerate(zip(st3, ref3["outer"])):
            if (s["n_surf"], s["n_corner"], s["lm_iterations"], s["termination"]) != (r["n_surf_sel"], r["n_corner_sel"], r["lm_iterations"], r["termination"]):
                raise SystemExit(f"SCAN2MAP with_ua {gm} {gr} seed {gseed} {what}: outer {o_}: {(s['n_surf'], s['n_corner'], s['lm_iterations'], s['termination'])} vs {(r['n_surf_sel'], r['n_corner_sel'], r['lm_iterations'], r['termination'])}")
            tot["lm"] += s["lm_iterations"]
        dt, dr = pose_err(pose3, ref3["pose"])
        if dt > 1e-7 or dr > 1e-7:
            raise SystemExit(f"SCAN2MAP with_ua {gm} {gr} POSE {what}: {dt:.2e} m {dr:.2e} rad")
    else:
        poses, st = ctx.gn_solve_blocks(np.array([p0]), 4, [10], [100.0], [0], opts)
        ref = O.gn_iterations(maps[0], maps[1], feats[0], feats[1], p0, prm, 4)
        for it in range(4):
            s, r = st[it][0], ref["iters"][it]
            if (s["n_surf"], s["n_corner"]) != (r["n_surf"], r["n_corner"]):
                raise SystemExit(f"GN COUNTS (N_NEIGH 10) {what}: iteration {it}")
        dt, dr = pose_err(poses[0], ref["pose"])
        if dt > 1e-7 or dr > 1e-7:
            raise SystemExit(f"GN POSE (N_NEIGH 10) {what}: {dt:.2e} m {dr:.2e} rad")
    print(f"  ok {what}", flush=True)
ctx.close()
print(f"parity soak: {trials} random trials, seed {seed}: {tot['features']} features matched ({tot['valid']} valid), 0 decision flips, coefficient bits equal, "
      f"{tot['lm']} LM iterations with equal counts and terminations, poses within 1e-7; {time.time() - t0:.0f} s")
