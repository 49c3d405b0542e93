"""Soak of the Gauss-Newton schedule's state machine (deferred finish, bounded search, final-in-successor, split scan2map): two contexts get the SAME random
sequence of calls -- one with every switch on, one with the classic schedule -- and every pose that comes back must be the same bits. Operations: synchronous
solves (with / without statistics, 1..6 iterations), submit / chained submit / collect in every legal interleaving (up to two in flight), scan2map (synchronous and
split), feature sets of changing size between frames (more tiles than the record buffer holds, fewer), map re-staging (plain and overlapped), pose-block solves.
usage: python scripts/soak_schedule.py [seconds] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest, oracle as O
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O.build()
case = conftest._make_case(synth, "50k", 16, 1)
feats = conftest.features_from_extraction(synth, case["scans"], lambda s: O.extract(s.points, s.scan_start, s.scan_end))
p0 = case["p0"]
ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
ctxs = [mla.Context(0), mla.Context(0)]
ctxs[0].set_gn_schedule(1, 1, 1)
ctxs[1].set_gn_schedule(0, 0, 0)
for c in ctxs:
    c.map_set_pair(case["surf_map"], case["corner_map"])
    c.features_set(mla.SURF, feats[0]); c.features_set(mla.CORNER, feats[1])
in_flight = []          # kinds of the solves in flight ("gn" / "s2m"), oldest first
n_ops = n_cmp = 0
submitted_once = False
t0 = time.time()
# scan2map's Levenberg-Marquardt loop (round 5): context 0 runs it as ONE launch per outer iteration (an explicit look-ahead: consumer-side launches), context 1 as
# round 4's launches (the step in the last workgroup). The switches are read at every call. `same_lm`: a submission with lm_lookahead = 0, where the one-launch loop
# cannot overflow and the launches can -- both contexts then take the default, so that they keep agreeing on what a frame's status is.
LM_ENV = ({"MLH_LM_CONSUMER": "1", "MLH_LM_LOOP": "1"}, {"MLH_LM_CONSUMER": "0", "MLH_LM_LOOP": "0"})
def both(fn, same_lm=False):
    out = []
    for i, c in enumerate(ctxs):
        os.environ.update(LM_ENV[0] if same_lm else LM_ENV[i])
        try:
            out.append(fn(c))
        except Exception as ex:
            raise SystemExit(f"ERROR in context {i} ({'round-4 schedule' if i == 0 else 'classic schedule'}) after {n_ops} operations: {ex}\nlast operations: " + " | ".join(trace[-40:]))
    return out
trace = []
def check(a, b, what):
    global n_cmp
    n_cmp += 1
    if not np.array_equal(np.asarray(a), np.asarray(b)):
        raise SystemExit(f"MISMATCH after {n_ops} operations in `{what}`: {a} vs {b}\nlast operations: " + " | ".join(trace[-40:]))
while time.time() - t0 < budget:
    n_ops += 1
    ops = ["sync", "sync_stats", "s2m", "feat", "map"]
    if len(in_flight) < 2:
        ops += ["begin", "begin", "s2m_begin"]
        if submitted_once:        # (the first frame of a context is submitted with its pose: include/mloam_hip.h)
            ops += ["chained", "chained", "chained", "s2m_chained"]
    if in_flight:
        ops += ["end", "end", "end"]
    if not in_flight:
        ops += ["blocks", "blocks"]
    op = ops[rng.integers(len(ops))]
    trace.append(op)
    if op in ("begin", "s2m_begin"):
        submitted_once = True
    if op in ("sync", "sync_stats"):
        n = int(rng.integers(1, 7))
        trace[-1] += f"({n})"
        r = both(lambda c: c.gn_solve(p0, n, want_stats=(op == "sync_stats"))[0])
        check(r[0], r[1], op)
    elif op == "s2m":
        r = both(lambda c: c.scan2map(p0, want_stats=False)[0])
        check(r[0], r[1], op)
    elif op == "begin":
        n = int(rng.integers(1, 7))
        both(lambda c: c.gn_solve_begin(p0, n)); in_flight.append("gn"); trace[-1] += f"({n})"
    elif op == "chained":
        n = int(rng.integers(1, 7))
        both(lambda c: c.gn_solve_begin_chained(ident, ident, n)); in_flight.append("gn"); trace[-1] += f"({n})"
    elif op == "s2m_begin":
        la = int(rng.integers(0, 8))
        both(lambda c: c.scan2map_begin(p0, lm_lookahead=la), same_lm=(la == 0)); in_flight.append("s2m"); trace[-1] += f"({la})"
    elif op == "s2m_chained":
        la = int(rng.integers(0, 8))
        both(lambda c: c.scan2map_begin_chained(ident, ident, lm_lookahead=la), same_lm=(la == 0)); in_flight.append("s2m"); trace[-1] += f"({la})"
    elif op == "end":
        kind = in_flight.pop(0)
        trace[-1] += f"[{kind}]"
        if kind == "gn":
            r = both(lambda c: c.gn_solve_end())
            check(r[0], r[1], "gn_solve_end")
        else:
            r = both(lambda c: c.scan2map_end())
            # status 1 hands the START pose back (the caller re-solves): a younger solve behind it then starts from an unfinished pose in BOTH contexts alike
            check(r[0][0], r[1][0], "scan2map_end"); assert r[0][1] == r[1][1]
    elif op == "feat":
        # features change size between frames; half of the time with solves still in flight (their launches are already enqueued on the old features: stream
        # order protects them; a scan2map in flight that overflows its look-ahead then hands its frame back, status 1, in both contexts alike)
        while in_flight and rng.random() < 0.5:
            kind = in_flight.pop(0)
            r = both(lambda c: c.gn_solve_end() if kind == "gn" else c.scan2map_end()[0])
            check(r[0], r[1], "drain")
        k = int(rng.integers(1, 14))
        fs = np.ascontiguousarray(np.tile(feats[0], (k, 1))[: int(rng.integers(400, len(feats[0]) * k + 1))])
        fc = np.ascontiguousarray(np.tile(feats[1], (k, 1))[: int(rng.integers(60, len(feats[1]) * k + 1))])
        trace[-1] += f"({len(fs)},{len(fc)})"
        both(lambda c: (c.features_set(mla.SURF, fs), c.features_set(mla.CORNER, fc)))
    elif op == "map":
        trace[-1] += f"[{len(in_flight)} in flight]"
        if len(in_flight) <= 1 and rng.random() < 0.7:
            both(lambda c: c.map_set_pair_overlapped(case["surf_map"], case["corner_map"]) if in_flight else c.map_set_pair(case["surf_map"], case["corner_map"]))
        else:            # staged on the solver's own stream, behind whatever is in flight
            trace[-1] += "[own stream]"
            both(lambda c: c.map_set_pair(case["surf_map"], case["corner_map"]))
    elif op == "blocks":
        # 1..4 pose blocks of random sizes (a block may hold a handful of features, or no corner features at all), N_NEIGH 5 / 10 and freeze flags at random
        nb = int(rng.integers(1, 5))
        cut_s = np.sort(rng.integers(1, len(feats[0]), nb - 1)) if nb > 1 else np.array([], int)
        cut_c = np.sort(rng.integers(0, len(feats[1]) + 1, nb - 1)) if nb > 1 else np.array([], int)
        sb = [np.ascontiguousarray(x) for x in np.split(feats[0], cut_s)]
        cb = [np.ascontiguousarray(x) for x in np.split(feats[1], cut_c)]
        if any(len(x) == 0 for x in sb):
            continue
        both(lambda c: (c.features_set_blocks(mla.SURF, sb), c.features_set_blocks(mla.CORNER, cb)))
        n = int(rng.integers(1, 6))
        kk = [int(rng.choice([5, 10])) for _ in range(nb)]
        th = [float(rng.choice([100.0, 70.0, 1e9])) for _ in range(nb)]       # (1e9: every direction "degenerate")
        fz = [int(rng.integers(0, 2)) for _ in range(nb)]
        trace[-1] += f"(nb={nb},n={n},K={kk},sizes={[len(x) for x in sb]}/{[len(x) for x in cb]})"
        r = both(lambda c: c.gn_solve_blocks(np.array([p0] * nb), n, kk, th, fz, want_stats=False)[0])
        check(r[0], r[1], f"blocks nb={nb} K={kk} thre={th} freeze={fz} iters={n} sizes={[len(x) for x in sb]}/{[len(x) for x in cb]}")
        both(lambda c: (c.features_set(mla.SURF, feats[0]), c.features_set(mla.CORNER, feats[1])))
while in_flight:
    kind = in_flight.pop(0)
    r = both(lambda c: c.gn_solve_end() if kind == "gn" else c.scan2map_end()[0])
    check(r[0], r[1], "final drain")
for c in ctxs:
    c.close()
print(f"schedule soak: {n_ops} operations, {n_cmp} poses compared bit for bit between the current schedules (deferred GN finish, bounded search, final-in-successor; scan2map's LM loop "
      f"as one launch / consumer-side launches) and the classic ones, seed {seed}, {time.time() - t0:.0f} s: all equal")
