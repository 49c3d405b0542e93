"""Soak of the sharded solver over the mailbox communicator (ranks may share one GPU): every rank draws the SAME random call sequence -- Gauss-Newton solves with
1-6 iterations (with / without statistics, synchronous / split), scan2MapOptimization, sharded pose blocks, raw all-reduces, feature sets of changing size -- and
after every call (a) every rank must hold the same bits and (b) rank 0 compares with an unsharded context of its own (1e-9: the sums associate differently).
launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/soak_p2p.py [operations] [seed] [map|features]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "map"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dev = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
mla = importlib.import_module("m-loam_amd"); synth = importlib.import_module("m-loam_amd.synth"); shard = importlib.import_module("m-loam_amd.shard")
import conftest
case = conftest._make_case(synth, "50k", 16, 1)
ctx = mla.Context(dev)
feats = conftest.features_from_extraction(synth, case["scans"], lambda s: ctx.extract(s.points, s.scan_start, s.scan_end))
p0 = case["p0"]; centre = p0[:2]
if mode == "map":
    ctx.shard_set(*shard.wedge_planes(centre, world, rank))
    ms = np.ascontiguousarray(case["surf_map"][shard.shard_points_mask(case["surf_map"], centre, world, rank)])
    mc = np.ascontiguousarray(case["corner_map"][shard.shard_points_mask(case["corner_map"], centre, world, rank)])
    ctx.map_set_pair(ms, mc)
else:
    ctx.shard_set_features(world, rank)
    ctx.map_set_pair(case["surf_map"], case["corner_map"])
handles = [None] * world
dist.all_gather_object(handles, ctx.p2p_mailbox())
ctx.p2p_comm_init(world, rank, handles)
dist.barrier()
one = None
if rank == 0:
    one = mla.Context(dev)
    one.map_set_pair(case["surf_map"], case["corner_map"])
rng = np.random.default_rng(seed)            # the same on every rank: the same calls in the same order
fs, fc = feats
def set_feats(s, c):
    ctx.features_set(mla.SURF, s); ctx.features_set(mla.CORNER, c)
    if one is not None:
        one.features_set(mla.SURF, s); one.features_set(mla.CORNER, c)
set_feats(fs, fc)
t0 = time.time()
counts = {}
for it in range(n_ops):
    op = str(rng.choice(["gn", "gn", "gn_stats", "split", "s2m", "allreduce", "feat", "blocks"]))
    counts[op] = counts.get(op, 0) + 1
    got = ref = None
    if op in ("gn", "gn_stats"):
        n = int(rng.integers(1, 7))
        got = ctx.gn_solve(p0, n, want_stats=(op == "gn_stats"))[0]
        if one is not None: ref = one.gn_solve(p0, n, want_stats=False)[0]
    elif op == "split":
        n = int(rng.integers(1, 7))
        ctx.gn_solve_begin(p0, n); got = ctx.gn_solve_end()
        if one is not None: ref = one.gn_solve(p0, n, want_stats=False)[0]
    elif op == "s2m":
        got = ctx.scan2map(p0, want_stats=False)[0]
        if one is not None: ref = one.scan2map(p0, want_stats=False)[0]
    elif op == "allreduce":
        v = rng.normal(size=int(rng.integers(1, 200)))
        got = ctx.allreduce_f64(v * (rank + 1))
        if one is not None: ref = v * (world * (world + 1) / 2)
    elif op == "feat":
        k = int(rng.integers(1, 6))
        s = np.ascontiguousarray(np.tile(fs, (k, 1))[: int(rng.integers(400, len(fs) * k + 1))])
        c = np.ascontiguousarray(np.tile(fc, (k, 1))[: int(rng.integers(60, len(fc) * k + 1))])
        set_feats(s, c)
        continue
    elif op == "blocks":
        if mode != "map":
            continue
        half, hc = len(fs) // 2, len(fc) // 2
        sb, cb = [fs[:half], fs[half:]], [fc[:hc], fc[hc:]]
        args = (np.array([p0, p0]), int(rng.integers(1, 5)), [5, 10], [100.0, 70.0], [0, 1])
        ctx.features_set_blocks(mla.SURF, sb); ctx.features_set_blocks(mla.CORNER, cb)
        got = ctx.gn_solve_blocks(*args, want_stats=False)[0]
        if one is not None:
            one.features_set_blocks(mla.SURF, sb); one.features_set_blocks(mla.CORNER, cb)
            ref = one.gn_solve_blocks(*args, want_stats=False)[0]
        set_feats(fs, fc)
    got = np.ascontiguousarray(got, np.float64)
    all_got = [None] * world
    dist.all_gather_object(all_got, got.tobytes())
    if any(b != all_got[0] for b in all_got):
        raise SystemExit(f"[rank {rank}] RANKS DISAGREE after {it} operations in `{op}`")
    if one is not None:
        err = float(np.abs(got - np.asarray(ref)).max())
        if not err <= 1e-9 * max(1.0, float(np.abs(np.asarray(ref)).max())):
            raise SystemExit(f"SHARDED != UNSHARDED after {it} operations in `{op}`: max |d| {err:.3e}")
dist.barrier()
ctx.close()
if one is not None:
    one.close()
    print(f"mailbox soak: {world} ranks ({'sharing one GPU' if world > torch.cuda.device_count() else 'a GPU each'}), mode {mode}, seed {seed}: {n_ops} operations {counts}: every rank the same bits after every call, "
          f"rank 0 within 1e-9 of its unsharded context  [{time.time() - t0:.0f} s]")
dist.destroy_process_group()
