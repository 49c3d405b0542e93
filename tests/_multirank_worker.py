"""Worker of tests/test_gpu_multirank.py: one process per GPU (torch.distributed.run), a real RCCL communicator with WORLD_SIZE ranks.
Every rank stages its share of the 50k case (map wedge + halo and ownership by position, or the whole map and round-robin ownership),
runs the device-resident sharded Gauss-Newton solve -- per iteration: correspondence kernel + fit kernel (local reduce) + ONE
ncclAllReduce of 32 f64 on the context's stream + the redundant solve -- and rank 0 checks the pose against the unsharded solve of a
second, communicator-less context on its own GPU. Prints one JSON line (rank 0)."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    mode = sys.argv[1] if len(sys.argv) > 1 else "map"
    comm = sys.argv[2] if len(sys.argv) > 2 else "rccl"          # "rccl": one GPU per rank; "p2p": the mailbox communicator, ranks may share a GPU
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if comm == "p2p":
        # more ranks than GPUs: they share (RCCL refuses that; the mailboxes do not care). MLH_P2P_SHARE_GPU=1: all ranks on GPU 0 whatever the box has
        local_rank = 0 if os.environ.get("MLH_P2P_SHARE_GPU") == "1" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl" if comm == "rccl" else "gloo", rank=rank, world_size=world)
    mla = importlib.import_module("m-loam_amd")
    synth = importlib.import_module("m-loam_amd.synth")
    shard = importlib.import_module("m-loam_amd.shard")
    import conftest
    case = conftest._make_case(synth, "50k", 16, 1)
    ctx = mla.Context(local_rank)
    feats = conftest.features_from_extraction(synth, case["scans"], lambda s: ctx.extract(s.points, s.scan_start, s.scan_end))
    p0 = case["p0"]
    centre = p0[:2]
    if comm == "rccl":
        uid = [mla.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
    if mode == "map":
        ctx.shard_set(*shard.wedge_planes(centre, world, rank))
        far = np.full((1, 3), 1.0e6, np.float32)
        ms = np.ascontiguousarray(case["surf_map"][shard.shard_points_mask(case["surf_map"], centre, world, rank)])
        mc = np.ascontiguousarray(case["corner_map"][shard.shard_points_mask(case["corner_map"], centre, world, rank)])
        ctx.map_set_pair(ms if len(ms) else far, mc if len(mc) else far)
    else:
        ctx.shard_set_features(world, rank)
        ctx.map_set_pair(case["surf_map"], case["corner_map"])
    if comm == "rccl":
        ctx.comm_init(world, rank, uid[0])
    else:
        handles = [None] * world
        dist.all_gather_object(handles, ctx.p2p_mailbox())       # the out-of-band exchange: 64 bytes per rank
        ctx.p2p_comm_init(world, rank, handles)
        dist.barrier()                                           # every rank has mapped every mailbox before the first record is sent
    ones = ctx.allreduce_f64(np.ones(32))                       # the communicator really spans `world` ranks
    if len(sys.argv) > 3 and sys.argv[3] == "timeout":
        # a peer that never arrives: rank 0 solves alone, the others do not. The exchange inside the fit kernel's finish gives up after its bound (5 s), nothing is
        # solved on the partial sums, and the call REPORTS it (it used to return a pose built from stale slots, the error surfacing in a later call)
        out = None
        if rank == 0:
            ctx.features_set(mla.SURF, feats[0]); ctx.features_set(mla.CORNER, feats[1])
            try:
                ctx.gn_solve(p0, 2, want_stats=False)
                out = dict(raised=False)
            except mla.MlhError as e:
                out = dict(raised=True, message=str(e))
            try:                                                 # the error word was consumed: it does not resurface in an unrelated call
                ctx.map_rebuild(mla.ALL_KINDS); ctx.synchronize()
                out["later_call_ok"] = True
            except mla.MlhError as e:
                out["later_call_ok"] = False; out["later_message"] = str(e)
        dist.barrier()
        ctx.close()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return
    ctx.features_set(mla.SURF, feats[0])
    ctx.features_set(mla.CORNER, feats[1])
    pose, stats = ctx.gn_solve(p0, 5)
    pose_s2m, _ = ctx.scan2map(p0)
    split_equal = None
    if comm == "p2p":                                            # split submission works under the mailbox communicator: the exchange is inside the launches
        ctx.gn_solve_begin(p0, 5)
        split_equal = bool(np.array_equal(ctx.gn_solve_end(), ctx.gn_solve(p0, 5, want_stats=False)[0]))
    # the collective's own time where it is a launch of its own (RCCL: HIP events around it on the context's stream; waiting for the slowest peer is inside), and
    # the sharded solve's wall time per call either way (with the mailbox communicator the exchange happens inside the fit kernel's finish: no separate launch)
    import time
    ctx.profile_enable(1 << mla.K_ALLREDUCE); ctx.profile_reset()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.gn_solve(p0, 5, want_stats=False)
    solve_ms = 1e3 * (time.perf_counter() - t0) / 20
    ar_ms, ar_n = ctx.profile_get(mla.K_ALLREDUCE)
    ctx.profile_enable(0)
    out = None
    if rank == 0:
        one = mla.Context(local_rank)                           # the same frame, unsharded, no communicator
        one.map_set_pair(case["surf_map"], case["corner_map"])
        one.features_set(mla.SURF, feats[0]); one.features_set(mla.CORNER, feats[1])
        ref, ref_stats = one.gn_solve(p0, 5)
        ref_s2m, _ = one.scan2map(p0)
        one.close()
        out = dict(world=world, mode=mode, comm=comm, split_submission_equal=split_equal, allreduce_of_ones=float(ones[0]), pose_diff=float(np.abs(pose - ref).max()),
                   scan2map_pose_diff=float(np.abs(pose_s2m - ref_s2m).max()), allreduce_us=round(1e3 * ar_ms / max(ar_n, 1), 2), allreduce_launches=int(ar_n), gn_solve5_ms=round(solve_ms, 4),
                   counts=[(int(s["n_surf"]), int(s["n_corner"])) for s in stats],
                   counts_unsharded=[(int(s["n_surf"]), int(s["n_corner"])) for s in ref_stats])
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
