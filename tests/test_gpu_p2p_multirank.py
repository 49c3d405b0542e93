"""Multi-rank execution on whatever GPUs there are -- ONE is enough: the mailbox communicator (mlh_p2p_mailbox / mlh_p2p_comm_init) lets ranks share a device,
so the sharded solver runs as 2, 3 and 4 real processes with real inter-process exchange (hipIpc-mapped mailboxes, one kernel per all-reduce) on the single GPU of
the test box. Each rank stages its share (map wedge + halo, or the whole map with round-robin feature ownership), every evaluation's packed normal equations are
all-reduced through the mailboxes, every rank applies the same update; rank 0 compares with the unsharded solve. (tests/test_gpu_multirank.py is the same worker
over RCCL, which needs one GPU per rank.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_multirank_worker.py")
_PORT = [29651]


def _run(world, mode, share_gpu=True, extra=()):
    _PORT[0] += 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MLH_P2P_SHARE_GPU="1" if share_gpu else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_PORT[0]),
           WORKER, mode, "p2p", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-1500:]
    return json.loads(lines[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "map"), (2, "features"), (3, "features"), (4, "map")])
def test_sharded_solver_over_the_mailbox_communicator(world, mode):
    out = _run(world, mode)
    assert out["world"] == world and out["comm"] == "p2p"
    assert out["allreduce_of_ones"] == float(world)                    # the communicator really spans `world` ranks
    assert out["counts"] == out["counts_unsharded"]                    # same matched features every iteration ...
    assert out["pose_diff"] < 1e-9 and out["scan2map_pose_diff"] < 1e-9      # ... same pose (the records are summed in another order: not bit for bit)
    assert out["split_submission_equal"] is True                      # mlh_gn_solve_begin / _end under the communicator: the same bits as mlh_gn_solve


@pytest.mark.gpu
def test_a_peer_that_never_arrives_is_reported_by_the_solve_itself():
    """with the mailbox communicator a solve whose peer timed out must fail in THAT call, with a message that names the cause"""
    out = _run(2, "features", extra=("timeout",))
    assert out["raised"] is True and "peer rank did not arrive" in out["message"], out
    assert out["later_call_ok"] is True, out


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="one rank per GPU over the mailbox communicator needs two GPUs (peer stores over xGMI)")
@pytest.mark.parametrize("mode", ["map", "features"])
def test_mailbox_communicator_between_gpus(mode):
    """the same with a GPU per rank: the mailboxes are peer-mapped device memory of OTHER GPUs (runs by itself where there are two)"""
    world = min(_n_gpus(), 4)
    out = _run(world, mode, share_gpu=False)
    assert out["allreduce_of_ones"] == float(world)
    assert out["counts"] == out["counts_unsharded"]
    assert out["pose_diff"] < 1e-9 and out["scan2map_pose_diff"] < 1e-9
    assert out["split_submission_equal"] is True
