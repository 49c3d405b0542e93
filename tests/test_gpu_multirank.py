"""Multi-rank execution of the RCCL path: runs by itself the day two GPUs are visible, is skipped on a one-GPU box.
One process per GPU through torch.distributed.run (rendezvous on 127.0.0.1), a real ncclCommInitRank with n > 1."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:      # noqa: BLE001
        return 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(n, script, *args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           script, *args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


needs_two = pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs on this node")


@needs_two
@pytest.mark.parametrize("mode", ["map", "features"])
def test_sharded_solver_over_two_rccl_ranks(mode):
    """the device-resident sharded solve under a 2-rank RCCL communicator equals the unsharded solve: same matched counts in every iteration, pose to 1e-9"""
    r = _torchrun(2, os.path.join(ROOT, "tests", "_multirank_worker.py"), mode)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["world"] == 2 and d["allreduce_of_ones"] == 2.0          # RCCL saw two ranks
    assert d["counts"] == d["counts_unsharded"]
    assert d["pose_diff"] < 1e-9 and d["scan2map_pose_diff"] < 1e-9, d


@needs_two
def test_bench_two_ranks():
    """bench.py --gpus 2 as the driver launches it: one JSON line from rank 0, the communicator formed, the all-reduce measured"""
    r = _torchrun(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    mg = d["multi_gpu"]
    assert mg is not None and len(mg["owned_features_per_rank"]) == 2 and mg["allreduce_us_per_call_rank0"] is not None
    assert sum(a + b for a, b in mg["owned_features_per_rank"]) == d["config"]["features_surf"] + d["config"]["features_corner"]


def test_rccl_is_loadable_and_a_one_rank_communicator_forms():
    """what CAN run on one GPU: the library finds RCCL and a 1-rank communicator all-reduces on the context's stream (the N > 1 tests above are skipped here)"""
    import importlib
    import numpy as np
    mla = importlib.import_module("m-loam_amd")
    c = mla.Context(0)
    try:
        c.comm_init(1, 0, mla.comm_unique_id())
        assert np.array_equal(c.allreduce_f64(np.arange(32, dtype=np.float64)), np.arange(32, dtype=np.float64))
    finally:
        c.close()
