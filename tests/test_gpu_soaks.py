"""Bounded runs of the soak scripts (scripts/soak_*.py; the long runs and what they found: profiles/r04_soak.txt) so that the GPU suite itself exercises them:
random call sequences under both Gauss-Newton schedules, history-independence of the whole C-ABI on a long-lived context (and on four threads), the HIP path
against the oracle on random problems, the rows around the solver on random inputs, the mailbox communicator under random calls with ranks sharing the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    return r.stdout


def test_schedule_soak_bounded():
    out = _run([os.path.join(ROOT, "scripts", "soak_schedule.py"), "4", "101"])
    assert "all equal" in out


@pytest.mark.parametrize("threads", [1, 4])
def test_api_soak_bounded(threads):
    out = _run([os.path.join(ROOT, "scripts", "soak_api.py"), "4", "102", str(threads)])
    assert "every output equal to a fresh context's" in out


def test_parity_soak_bounded():
    out = _run([os.path.join(ROOT, "scripts", "soak_parity.py"), "6", "103"])
    assert "0 decision flips" in out


def test_frontend_parity_soak_bounded():
    out = _run([os.path.join(ROOT, "scripts", "soak_parity_frontend.py"), "4", "104"])
    assert "all equal" in out


HARD = {"MLOAM_SCENE_FAMILY": "hard"}       # synth.scene_family: poles, vegetation, grazing slabs, duplicated map points, a four-fold dense patch


def test_parity_soak_hard_scene_family():
    """HIP == oracle on the harder scenes: validity flags and coefficient bits, normal equations, GN counts, LM bookkeeping (profiles/r06_soak.txt: the long runs)"""
    out = _run([os.path.join(ROOT, "scripts", "soak_parity.py"), "6", "203"], env=HARD)
    assert "0 decision flips" in out


def test_frontend_parity_soak_hard_scene_family():
    out = _run([os.path.join(ROOT, "scripts", "soak_parity_frontend.py"), "3", "204", "track,segment,rough,select,odom_select,voxel"], env=HARD)
    assert "all equal" in out


def test_mailbox_soak_bounded():
    """three real processes sharing the one GPU (torch.distributed.run), 300 random calls"""
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(port),
                os.path.join(ROOT, "scripts", "soak_p2p.py"), "300", "105", "map"], env={"OMP_NUM_THREADS": "4"})
    assert "every rank the same bits after every call" in out
