"""bench.py honours the driver's contract: exactly one JSON line on stdout with the required keys and types."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-seconds", "1.0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0.0 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert d["value"] > 20 * cb["value"]       # the north star asks for >= 10x the CPU path at 1 GPU
    # value = valid correspondences (residual + Jacobian produced and reduced) per second, summed over the iterations of a step
    n_valid = sum(a + b for a, b in d["config"]["n_valid_per_iter_surf_corner"])
    assert n_valid == d["valid_correspondences_per_step"] and len(d["config"]["n_valid_per_iter_surf_corner"]) == d["config"]["gn_iters_per_step"]
    assert abs(d["value"] - n_valid / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    n_q = (d["config"]["features_surf"] + d["config"]["features_corner"]) * d["config"]["gn_iters_per_step"]
    assert abs(d["queries_per_s"] - n_q / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["queries_per_s"] and n_valid <= n_q
    # the corner half of the workload is real: >= 30 % of the corner queries end in a linearised correspondence
    assert min(b for _, b in d["config"]["n_valid_per_iter_surf_corner"]) >= 0.3 * d["config"]["features_corner"]
    # round 3: what binds is part of the line -- the fraction on unavoidable bytes, the binding resource, and the time-dominant kernel's own roofline
    assert rf["binding"] in ("latency", "valu", "hbm") and 0.0 < rf["unavoidable_frac"] < rf["frac"]
    rt = d["roofline_time_dominant_kernel"]
    assert rt["bound"] == "hbm" and rt["peak"] == 8000.0 and abs(rt["frac"] - rt["achieved"] / rt["peak"]) < 1e-4 and rt["avg_kernel_us"] > 0
    assert d["gpu_clock_spinup_ms"] >= 0 and d["ms_per_step_map_outgrows_its_grid_box"] > 0
    # frames are submitted pipelined by default (pose k collected after frame k + 1's staging is enqueued); the synchronous loop is reported beside it
    assert d["frame_submission"].startswith("pipelined") and d["ms_per_step_synchronous_submission"] > 0
