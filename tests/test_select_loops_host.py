"""The host-side selection loops of goodFeatureMatching (m-loam_amd/csrc/select.hip) without a GPU: the translation unit is compiled host-side into the
harness scripts/exp/select_loop_bench.hip (the HIP symbols of its launch helpers stay unresolved and are never called) and run with --check:
  * the random loop over the survivor pool (csrc/alive_pool.hpp) draws and keeps exactly what the reference's loop over a std::vector it erases from does
    (same selections, same information matrix bits, same engine state afterwards);
  * the greedy loop ranked by j H^-1 j^T (determinant lemma, Sherman-Morrison) picks exactly what the literal Cholesky-logdet scoring picks, rows repeated
    verbatim (exact score ties) included.
180 combinations of size (7 .. 11197), matched fraction, ratio and seed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


@pytest.mark.skipif(HIPCC is None, reason="needs hipcc (the loops live in a .hip translation unit)")
def test_selection_loops_against_their_literal_forms(tmp_path):
    exe = str(tmp_path / "select_loop_check")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "m-loam_amd", "csrc"), os.path.join(ROOT, "scripts", "exp", "select_loop_bench.hip"), "-o", exe,
                    "-Wl,--unresolved-symbols=ignore-all"], check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe, "--check"], check=True, capture_output=True, text=True, timeout=300).stdout
    assert out.strip().startswith("ok "), out
