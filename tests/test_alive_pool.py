"""The host-side container behind the good-feature draw loops and the segmenter's outlier erasure (m-loam_amd/csrc/alive_pool.hpp),
checked against the std::vector it replaces (tests/native/alive_pool_check.cpp). Compiled with g++: no GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("extra", [[], ["-U__SSE2__"]], ids=["sse2", "plain-loops"])   # the vector form and the portable one behind the same interface
def test_alive_pool_answers_as_the_vector_it_replaces(tmp_path, extra):
    exe = str(tmp_path / "alive_pool_check")
    subprocess.run(["g++", "-O2", "-std=c++17", *extra, "-I", os.path.join(ROOT, "m-loam_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "alive_pool_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    assert out.strip() == "ok", out
