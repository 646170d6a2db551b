"""CPU tier: the kernels' per-lane math (simdjson_amd/csrc/sj_block.h) compiled for the host and
checked against byte-at-a-time definitions and the oracle scan (tests/host/test_block_math.cpp)."""
import os
import subprocess

from simdjson_amd import _paths


def test_block_math_host(tmp_path):
    exe = str(tmp_path / "test_block_math")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_block_math.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", _paths.CSRC_DIR, "-I", _paths.ORACLE_DIR, src,
                    os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
