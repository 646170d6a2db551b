"""CPU tier: groundwork for carrying the escape state in the scan (no escape table, no second read of long backslash runs): the scheme
-- assume, publish per-segment elements, repair in the resolve step -- as a byte-level model (tests/host/test_escape_carry_model.cpp)
against a sequential scan that is itself anchored on the oracle, on adversarial documents of backslash runs of every length across
segments of 8 ... 64 bytes with look-backs of 2 ... 8.  Each of the scheme's repairs must be NECESSARY: leaving one out has to fail."""
import os
import subprocess

import pytest

from simdjson_amd import _paths


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("esc") / "test_escape_carry_model")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_escape_carry_model.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", _paths.ORACLE_DIR, src, os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"), "-lm", "-o", exe], check=True)
    return exe


def test_the_segmented_scan_without_a_table_equals_the_sequential_scan(model):
    for seed in (1, 2026):
        p = subprocess.run([model, str(seed), "150000"], capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[:2000]
        out = p.stdout.decode()
        assert "150000 documents" in out
        counts = [int(x) for x in __import__("re").findall(r"(\d+)[ ,)]", out.split("(", 1)[1])]
        assert all(c > 500 for c in counts), out  # every case and every repair occurred


@pytest.mark.parametrize("repair", [1, 2, 3, 4])
def test_every_repair_is_necessary(model, repair):
    p = subprocess.run([model, "1", "150000", str(repair)], capture_output=True)
    assert p.returncode != 0, "the model passes without repair %d: the test has no teeth" % repair
