"""CPU tier: the escape carry inside the scan (simdjson_amd/csrc/sj_xcarry.h) -- the product's own span_xword / xs_apply / xs_compact run
inside a byte-level model of the pipeline (tests/host/test_xcarry_model.cpp: spans that assume, summaries that carry x, groups, one
patched bit per span) against the oracle's sequential scan, on adversarial documents of backslash runs of every length across spans of
8 ... 64 bytes with look-backs of 2 ... 8.  Every rule must be NECESSARY: breaking one has to fail."""
import os
import re
import subprocess

import pytest

from simdjson_amd import _paths


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("xcarry") / "test_xcarry_model")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_xcarry_model.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", _paths.ORACLE_DIR, "-I", os.path.join(_paths.PKG_DIR, "csrc"), src,
                    os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"), "-lm", "-o", exe], check=True)
    return exe


def test_spans_that_assume_equal_the_sequential_scan(model):
    for seed in (1, 2026):
        p = subprocess.run([model, str(seed), "150000"], capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[:2000]
        out = p.stdout.decode()
        assert "150000 documents" in out
        counts = [int(x) for x in re.findall(r"(\d{2,})[ ,)]", out.split("(", 1)[1])]
        assert all(c > 200 for c in counts), out  # every kind, every repair, resolved spans included


@pytest.mark.parametrize("rule", [1, 2, 3, 4, 5])
def test_every_rule_is_necessary(model, rule):
    p = subprocess.run([model, "1", "150000", str(rule)], capture_output=True)
    assert p.returncode != 0, "the model passes without rule %d: the test has no teeth" % rule
