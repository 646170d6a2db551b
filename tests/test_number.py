"""CPU tier: the per-token number code of the tape kernels (simdjson_amd/csrc/sj_number.h, host + device functions) built with
g++ and compared with the reference's tape (its parse_number, Eisel-Lemire + from_chars fallback) and with the oracle (strtod)."""
import os
import re
import subprocess

import numpy as np
import pytest

import checkers
import jsongen
from simdjson_amd import _paths


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("number") / "test_number")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_number.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(_paths.PKG_DIR, "csrc"), src, "-o", exe], check=True)

    def run(texts):
        p = subprocess.run([exe], input="\n".join(texts) + "\n", capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        rows = [ln.split() for ln in p.stdout.strip().split("\n")]
        assert len(rows) == len(texts)
        slow = int(re.search(r"slow tokens: (\d+)", p.stderr).group(1))
        return [(int(e), t, int(b, 16)) for e, t, b in rows], slow
    return run


def _expected(parse, text):
    """(error, type, bits) of the number `text` from a dom parse of "[text]" (parse: data -> (err, tape, string_buf))"""
    err, tape, _ = parse(b"[" + text.encode() + b"]")
    if err:
        return (err, "-", 0)
    return (0, chr(int(tape[2]) >> 56), int(tape[3]))


def test_pow5_table_is_what_it_says():
    """entries of the generated table against exact integer arithmetic: top 128 bits of 5^q, truncated (q >= 0) / the reciprocal rounded up"""
    words = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", open(os.path.join(_paths.PKG_DIR, "csrc", "sj_pow5_table.inc")).read())]
    assert len(words) == 2 * 651
    for q in (-342, -341, -200, -28, -27, -26, -18, -1, 0, 1, 27, 28, 55, 56, 300, 308):
        v = (words[2 * (q + 342)] << 64) | words[2 * (q + 342) + 1]
        assert v >> 127 == 1
        if q >= 0:
            p = 5 ** q
            sh = p.bit_length() - 128
            assert v == (p >> sh if sh > 0 else p << -sh)
        else:
            p = 5 ** -q
            b = v.bit_length() + p.bit_length() - 1
            # v is within one unit of 2^b / p for the b that makes the quotient 128 bits long, never below it
            b = next(bb for bb in (b - 1, b, b + 1) if ((1 << bb) // p).bit_length() == 128)
            assert 0 <= v - (1 << b) // p <= 1


def test_numbers_like_the_oracle(driver):
    orc = checkers.Oracle()
    texts = [t for t in jsongen.number_corner_cases() if "\n" not in t]
    got, slow = driver(texts)
    assert slow >= 20  # the exact halfway cases do reach the big-integer decision
    for text, g in zip(texts, got):
        assert g == _expected(orc.dom_parse, text), text


@pytest.mark.skipif(not checkers.have_reference_lib(), reason="oracle/_ref/libsjref.so not built")
def test_numbers_like_the_reference(driver):
    ref = checkers.Reference()
    impl = ref.best_impl()
    texts = [t for t in jsongen.number_corner_cases() if "\n" not in t]
    rng = np.random.default_rng(31337)
    for _ in range(20000):  # doubles printed with 17 significant digits, and the same with digits appended / removed
        x = float(np.frombuffer(rng.bytes(8), dtype=np.float64)[0])
        if not np.isfinite(x):
            continue
        t = repr(abs(x))
        texts.append(t)
        texts.append(t.replace("e", "1234567890123e") if "e" in t else t + "1234567890123")
    got, _ = driver(texts)
    for text, g in zip(texts, got):
        assert g == _expected(lambda d: ref.dom_parse(impl, d), text), text
