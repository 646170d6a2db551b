"""CPU tier: the C oracle (oracle/sj_oracle.c) against the committed golden fixtures that
tests/golden/make_golden.py produced from the real reference.  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

import checkers
from simdjson_amd import corpus

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


@pytest.fixture(scope="module")
def orc():
    return checkers.Oracle()


def test_utf8_known_answer_vectors(orc):
    v = load("utf8_vectors.json")
    assert len(v["good"]) == 8 and len(v["bad"]) == 29
    for g in v["good"]:
        assert orc.validate_utf8(bytes.fromhex(g)), g
    for b in v["bad"]:
        assert not orc.validate_utf8(bytes.fromhex(b)), b


def test_validate_shapes_of_reference_basictests(orc):
    # tests/dom/basictests.cpp:1814-1909: 0..128 spaces are valid; a lone 0xFF anywhere is not;
    # F0 8F BF BF is an overlong 4-byte sequence.
    for n in range(0, 129):
        assert orc.validate_utf8(b" " * n)
        for off in range(n):
            a = bytearray(b" " * n)
            a[off] = 0xFF
            assert not orc.validate_utf8(bytes(a))
    assert not orc.validate_utf8(b"\xf0\x8f\xbf\xbf")


def test_small_cases_all_modes(orc):
    g = load("small_cases.json")
    for case in g["cases"]:
        data = bytes.fromhex(case["hex"])
        for mname, mode in checkers.MODES.items():
            obs = checkers.observable(data, mode, *orc.stage1(data, mode))
            want = case["stage1"][mname]
            got = {"err": obs[0]} if len(obs) == 1 else {"err": obs[0], "n": obs[1], "idx": list(obs[2])}
            assert got == want, (data, mname)
        merr, mout = orc.minify(data)
        assert merr == case["minify"]["err"] and bytes(mout).hex() == case["minify"]["hex"], data
        assert orc.validate_utf8(data) == case["utf8"], data


def make_corpus(name):
    kind, *args = name.split(":")
    if kind in ("large_random", "amazon_ndjson", "twitter_like"):
        return getattr(corpus, kind)(int(args[0]), int(args[1]))[0]
    if kind == "deep_nesting":
        return corpus.deep_nesting(int(args[0]))
    if kind == "backslash_runs":
        runs = [1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 16383, 16384, 16385, (1 << 20) - 1, 1 << 20, (1 << 20) + 1]
        return corpus.backslash_runs(runs, int(args[0]))
    if kind.startswith("jsonexamples/"):  # committed fixtures (tests/golden/jsonexamples/README.md)
        return np.fromfile(os.path.join(GOLD, kind), dtype=np.uint8)
    raise ValueError(name)


def test_corpora_digests(orc):
    g = load("corpora.json")
    checked = 0
    for d in g["corpora"]:
        if d["len"] > (20 << 20):
            continue  # the 100 MiB cases are checked on the GPU tier
        a = make_corpus(d["name"])
        assert len(a) == d["len"] and orc.fnv(a) == d["buf_fnv"], d["name"]
        err, n, idx = orc.stage1(a, 0)
        assert (err, n) == (d["stage1_err"], d["n"]), d["name"]
        assert orc.fnv(idx) == d["idx_fnv"], d["name"]
        merr, mout = orc.minify(a)
        assert (merr, len(mout), orc.fnv(mout)) == (d["minify_err"], d["minify_len"], d["minify_fnv"]), d["name"]
        assert orc.validate_utf8(a) == d["utf8"], d["name"]
        checked += 1
    assert checked >= 20


def random_case_stream(seed):
    rng = np.random.default_rng(seed)
    for it in range(400):
        n = int(rng.integers(0, 600))
        yield corpus.random_adversarial(n, int(rng.integers(0, 1 << 31)), ascii_only=bool(it % 3 == 0),
                                        p_backslash=0.15 if it % 4 == 0 else 0.0)


def digest_results(orc, stage1, minify, utf8, seed):
    h = []
    for a in random_case_stream(seed):
        for mode in range(7):
            obs = checkers.observable(a, mode, *stage1(a, mode))
            h.append(np.array([obs[0]], np.uint32))
            if len(obs) > 1:
                h.append(np.array([obs[1]], np.uint32))
                h.append(np.array(obs[2], np.uint32))
        merr, mout = minify(a)
        h.append(np.array([merr, len(mout)], np.uint32))
        h.append(mout.astype(np.uint32))
        h.append(np.array([int(utf8(a))], np.uint32))
    return orc.fnv(np.concatenate(h))


def test_random_adversarial_digests(orc):
    g = load("random_digest.json")
    for d in g["digests"]:
        assert digest_results(orc, orc.stage1, orc.minify, orc.validate_utf8, d["seed"]) == d["fnv"], d["seed"]


def test_strings_known_answers(orc):
    """the oracle's parse_string / string buffer against what the reference's x86 kernel produced
    (tests/golden/make_strings_golden.py): 4 356 string bodies x allow_replacement, and document::string_buf of the fixtures."""
    g = load("strings.json")
    for body_hex, allow, want in g["vectors"]:
        got = orc.parse_string(bytes.fromhex(body_hex), bool(allow))
        assert (None if got is None else got.hex()) == want, (body_hex, allow)
    for name, want in g["buffers"].items():
        data = np.frombuffer(open(os.path.join(GOLD, "jsonexamples", name), "rb").read(), dtype=np.uint8)
        err, n, idx = orc.stage1(data, 0)
        assert err == 0
        e2, buf, off, strings, bad = orc.string_buffer(data, idx, n)
        assert (e2, strings, len(buf), orc.fnv(buf)) == (0, want["strings"], want["bytes"], want["fnv1a64"]), name
