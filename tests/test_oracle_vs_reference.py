"""CPU tier, build container only: live differential of the C oracle against the REAL reference
kernels (oracle/_ref/libsjref.so), in the style of the reference's cross-implementation fuzzers
(/root/reference/fuzz/fuzz_minifyimpl.cpp:19-64, fuzz/fuzz_utf8.cpp:35-80)."""
import numpy as np
import pytest

import checkers
from simdjson_amd import corpus

pytestmark = pytest.mark.skipif(not checkers.have_reference_lib(), reason="oracle/_ref/libsjref.so not built")


@pytest.fixture(scope="module")
def both():
    return checkers.Oracle(), checkers.Reference()


def test_x86_kernels_agree_and_match_oracle(both):
    orc, ref = both
    impls = [i for i in ("icelake", "haswell", "westmere") if ref.available(i)]
    assert impls
    rng = np.random.default_rng(20260921)
    for it in range(3000):
        n = int(rng.integers(0, 400))
        a = corpus.random_adversarial(n, int(rng.integers(0, 1 << 31)), ascii_only=bool(it % 3 == 0),
                                      p_backslash=0.2 if it % 5 == 0 else 0.0)
        for mode in range(7):
            want = checkers.observable(a, mode, *orc.stage1(a, mode))
            for impl in impls:
                assert checkers.observable(a, mode, *ref.stage1(impl, a, mode)) == want, (bytes(a), mode, impl)
        om = orc.minify(a)
        ou = orc.validate_utf8(a)
        for impl in impls:
            rm = ref.minify(impl, a)
            assert rm[0] == om[0] and bytes(rm[1]) == bytes(om[1]), (bytes(a), impl)
            assert ref.validate_utf8(impl, a) == ou, (bytes(a), impl)


def test_capacity_guard(both):
    orc, ref = both
    impl = ref.best_impl()
    data = b"[1,2,3]"
    assert ref.stage1(impl, data, 0, capacity=3)[0] == checkers.CAPACITY
    assert orc.stage1(data, 0, capacity=3)[0] == checkers.CAPACITY


@pytest.mark.parametrize("kind", ["large_random", "amazon_ndjson", "twitter_like"])
def test_bulk_corpora(both, kind):
    orc, ref = both
    impl = ref.best_impl()
    a, _ = getattr(corpus, kind)(3 << 20, 77)
    for mode in (0, 2):
        assert checkers.observable(a, mode, *ref.stage1(impl, a, mode)) == checkers.observable(a, mode, *orc.stage1(a, mode))
    rm, om = ref.minify(impl, a), orc.minify(a)
    assert rm[0] == om[0] and np.array_equal(rm[1], om[1])
    # corrupt one byte at a time near block boundaries
    for pos in (63, 64, 65, 4095, 4096, 4097, len(a) - 1):
        for val in (0x22, 0x5C, 0xFF, 0xE2, 0x01):
            b = a.copy()
            b[pos] = val
            assert checkers.observable(b, 0, *ref.stage1(impl, b, 0)) == checkers.observable(b, 0, *orc.stage1(b, 0)), (pos, val)
            assert ref.validate_utf8(impl, b) == orc.validate_utf8(b)


def test_worst_case_generators_against_the_reference(both):
    """The shapes bench.py --workload deep_nesting / escape_heavy and the escape-table tests are built from: backslash
    runs of 16 KiB +- 1 and more at chosen distances from 16 KiB boundaries, one offset per byte, one long run."""
    orc, ref = both
    impls = [i for i in ("icelake", "haswell", "westmere") if ref.available(i)]
    rng = np.random.default_rng(99)
    SEG = 16384
    parts, at, k = [b"["], 1, 0
    lengths = [SEG - 1, SEG, SEG + 1, 2 * SEG - 1, 2 * SEG + 1, 3 * SEG, 100001, 63, 64, 65, 4095, 4097]
    while at < (3 << 20):
        r = lengths[k % len(lengths)]
        k += 1
        pad = (-(at + 1) + int(rng.choice([-2, -1, 0, 1, 2, 63, 64, 4095]))) % SEG
        body = b" " * pad + b'"' + b"\\" * r + (b'"' if r % 2 == 0 else b'""') + b', {"k": [1, 2]}, "x\\"y", '
        parts.append(body)
        at += len(body)
    parts.append(b"0]")
    docs = {"runs around segment boundaries": np.frombuffer(b"".join(parts), np.uint8),
            "escape_heavy": corpus.escape_heavy(2 << 20, 7)[0],
            "deep_nesting": corpus.deep_nesting_doc(1 << 20, 1)[0],
            "one long even run": np.frombuffer(b'["' + b"\\" * (1 << 20) + b'", 1]', np.uint8),
            "one long odd run": np.frombuffer(b'["' + b"\\" * ((1 << 20) + 1) + b'", 1]', np.uint8)}
    for name, a in docs.items():
        for impl in impls:
            for mode in (0, 2):
                assert checkers.observable(a, mode, *orc.stage1(a, mode)) == checkers.observable(a, mode, *ref.stage1(impl, a, mode)), \
                    (name, impl, mode)
            assert orc.validate_utf8(a) == ref.validate_utf8(impl, a), (name, impl)
            om, rm = orc.minify(a), ref.minify(impl, a)
            assert om[0] == rm[0] and np.array_equal(om[1], rm[1]), (name, impl)
