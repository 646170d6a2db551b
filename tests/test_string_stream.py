"""CPU tier: the string buffer as a stream compaction of the document (simdjson_amd/csrc/sj_string_stream.h), block by block on the
host with the functions the kernels of sjgpu_string_stream.hip call (tests/host/test_string_stream.cpp), against the oracle's
string-at-a-time walk: byte for byte when every string is valid, flagged when the reference rejects one."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import jsongen
from simdjson_amd import _paths, corpus
from test_oracle_vs_reference import STRING_BODIES


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("strstream") / "test_string_stream")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_string_stream.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I", _paths.CSRC_DIR, "-I", _paths.ORACLE_DIR, src, os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"),
                    "-lm", "-o", exe], check=True)

    def run(docs, allow=0):
        blob = b"".join(struct.pack("<I", len(d)) + bytes(d) for d in docs)
        p = subprocess.run([exe, str(allow)], input=blob, capture_output=True)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[:2000]
        m = re.match(r"(\d+) documents, (\d+) byte for byte, (\d+) rejected, (\d+) with unlisted quotes, (\d+) declined", p.stdout.decode())
        assert m, p.stdout
        total, exact, rejected, unlisted, declined = (int(x) for x in m.groups())
        # declined: a \\u pattern the reference rejects somewhere in a document whose listed strings are all valid -- a lone surrogate under
        # allow_replacement, or an escape outside every string; the kernels send such a document down the per-string road.  Counted with
        # the rejected ones: what matters is that none of them is written by the stream.
        run.declined = declined
        return total, exact, rejected + declined, unlisted
    return run


def test_fixtures_and_corpora(model):
    docs = []
    for name in ("twitter.json", "citm_catalog.json", "amazon_cellphones.ndjson"):
        path = os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples", name)
        if os.path.exists(path):
            docs.append(open(path, "rb").read())
    docs += [corpus.twitter_like(2 << 20, 5)[0].tobytes(), corpus.large_random(1 << 20, 3)[0].tobytes(), corpus.escape_heavy(1 << 20)[0].tobytes(),
             b'["","","a",""]', b"[1,2,{}]", b'""', b'"a"', b""]
    total, exact, rejected, unlisted = model(docs)
    assert (exact, rejected, unlisted) == (total, 0, 0)


def test_escapes_at_every_offset(model):
    """every kind of escape at every position relative to the 64-byte blocks (the \\u bookkeeping looks back 10 bytes)"""
    escapes = [b'\\n', b'\\u0041', b'\\u00e9', b'\\u20ac', b'\\ud83d\\ude00', b'\\"', b'\\\\', b'\\/', b'\\t\\r\\b\\f', b'\\ud83d\\ude00\\ud83d\\ude00', b'\\u0041\\u0042\\u0043',
               b'\\\\u0041', b'\\\\\\u0041', b'\\\\\\\\u0041\\ud83d\\ude00']
    docs = []
    for pre in range(0, 140):
        for esc in escapes:
            docs.append(b'["' + b'a' * pre + esc + b'tail","' + esc + b'"]')
    total, exact, rejected, unlisted = model(docs)
    assert (exact, rejected, unlisted) == (total, 0, 0)
    bad = []
    for pre in range(0, 140, 3):
        for esc in (b'\\q', b'\\ud800', b'\\udc00x', b'\\u12G4', b'\\ud83d\\u0041', b'\\ud800\\ud800\\udc00', b'\\udc00\\udc00'):
            bad.append(b'["ok","' + b'a' * pre + esc + b'tail",1]')
    total, exact, rejected, unlisted = model(bad)
    assert rejected == total
    # with replacement characters the surrogate cases are valid strings: the stream declines them (the per-string road substitutes)
    total, exact, rejected, unlisted = model(bad, allow=1)
    assert exact + rejected == total and model.declined > total // 2


def test_hand_written_bodies(model):
    for allow in (0, 1):
        docs = [b'["' + b + b',0]' for b in STRING_BODIES]
        total, exact, rejected, unlisted = model(docs, allow)
        assert total - 10 < exact + rejected + unlisted <= total and exact > 10 and rejected > 10


def test_random_documents(model):
    rng = np.random.default_rng(99)
    docs = [jsongen.random_document(rng) for _ in range(4000)]
    total, exact, rejected, unlisted = model(docs)
    assert (exact, rejected, unlisted) == (total, 0, 0)
    docs = [jsongen.mutate(rng, jsongen.random_document(rng, max_depth=4)) for _ in range(20000)]
    total, exact, rejected, unlisted = model(docs)
    assert total - 1500 < exact + rejected + unlisted <= total and rejected > 100 and unlisted > 100 and exact > 5000  # (documents stage 1 rejects as UTF-8 are skipped)


def test_random_escape_soup(model):
    """strings drawn from an alphabet of backslashes, 'u', hex digits, surrogate halves and quotes; and from one of valid escapes only"""
    rng = np.random.default_rng(4242)
    hostile = [b'a', b'Z', b' ', b'\\', b'\\\\', b'\\n', b'\\ud83d\\ude00', b'\\u0041', b'\\u20ac', b'u', b'u', b'd', b'D', b'8', b'c', b'0', b'f', b'F', b'9', b'n', b'\\"', b'/', b'x',
               "\u00e9".encode(), "\u65e5".encode(), b'\\ud83d', b'\\ude00', b'\\u00']
    valid = [b'a', b'Z', b' ', b'\\\\', b'\\n', b'\\t', b'\\/', b'\\ud83d\\ude00', b'\\u0041', b'\\u20ac', b'\\u00e9', b'\\uFFFF', b'u', b'd', b'8', b'0', b'f', b'9', b'n', b'\\"', b'/', b'x',
             "\u00e9".encode(), "\u65e5".encode()]
    for allow in (0, 1):
        for alphabet, all_valid in ((hostile, False), (valid, True)):
            docs = []
            for _ in range(5000):
                strings = []
                for _ in range(int(rng.integers(1, 5))):
                    k = int(rng.integers(0, 60))
                    strings.append(b'"' + b''.join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), k)) + b' "')
                docs.append(b'[' + b','.join(strings) + b']')
            total, exact, rejected, unlisted = model(docs, allow)
            if all_valid:
                assert (exact, rejected, unlisted) == (total, 0, 0)
            else:
                assert total - 1500 < exact + rejected + unlisted <= total and exact > 100 and rejected > 1000
