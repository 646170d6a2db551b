"""CPU tier: the data-parallel tape construction (simdjson_amd/csrc/sjgpu_tape.hip) executed step by step on the host with the same
per-token functions (tests/host/test_tape_model.cpp), against the oracle's serial walk: valid documents word for word, broken ones by
error code -- the reference stops at the FIRST token it does not expect, and the model must find that token without walking."""
import os
import struct
import subprocess

import numpy as np
import pytest

import jsongen
from simdjson_amd import _paths


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tape") / "test_tape_model")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_tape_model.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I", _paths.CSRC_DIR, "-I", _paths.ORACLE_DIR, src, os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"),
                    os.path.join(_paths.ORACLE_DIR, "sj_oracle_stage2.c"), "-lm", "-o", exe], check=True)

    def run(docs, max_depth=1024):
        blob = b"".join(struct.pack("<I", len(d)) + d for d in docs)
        p = subprocess.run([exe, str(max_depth)], input=blob, capture_output=True)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[:2000]
        return p.stdout.decode()
    return run


def test_fixtures(model):
    docs = []
    for name in ("twitter.json", "citm_catalog.json", "amazon_cellphones.ndjson"):
        path = os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples", name)
        if os.path.exists(path):
            docs.append(open(path, "rb").read())
    assert len(docs) >= 2
    out = model(docs)
    assert "2 valid" in out or "3 valid" in out, out  # the NDJSON file is many documents: a TAPE_ERROR as one


def test_random_valid_documents(model):
    rng = np.random.default_rng(2026)
    out = model([jsongen.random_document(rng) for _ in range(4000)])
    assert "4000 documents, 4000 valid" in out, out


def test_broken_documents(model):
    rng = np.random.default_rng(77)
    docs = [jsongen.mutate(rng, jsongen.random_document(rng, max_depth=4)) for _ in range(20000)]
    out = model(docs)
    for code in (3, 5, 6, 7, 8, 9, 10):
        assert f"code {code}:" in out, out


def test_numbers_and_hand_written_cases(model):
    docs = []
    for t in jsongen.number_corner_cases():
        t = t.encode()
        docs += [b"[" + t + b"]", b'{"k":' + t + b" }", t, b"[1," + t + b",2]"]
    docs += [b"[,]", b"[ ,1]", b'{"a":,}', b"[1,,2]", b'{"a":1,,}', b'{"a":1 "b":2}', b'{"a" "b"}', b'{"a"}', b"[1 2]", b"[}", b"{]", b"[1}", b'{"a":1]', b"]", b"}", b"[]]", b"{}}",
             b"[[]", b"[[1]", b'{"a":{}', b"[", b"{", b'"a" "b"', b"1 2", b"[] []", b"nul", b"tru", b"fals", b"truex", b"[truex]", b"[nullx]", b"[falsey]", b"!", b"[!]", b'{"a":!}',
             b"[1]x", b"x", b"[x]", b'{x:1}', b'{"a":1,x:2}', b"[:]", b"[1:2]", b'{"a"::1}', b'{:1}', b'{,}', b'[,', b"\x01", b"[\x01]", b'["\\q"]', b'{"\\q":1}', b'{"a":"\\q"}',
             b'["a","\\ud800"]', b"[-]", b"[1e]", b"[0123]", b"[1.]", b'[1,"a",true,null,false,{},[],{"b":[]}]', b'{"a":[],"b":{},"c":[[],[[]],{}]}']
    model(docs)


def test_depth_limits(model):
    for max_depth in (1, 2, 3, 4, 16, 1024):
        docs = []
        for depth in (1, 2, 3, 4, 5, 15, 16, 17, 1023, 1024, 1025):
            for inner in (b"", b"1", b"{}", b'{"a":[]}'):
                docs.append(b"[" * depth + inner + b"]" * depth)
                docs.append(b'{"a":' * depth + (inner or b"0") + b"}" * depth)
        docs += [b"[]", b"[[]]", b"[[[]]]", b"[[[[1]]]]", b'{"a":{"b":{"c":{}}}}', b"1", b"[1,[2,[3,[4]]]]"]
        model(docs, max_depth)


def test_wide_containers_saturate_their_count(model):
    """more than 0xFFFFFF elements: the count field of the opening word saturates (tape_builder.h:402-404)"""
    n = 0xFFFFFF + 5
    model([b"[" + b"1," * (n - 1) + b"1]", b"[" + b"1," * 0xFFFFFE + b"1]"])


def test_the_rule_from_tables_is_the_rule(tmp_path):
    """k_tape_rules applies the walk's rule from three small tables (byte properties, state behind a byte, what a state accepts) instead of
    the spelled-out rule this file's model runs: both forms on every combination that can make a difference (2 x 10^9 of them)"""
    exe = str(tmp_path / "test_tape_rules")
    src = os.path.join(_paths.REPO_ROOT, "tests", "host", "test_tape_rules.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", _paths.CSRC_DIR, src, "-o", exe], check=True)
    p = subprocess.run([exe], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[:2000]
    assert b"combinations agree" in p.stdout
