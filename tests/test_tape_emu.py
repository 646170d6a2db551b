"""CPU tier: the gfx950 KERNEL SOURCES of stage 2 -- sjgpu_tape.hip, sjgpu_string_stream.hip, sjgpu_strings.hip and the scans of
sjgpu_finish.hip, compiled as C++ against tests/host/emu (a workgroup = an OS thread, a lane = a fiber) -- run the three launches of
sjgpu_stage2_device on whole documents and are compared with the oracle's serial walk (tests/host/test_tape_emu.cpp): error code
always, tape and string buffer byte for byte when the document is valid.  tests/test_tape_model.py checks the construction; this
checks the kernels as written.  The GPU tier is left with what hipcc and the hardware make of the same source."""
import os
import struct
import subprocess

import numpy as np
import pytest

import jsongen
from simdjson_amd import _paths

CSRC = os.path.join(_paths.PKG_DIR, "csrc")
EMU = os.path.join(_paths.REPO_ROOT, "tests", "host", "emu")
KERNEL_TUS = ("sjgpu_tape", "sjgpu_strings", "sjgpu_string_stream", "sjgpu_finish")


def build(out, defines=()):
    inc = ["-I", EMU, "-I", _paths.INCLUDE_DIR, "-I", CSRC, "-I", _paths.ORACLE_DIR]
    jobs = []
    for name in KERNEL_TUS:
        jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O1", "-Wno-attributes", "-Wno-unknown-pragmas", *defines, "-x", "c++", *inc, "-c",
                                      os.path.join(CSRC, name + ".hip"), "-o", str(out / (name + ".o"))]))
    jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O2", *inc, "-c", os.path.join(EMU, "sj_emu.cpp"), "-o", str(out / "sj_emu.o")]))
    jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O2", "-Wno-attributes", *inc, "-c",
                                  os.path.join(_paths.REPO_ROOT, "tests", "host", "test_tape_emu.cpp"), "-o", str(out / "driver.o")]))
    for name in ("sj_oracle", "sj_oracle_stage2"):
        jobs.append(subprocess.Popen(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-c", os.path.join(_paths.ORACLE_DIR, name + ".c"),
                                      "-o", str(out / (name + ".o"))]))
    assert all(j.wait() == 0 for j in jobs)
    exe = str(out / "test_tape_emu")
    objs = [str(out / (f + ".o")) for f in (*KERNEL_TUS, "sj_emu", "driver", "sj_oracle", "sj_oracle_stage2")]
    subprocess.run(["g++", *objs, "-lpthread", "-lm", "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = build(tmp_path_factory.mktemp("tape_emu"))

    def run(docs, max_depth=1024, force_walk=0, expect_ok=True):
        blob = b"".join(struct.pack("<I", len(d)) + d for d in docs)
        p = subprocess.run([exe, str(max_depth), str(force_walk)], input=blob, capture_output=True, timeout=1500)
        if expect_ok:
            assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
            assert " 0 mismatches" in p.stdout.decode()
        return p.stdout.decode()
    return run


def _big(rng, lo, hi, sep=b",\n"):
    return b"[" + sep.join(jsongen.random_document(rng) for _ in range(int(rng.integers(lo, hi)))) + b"]"


def test_fixtures(emu):
    docs = [open(os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples", name), "rb").read() for name in ("twitter.json", "citm_catalog.json")]
    assert "2 documents, 2 valid" in emu(docs)
    assert "2 documents, 2 valid" in emu(docs, force_walk=1)  # the per-string kernels write the same buffer


def test_small_documents_valid_and_broken(emu):
    rng = np.random.default_rng(11)
    assert "400 documents, 400 valid" in emu([jsongen.random_document(rng) for _ in range(400)])
    out = emu([jsongen.mutate(rng, jsongen.random_document(rng, max_depth=4)) for _ in range(2500)])
    for code in (3, 5, 6, 7, 8, 9):
        assert f"code {code}:" in out, out


def test_documents_of_many_blocks(emu):
    """token blocks of 4 096, sort tiles of 2 048 elements, string segments of 16 KiB: documents that span dozens of each, valid and broken"""
    rng = np.random.default_rng(12)
    docs = [_big(rng, 50, 600) for _ in range(30)] + [_big(rng, 3000, 6000) for _ in range(2)]
    assert f"{len(docs)} documents, {len(docs)} valid" in emu(docs)
    out = emu([jsongen.mutate(rng, _big(rng, 50, 400, b",")) for _ in range(80)])
    assert "code 3:" in out and "code 0:" in out, out


def test_hand_written_cases_numbers_and_depth_limits(emu):
    docs = []
    for t in jsongen.number_corner_cases()[::7]:
        t = t.encode()
        docs += [b"[" + t + b"]", t, b'{"k":' + t + b" }"]
    docs += [b"[,]", b"[ ,1]", b'{"a":,}', b"[1,,2]", b'{"a":1,,}', b'{"a":1 "b":2}', b'{"a" "b"}', b'{"a"}', b"[1 2]", b"[}", b"{]", b"[1}", b'{"a":1]', b"]", b"}", b"[]]", b"{}}",
             b"[[]", b"[[1]", b'{"a":{}', b"[", b"{", b'"a" "b"', b"1 2", b"[] []", b"nul", b"truex", b"[truex]", b"!", b"[!]", b'{"a":!}', b"[1]x", b'{x:1}', b"[:]", b'{"a"::1}',
             b'["\\q"]', b'{"\\q":1}', b'["a","\\ud800"]', b"[-]", b"[0123]", b'[1,"a",true,null,false,{},[],{"b":[]}]', b'{"a":[],"b":{},"c":[[],[[]],{}]}']
    emu(docs)
    for max_depth in (1, 2, 3, 16):
        docs = []
        for depth in (1, 2, 3, 4, 15, 16, 17, 70, 130):
            for inner in (b"", b"1", b"{}", b'{"a":[]}'):
                docs.append(b"[" * depth + inner + b"]" * depth)
                docs.append(b'{"a":' * depth + (inner or b"0") + b"}" * depth)
        emu(docs, max_depth)
    deep = [b"[" * d + b"1" + b"]" * d for d in (63, 64, 65, 200, 1023)] + [b'{"a":[' * 100 + b"{}" + b"]}" * 100]  # both sort paths
    emu(deep)


ESCAPES = [b'\\n', b'\\u0041', b'\\u00e9', b'\\u20ac', b'\\ud83d\\ude00', b'\\"', b'\\\\', b'\\t\\r\\b\\f', b'\\ud83d\\ude00\\ud83d\\ude00', b'\\u0041\\u0042\\u0043', b'\\\\u0041',
           b'\\\\\\u0041', b'\\uD7FF\\uE000', b'\\u07ff\\u0800\\u007f\\u0080', b'\\uDBFF\\uDFFF\\uD800\\uDC00']


def test_escapes_across_chunks_and_segments(emu):
    """The string stream decides which bytes of a \\\\u escape stay by mask algebra with a hand-over from block to block, chunk to chunk (the
    wave's carry) and segment to segment (from the look-back bytes): every escape at every offset around 4 KiB, 16 KiB and 32 KiB, dense runs
    of escapes over several segments; rejected escapes at the same places must send the document down the per-string road."""
    rng = np.random.default_rng(13)
    docs = []
    for boundary in (4096, 16384, 32768):
        for delta in range(-14, 4):
            for esc in ESCAPES:
                docs.append(b'["' + b"a" * (boundary + delta - 2) + esc + b'tail","' + esc * 3 + b'"]')
    for _ in range(6):
        parts, size = [], 0
        while size < 70000:
            parts.append(ESCAPES[int(rng.integers(0, len(ESCAPES)))] if rng.integers(0, 12) else b"x" * int(rng.integers(1, 70)))
            size += len(parts[-1])
        docs.append(b'{"k":"' + b"".join(parts) + b'","n":[1,2,"' + b"".join(parts[:50]) + b'"]}')
    # chunks whose output does not fit k_strs_write's window (empty strings: five bytes out of three) go through it in two passes of 32 lanes: at every
    # phase against the lanes, beside chunks that take one pass, with escapes to patch in both halves
    for phase in range(0, 64, 9):
        docs.append(b"[" + b" " * phase + b'"",' * 900 + b'"ab\\u00e9\\n\\ud83d\\ude00c",' * 40 + b'"",' * 2100 + b'"x\\ty","z"]')
    docs.append(b'{' + b'"":"",' * 6000 + b'"k":""}')
    out = emu(docs)
    assert f"{len(docs)} documents, {len(docs)} valid" in out and f"stream {len(docs)}," in out, out
    bad = []
    for boundary in (4096, 16384):
        for delta in range(-14, 4):
            for esc in (b'\\q', b'\\ud800', b'\\udc00x', b'\\u12G4', b'\\ud83d\\u0041', b'\\ud800\\ud800\\udc00', b'\\udc00\\udc00', b'\\ud83dxude00', b'\\ud83d\\nde00'):
                bad.append(b'["ok","' + b"a" * (boundary + delta - 7) + esc + b'tail",1]')
    out = emu(bad)
    assert f"code 5: {len(bad)}" in out and f"per-string {len(bad)};" in out and f"second rounds: {len(bad)})" in out, out  # declined by the stream, run again


def _staged_front_documents(rng):
    """What k_tok_stage's grouping has to get right: tokens packed densely and far apart in one list (staged groups and rows read in place, side by
    side), numbers and atoms at the last bytes of a window and of the document, numbers longer than the bytes staged behind a group's last token."""
    docs = []
    longs = ["1" * 70 + ".0", "0." + "7" * 90, "-" + "9" * 18, "1" + "0" * 200 + "e-190", "3." + "1" * 400 + "e5", "1" * 25 + "e0", "12345678901234567890", "0.000000000000000000000000000001e31"]
    for gap in (0, 1, 7, 40, 61, 64, 70, 200, 1000, 5000):
        pad = b" " * gap
        for sep_in in (b"", pad):
            items = []
            for k in range(300):
                kind = int(rng.integers(0, 7))
                if kind == 0: items.append(str(int(rng.integers(-10**9, 10**9))).encode())
                elif kind == 1: items.append(repr(float(rng.random() * 10.0 ** int(rng.integers(-20, 20)))).encode())
                elif kind == 2: items.append([b"true", b"false", b"null"][int(rng.integers(0, 3))])
                elif kind == 3: items.append(b'"s%d"' % k)
                elif kind == 4: items.append(b'{"a":' + sep_in + b'1.5e3,"b":[true,null]}')
                elif kind == 5: items.append(longs[int(rng.integers(0, len(longs)))].encode())
                else: items.append(b"[]")
            docs.append(b"[" + (b"," + pad).join(items) + pad + b"]")
    # the end of the document: no byte behind the token
    docs += [b"123", b"-0", b"1.5", b"true", b"false", b"null", b"[1,2,3", b"[true", b'{"a":null', b"1" * 70, b"[" + b"1" * 70 + b"]", b"[1.0" + b"0" * 100 + b"]", b"[1," + b" " * 4000 + b"-" + b"9" * 19 + b"]",
             b"1." + b"5" * 70, b"[0" + b" " * 61 + b",1" + b"2" * 62 + b".5," + b"3" * 17 + b"]"]
    # a long run of dense tokens, a gap, dense again: groups of every size
    for gap in (3000, 4090, 4100, 9000, 70000):
        docs.append(b"[" + b",".join(b"%d.5" % k for k in range(3000)) + b"," + b" " * gap + b",".join(b"true" for _ in range(2000)) + b"," + b" " * gap + b'"x",1e5]')
    return docs


def test_staged_token_front(emu):
    rng = np.random.default_rng(14)
    docs = _staged_front_documents(rng)
    out = emu(docs)
    assert "code 0:" in out, out
    # broken spellings and numbers at the same places: the same error codes as the oracle's walk (compared inside the program)
    bad = []
    for d in docs[:40]:
        for _ in range(3):
            bad.append(jsongen.mutate(rng, d))
    emu(bad)


def test_staged_token_front_with_a_small_window(tmp_path_factory):
    """the same sources with a 1 KiB window: every wave of a dense document works through many groups, most rows of ordinary text are read in place"""
    exe = build(tmp_path_factory.mktemp("tape_emu_w1k"), defines=("-DSJGPU_STAGE_WINDOW=1024",))
    rng = np.random.default_rng(15)
    docs = _staged_front_documents(rng) + [_big(rng, 50, 600) for _ in range(6)]
    docs += [open(os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples", "twitter.json"), "rb").read()]
    blob = b"".join(struct.pack("<I", len(d)) + d for d in docs)
    p = subprocess.run([exe, "1024", "0"], input=blob, capture_output=True, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    assert " 0 mismatches" in p.stdout.decode()
