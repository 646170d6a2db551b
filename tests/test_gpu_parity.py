"""GPU tier (-m gpu): the HIP path, called through the C-ABI (simdjson_amd.capi -> libsjgpu.so), against
  * the committed golden fixtures produced by the real reference (tests/golden/*.json),
  * the C oracle run live on the same seeded inputs,
  * size-independent properties at BASELINE.json's full size (1 GiB).
Bit-exact everywhere: indices, sentinels, n, error codes, minified bytes, UTF-8 verdicts."""
import os

import numpy as np
import pytest

import checkers
from simdjson_amd import _paths, build, capi, corpus
from test_oracle_golden import digest_results, load, make_corpus

pytestmark = pytest.mark.gpu

CAP = 128 << 20


def _parser(kind, capacity=CAP):
    """kind: "fused" / "split" = that tile pipeline for EVERY document (the one-workgroup kernel for small documents
    switched off, streaming modes finished on the host); "docs" = the defaults (documents up to 64 KiB take the
    one-workgroup kernel, sjgpu_small.hip); "device_finish" = tile pipelines + every streaming-mode call finished on
    the device (sjgpu_finish.hip).  The switches are read when the context is made."""
    env = {"fused": {"SJGPU_SMALL_DOCS": "0", "SJGPU_FINISH": "host"}, "split": {"SJGPU_SMALL_DOCS": "0", "SJGPU_FINISH": "host"},
           "docs": {}, "device_finish": {"SJGPU_SMALL_DOCS": "0", "SJGPU_FINISH": "device"}}[kind]
    old = {k: os.environ.get(k) for k in ("SJGPU_SMALL_DOCS", "SJGPU_FINISH")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        p = capi.DomParserImplementation(capacity)  # raises loudly if the HIP library or the GPU is missing
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    p.set_pipeline({"fused": "fused", "split": "split"}.get(kind, "auto"))
    return p


@pytest.fixture(scope="module", params=["fused", "split", "docs", "device_finish"])
def gpu(request):
    """Every route a document can take must be bit-exact: the single-pass kernels, the 3-kernel split pipeline, the
    one-workgroup kernel for small documents, and the device-side finish of the streaming modes."""
    build.build_sjgpu()
    p = _parser(request.param)
    yield p
    p.close()


@pytest.fixture(scope="module")
def orc():
    return checkers.Oracle()


@pytest.fixture(scope="module")
def ref():
    """the real reference where its library travelled along (and the host CPU can run an x86 kernel), else None"""
    if not checkers.have_reference_lib():
        return None
    r = checkers.Reference()
    return r if r.best_impl() else None


def g_stage1(p, data, mode=0):
    err = p.stage1(data, mode)
    n = p.n_structural_indexes
    return err, n, p.structural_indexes[: n + 3].copy()


def first_diff(a, b):
    m = min(len(a), len(b))
    d = np.nonzero(a[:m] != b[:m])[0]
    return (int(d[0]), int(a[d[0]]), int(b[d[0]])) if len(d) else ("len", len(a), len(b))


def assert_same_stage1(p, orc, data, mode=0, tag=""):
    got = checkers.observable(data, mode, *g_stage1(p, data, mode))
    want = checkers.observable(data, mode, *orc.stage1(data, mode))
    if got != want:
        detail = (got[:2], want[:2])
        if len(got) > 2 and len(want) > 2:
            detail += (first_diff(np.array(got[2]), np.array(want[2])),)
        raise AssertionError(f"stage1 mismatch {tag} mode={mode} len={len(data)}: {detail}")


def assert_same_all(p, orc, data, tag="", modes=(0,)):
    for mode in modes:
        assert_same_stage1(p, orc, data, mode, tag)
    gerr, gout = p.minify(data)
    oerr, oout = orc.minify(data)
    assert gerr == oerr and np.array_equal(gout, oout), f"minify mismatch {tag} len={len(data)}: {gerr} vs {oerr}, " \
        f"{first_diff(np.asarray(gout), np.asarray(oout)) if gerr == oerr else ''}"
    assert p.validate_utf8(data) == orc.validate_utf8(data), f"utf8 mismatch {tag}"


# ---- golden fixtures (reference outputs) ------------------------------------------------------------------
def test_the_library_on_this_box_was_built_from_these_sources():
    """The driver's box runs a PREBUILT libsjgpu.so (it travels with the snapshot): its stamp -- the sha256 of the sources it was compiled from,
    written by simdjson_amd/build.py -- must be the hash of the sources lying next to it, or every result below is about some other tree."""
    from simdjson_amd import build
    assert build.sjgpu_is_current(), "simdjson_amd/lib/libsjgpu.so was not built from the sources of this tree (build/tests/STAMP.json)"


def test_small_cases_all_modes_golden(gpu):
    for case in load("small_cases.json")["cases"]:
        data = bytes.fromhex(case["hex"])
        for mname, mode in checkers.MODES.items():
            obs = checkers.observable(data, mode, *g_stage1(gpu, data, mode))
            got = {"err": obs[0]} if len(obs) == 1 else {"err": obs[0], "n": obs[1], "idx": list(obs[2])}
            assert got == case["stage1"][mname], (data, mname, got)
        merr, mout = gpu.minify(data)
        assert merr == case["minify"]["err"] and bytes(mout).hex() == case["minify"]["hex"], data
        assert gpu.validate_utf8(data) == case["utf8"], data


def test_utf8_known_answer_vectors_golden(gpu):
    v = load("utf8_vectors.json")
    for g in v["good"]:
        assert gpu.validate_utf8(bytes.fromhex(g)), g
    for b in v["bad"]:
        assert not gpu.validate_utf8(bytes.fromhex(b)), b
    # tests/dom/basictests.cpp:1814-1909 shapes
    for n in range(0, 129, 7):
        assert gpu.validate_utf8(b" " * n)
        for off in range(0, n, 5):
            a = bytearray(b" " * n)
            a[off] = 0xFF
            assert not gpu.validate_utf8(bytes(a))
    assert not gpu.validate_utf8(b"\xf0\x8f\xbf\xbf")


def test_random_adversarial_digests_golden(gpu, orc):
    g = load("random_digest.json")
    for d in g["digests"][:6]:
        got = digest_results(orc, lambda a, m: g_stage1(gpu, a, m), gpu.minify, gpu.validate_utf8, d["seed"])
        assert got == d["fnv"], d["seed"]


def test_corpora_digests_golden(gpu, orc):
    for d in load("corpora.json")["corpora"]:
        a = make_corpus(d["name"])  # incl. jsonexamples/* (BASELINE configs[0]): committed fixtures, they travel
        assert len(a) == d["len"] and orc.fnv(a) == d["buf_fnv"], d["name"]
        err, n, idx = g_stage1(gpu, a, 0)
        assert (err, n) == (d["stage1_err"], d["n"]), (d["name"], err, n)
        assert orc.fnv(idx) == d["idx_fnv"], d["name"]
        merr, mout = gpu.minify(a)
        assert (merr, len(mout), orc.fnv(mout)) == (d["minify_err"], d["minify_len"], d["minify_fnv"]), d["name"]
        assert gpu.validate_utf8(a) == d["utf8"], d["name"]


# ---- live oracle: boundaries, carries, adversarial --------------------------------------------------------------
BOUNDARIES = (64, 4096, 16384, 32768)


def test_payloads_straddling_every_boundary(gpu, orc):
    v = load("utf8_vectors.json")
    payloads = [bytes.fromhex(x) for x in v["good"] + v["bad"] if len(x) <= 20]
    payloads += [b'"', b'\\"', b'\\\\"', b'\\\\\\"', b'"a\\"b"', b'"ab"cd', b"true", b"[1,2]", b'{"a":"\\\\"}', b"\x01", b'"\x01"',
                 b"\x0c\x1a", b'"\\', b'\\', b"\xe2\x82\xac", b"\xf0\x9f\x98\x80", b'"\xf0\x9f\x98\x80"', b'a"b"c', b'" "', b"\x1e1"]
    for B in BOUNDARIES:
        for pay in payloads:
            for before in range(0, len(pay) + 2):
                for filler, pre in ((b" ", b""), (b"x", b'["'), (b" ", b'"')):
                    start = B - before
                    body = pre + filler * (start - len(pre)) + pay + filler * 70
                    assert_same_all(gpu, orc, body, tag=f"B={B} before={before} pay={pay!r} filler={filler!r}")


def test_quote_parity_and_escape_carry_across_segments(gpu, orc):
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = int(rng.integers(100_000, 400_000))
        a = corpus.random_adversarial(n, 100 + trial, ascii_only=bool(trial % 2), p_backslash=(0.0, 0.05, 0.3)[trial % 3])
        assert_same_all(gpu, orc, a, tag=f"random trial {trial}", modes=(0, 1, 2))
    # backslash runs of critical lengths ending exactly at block / chunk / segment boundaries
    for B in BOUNDARIES:
        for run in (1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 16383, 16384, 16385, 40000):
            for shift in (0, 1):
                start = max(1, B - run + shift)
                body = b'"' + b"x" * (start - 1) + b"\\" * run + b'" ,"y"' + b" " * 100
                assert_same_all(gpu, orc, body, tag=f"run={run} B={B} shift={shift}")
    # strings that stay open across many segments; parity flips exactly at boundaries
    for B in BOUNDARIES:
        for off in (-1, 0, 1):
            body = b"[" + b" " * (B + off - 2) + b'"' + b"z{[,:]}" * 9000 + b'"' + b",1]" + b" " * 50
            assert_same_all(gpu, orc, body, tag=f"long string B={B} off={off}")


def test_adversarial_shapes(gpu, orc):
    for k in (1, 31, 32, 33, 2047, 2048, 2049, 8191, 8192, 8193, 100_000, 1 << 20):
        assert_same_all(gpu, orc, corpus.deep_nesting(k), tag=f"deep_nesting {k}")
    # every emission regime of a 4 KiB chunk: one window, two windows (per-lane extraction), dense (block expansion);
    # densities from 0.05 to 1.0 in one document so that neighbouring chunks take different paths
    rng = np.random.default_rng(17)
    units = [b"1,", b"[],", b'{"a":1},', b'"k":12,', b"1234567,", b'"some text here",', b"              1,", b"[[[[]]]],"]
    for weights in ([1] * len(units), [8, 4, 1, 0, 0, 0, 0, 4], [0, 0, 1, 2, 4, 4, 2, 0], [1, 0, 0, 0, 0, 0, 0, 0]):
        pick = rng.choice(len(units), size=200000, p=np.array(weights) / sum(weights))
        assert_same_all(gpu, orc, b"[" + b"".join(units[i] for i in pick) + b"0]", tag=f"mixed density {weights}")
    runs = [1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 16383, 16384, 16385, (1 << 20) - 1, 1 << 20, (1 << 20) + 1]
    for pad in (0, 1, 37, 63):
        assert_same_all(gpu, orc, corpus.backslash_runs(runs, pad), tag=f"backslash_runs pad {pad}")
    # control characters 0x00-0x1F inside and outside strings, incl. the 0x0C / 0x1A operator quirk
    for c in range(0x20):
        assert_same_all(gpu, orc, b"[1," + bytes([c]) + b"2]", tag=f"ctrl outside {c}")
        assert_same_all(gpu, orc, b'["a' + bytes([c]) + b'b"]', tag=f"ctrl inside {c}")
    # every length 0..200 of a few shapes (tail handling)
    base = b'{"k":"v\\"w","n":[1,2,3],"u":"\xe2\x82\xac\xf0\x9f\x98\x80"} ' * 5
    for n in range(0, 201):
        assert_same_all(gpu, orc, base[:n], tag=f"prefix {n}", modes=(0, 1, 2))


def test_control_character_resolution(gpu, orc):
    """Segments whose first chunk holds a control character fix their own in-string carry-in (one mask
    plane); those without one keep both hypotheses.  Both kinds, valid and invalid, mixed in one buffer."""
    seg = 16384
    # minified (no control characters at all): every segment unresolved
    a, _ = corpus.twitter_like(300_000, 3)
    _, mini = orc.minify(a)
    assert_same_all(gpu, orc, mini, tag="minified: unresolved segments", modes=(0, 2))
    # a string that spans several segments with a raw newline at chosen places: UNESCAPED_CHARS, and the
    # resolving character is the offender (derived carry-in != true carry-in)
    for where in (seg + 5, seg + 4095, seg + 4096, 2 * seg + 100, 3 * seg - 1, 3 * seg):
        body = bytearray(b'["' + b"k" * (4 * seg) + b'",1,2,3]' + b" \n" * 10)
        body[where] = 0x0A
        assert_same_all(gpu, orc, bytes(body), tag=f"newline inside long string at {where}", modes=(0, 1, 2))
    # valid: long string without control characters (unresolved segments) followed by pretty-printed
    # content (resolved segments); quote parity must carry across the boundary between the two kinds
    for pad in (0, 1, 4095, 4096, 4097):
        body = b'{"s":"' + b"z[]{}:," * 7000 + b"q" * pad + b'",\n  "t": [1, 2, 3],\n' + b'  "u": "a\\"b",\n' * 3000 + b'"v":0}'
        assert_same_all(gpu, orc, body, tag=f"mixed resolved/unresolved pad {pad}", modes=(0,))
    # control character in the first chunk but INSIDE a string that started in an earlier segment
    body = b'["' + b"x" * (seg - 10) + b"\t" + b"y" * 100 + b'"]'
    assert_same_all(gpu, orc, body, tag="tab inside string, first chunk of segment 1")
    # 0x1E / 0x0C / 0x1A count as control characters too (they are also scalars / operators outside strings)
    for c in (0x1E, 0x0C, 0x1A, 0x00):
        body = b" " * (seg - 3) + bytes([c]) + b'{"a":"' + bytes([c]) + b'"}' + b" " * 50
        assert_same_all(gpu, orc, body, tag=f"ctrl {c:#x} outside then inside")
        body = b" " * (seg + 7) + bytes([c]) + b' {"a":"b"} ' * 2000
        assert_same_all(gpu, orc, body, tag=f"ctrl {c:#x} resolves segment 1", modes=(0, 3, 4))


def _random_document(rng, target):
    """Random JSON text (objects / arrays / strings with escapes and multi-byte UTF-8 / numbers / literals),
    sometimes pretty-printed, sometimes minified, roughly `target` bytes."""
    import json
    words = ["a", "key", "\u00e9t\u00e9", "\u65e5\u672c\u8a9e", "\U0001f600", "q\"uote", "back\\slash", "tab\tnl\n", "x" * 70,
             "\\" * 33, "{[,:]}", "\u0001ctrl", "", "\u2028", "~" * 130]

    def value(depth):
        r = rng.random()
        if depth > 6 or r < 0.35:
            k = rng.integers(0, 6)
            return [None, True, False, int(rng.integers(-10**9, 10**9)), float(rng.random() * 1e6),
                    " ".join(words[int(i)] for i in rng.integers(0, len(words), size=int(rng.integers(0, 6))))][int(k)]
        if r < 0.65:
            return [value(depth + 1) for _ in range(int(rng.integers(0, 8)))]
        return {words[int(rng.integers(0, len(words)))] + str(i): value(depth + 1) for i in range(int(rng.integers(0, 8)))}

    parts, size = [], 0
    while size < target:
        style = int(rng.integers(0, 3))
        t = json.dumps(value(0), ensure_ascii=bool(rng.integers(0, 2)), indent=(None, 1, 4)[style],
                       separators=(",", ":") if style == 0 else None)
        parts.append(t)
        size += len(t) + 1
    return ("\n" if rng.integers(0, 2) else " ").join(parts).encode("utf-8")


def test_differential_fuzz_structured_documents(gpu, orc):
    """Cross-implementation differential fuzzing in the spirit of the reference's fuzz/fuzz_implementations.cpp
    and fuzz/fuzz_minifyimpl.cpp: random structured documents, then random byte-level mutations, GPU vs oracle."""
    rng = np.random.default_rng(20260921)
    poison = np.frombuffer(b'"\\\\"\x00\x01\x1f\n\t{}[],: \xff\xc0\xe2\x82\xf0\x9f\x80\xed\xa0', dtype=np.uint8)
    for it in range(60):
        target = int(rng.choice([200, 5_000, 70_000, 300_000]))
        doc = np.frombuffer(_random_document(rng, target), dtype=np.uint8).copy()
        assert_same_all(gpu, orc, doc, tag=f"fuzz {it} pristine", modes=(0, 1, 2))
        for mut in range(3):
            b = doc.copy()
            k = int(rng.integers(1, 6))
            pos = rng.integers(0, len(b), size=k)
            b[pos] = poison[rng.integers(0, len(poison), size=k)]
            if mut == 2:  # truncate as a stream window would
                b = b[: int(rng.integers(1, len(b) + 1))]
            assert_same_all(gpu, orc, b, tag=f"fuzz {it} mutation {mut} at {pos.tolist()}", modes=(0, 1, 2, 5, 6))


def test_two_contexts_on_two_threads(orc):
    """One context per parser object, usable from any thread (document_stream's stage-1 worker pattern)."""
    import threading
    docs = [corpus.twitter_like(400_000, 50 + i)[0] for i in range(2)]
    want = [orc.stage1(d, 0) for d in docs]
    errors = []

    def worker(i):
        try:
            p = capi.DomParserImplementation(1 << 20)
            for _ in range(40):
                err = p.stage1(docs[i], 0)
                n = p.n_structural_indexes
                if (err, n) != want[i][:2] or not np.array_equal(p.structural_indexes[: n + 3], want[i][2]):
                    errors.append((i, err, n))
                    break
            p.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_streaming_modes_on_bulk(gpu, orc):
    a, _ = corpus.amazon_ndjson(3 << 20, 5)
    cut = a[: len(a) - 777]  # ends inside a line
    for mode in (1, 2):
        assert_same_stage1(gpu, orc, cut, mode, "ndjson cut")
    seq = np.frombuffer(b"".join(b"\x1e" + bytes(l) + b"\n" for l in bytes(a[:200_000]).split(b"\n") if l), dtype=np.uint8)
    for mode in (3, 4):
        assert_same_stage1(gpu, orc, seq, mode, "json_sequence")
        assert_same_stage1(gpu, orc, seq[:-300], mode, "json_sequence cut")
    com = np.frombuffer(b",".join(bytes(l) for l in bytes(a[:200_000]).split(b"\n") if l), dtype=np.uint8)
    for mode in (5, 6):
        assert_same_stage1(gpu, orc, com, mode, "comma_delimited")
        assert_same_stage1(gpu, orc, com[:-300], mode, "comma_delimited cut")


def test_three_gib_offsets_do_not_wrap(orc):
    """Near the API's size limit (u32 offsets, len <= 0xFFFFFFFF): positions beyond 2^31 and segment / group
    counts beyond 2^17 must not overflow anywhere.  3 GiB of NDJSON, exact digest against the oracle."""
    import torch
    size = int(os.environ.get("SJGPU_HUGE_SIZE", str(3 << 30)))
    if size == 0:
        pytest.skip("disabled")
    a, _ = corpus.amazon_ndjson(size, 31)
    L = len(a)
    assert L > (1 << 31)
    oerr, on, oidx = orc.stage1(a, 0)
    for pipeline in ("split", "fused"):
        p = capi.DomParserImplementation(L)
        p.set_pipeline(pipeline)
        buf = torch.from_numpy(a).cuda()
        words = on + 1000
        idx = torch.empty(words, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), words, stream) == 0
        n, flags, _ = p.result(stream)
        assert (n, flags) == (on, 0), (pipeline, n, on, flags)
        host = idx[: n + 3].cpu().numpy().view(np.uint32)
        assert orc.fnv(host) == orc.fnv(oidx), (pipeline, first_diff(host, oidx))
        # too small an index array is reported, not overrun
        small = torch.empty(1000, dtype=torch.int32, device="cuda")
        assert p.stage1_device(buf.data_ptr(), L, small.data_ptr(), 1000, stream) == 0
        assert p.result(stream)[1] & capi.F_IDX_OVERFLOW
        del buf, idx, small
        p.close()


def test_dense_non_ascii_text_with_errors(orc):
    """Every block holds multi-byte characters (the UTF-8 list fills up several times per span and is drained 64 blocks at
    a time); one broken byte anywhere must be found, by every route."""
    rng = np.random.default_rng(123)
    unit = ('"' + "\u65e5\u672c\u8a9e\u30c6\u30ad\u30b9\u30c8 \U0001F600 caf\u00e9 " * 40 + '",\n').encode()
    doc = np.frombuffer(b"[" + unit * (24 << 20 // len(unit) // 1) + b"0]", np.uint8) if False else np.frombuffer(b"[" + unit * ((24 << 20) // len(unit)) + b"0]", np.uint8)
    for kind in ("fused", "split", "docs"):
        p = _parser(kind, 32 << 20)
        assert_same_all(p, orc, doc, f"{kind}: dense non-ASCII, valid")
        for trial in range(6):
            bad = doc.copy()
            pos = int(rng.integers(2, len(bad) - 2))
            bad[pos] = [0xFF, 0xC0, 0x80, 0xED, 0xF5, 0xE0][trial]
            assert_same_all(p, orc, bad, f"{kind}: dense non-ASCII, byte {pos} broken")
        small = doc[: 50_000].copy()  # the one-workgroup kernel / 16 KiB tiles
        small[-1] = ord("]")
        small[37_001] = 0xFF
        assert_same_all(p, orc, small, f"{kind}: dense non-ASCII, 50 KB")
        p.close()


def test_dense_non_ascii_text_ending_inside_a_character(orc):
    """Dense non-ASCII chunks are validated in line from the scan's bit planes (utf8_dense_chunk) -- the sequence-open-at-the-end
    rule (utf8_lookup4_algorithm.h:164-171) included: inputs that end exactly on a 4 KiB chunk boundary, and one byte to either
    side of it, inside a 2-, 3- or 4-byte character, inside and outside a string; through the one-workgroup kernel, the
    16 KiB-tile kernel and both large pipelines; minify and validate_utf8 see the same inputs."""
    text = ("\u65e5\u672c\u8a9e \U0001F600\u00e9" * 2000).encode()
    for kind in ("fused", "split", "docs"):
        p = _parser(kind, 32 << 20)
        for size in (4096, 8192, 65536, 1 << 20, 12 << 20):
            big = size >= (1 << 20)  # the large pipelines: fewer combinations, the oracle walks them byte by byte
            for delta in ((0, 1) if big else (-1, 0, 1)):
                L = size + delta
                for tail in ((b"\xe6\x97", b"\xf0\x9f\x98", b"\xe6\x97\xa5") if big else (b"\xe6", b"\xe6\x97", b"\xf0\x9f", b"\xf0\x9f\x98", b"\xc3", b"\xe6\x97\xa5")):
                    body = (text * (L // len(text) + 1))[: L - 2 - len(tail)]
                    while body and (body[-1] & 0xC0) == 0x80:  # cut the filler on a character boundary
                        body = body[:-1]
                    if body and body[-1] >= 0xC0:
                        body = body[:-1]
                    body = body + b"x" * (L - 2 - len(tail) - len(body))
                    doc = np.frombuffer(b'["' + body + tail, np.uint8)  # ends inside the string: UNCLOSED_STRING outranks the UTF-8 verdict in stage 1
                    assert len(doc) == L
                    assert_same_all(p, orc, doc, f"{kind}: {L} bytes ending with {tail!r} inside a string")
                    closed = np.frombuffer(b'["' + body[:-2] + b'"]' + tail, np.uint8)  # the same bytes behind the document: UTF8_ERROR
                    assert len(closed) == L
                    assert_same_all(p, orc, closed, f"{kind}: {L} bytes ending with {tail!r} behind the document")
        p.close()


def test_pipelined_kernels_are_deterministic(orc):
    """Tile order, ticket order and look-back timing vary from run to run; the output must not."""
    import torch
    a = corpus.twitter_like(300 << 20, 17)[0]
    _, mini = orc.minify(a)
    L = len(mini)
    oerr, on, oidx = orc.stage1(mini, 0)
    want_idx = orc.fnv(oidx)
    p = capi.DomParserImplementation(L)
    p.set_pipeline("fused")
    buf = torch.from_numpy(mini.copy()).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    dst = torch.empty(L + 64, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(12):
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
        n, flags, _ = p.result(stream)
        assert (n, flags) == (on, 0), (rep, n, on, flags)
        assert orc.fnv(idx[: n + 3].cpu().numpy().view(np.uint32)) == want_idx, rep
        assert p.minify_device(buf.data_ptr(), L, dst.data_ptr(), stream) == 0
        _, mflags, mlen = p.result(stream)
        assert (mlen, mflags) == (L, 0), (rep, mlen, L, mflags)  # minified text is its own minification
        assert bool((dst[:L] == buf).all()), rep
    p.close()


# ---- many small documents in one launch (sjgpu_small.hip, sjgpu_stage1_many) --------------------------------------------------
def test_many_small_documents_in_one_launch(orc):
    rng = np.random.default_rng(77)
    p = capi.DomParserImplementation(1 << 20)
    tw = corpus.twitter_like(400_000, 9)[0]
    docs = [b"", b"[1]", b'{"a":"b"}', b'["unclosed', b'["ctrl \x01 inside"]', b"\xff", b'["bad utf8 \xff"]', b"   ", bytes(tw[:70_000]), bytes(tw[:16384]),
            bytes(tw[:16385]), bytes(tw[:4096]), bytes(tw[:4095]), b'"' + b"\\" * 5000 + b'"', b'["' + "\u65e5\u672c\u8a9e".encode() * 3000 + b'"]']
    for k in range(200):
        n = int(rng.integers(1, 3000))
        docs.append(bytes(corpus.random_adversarial(n, 9000 + k, ascii_only=bool(k % 2), p_backslash=(0.0, 0.2)[k % 2])))
    for k in range(40):
        docs.append(_random_document(rng, int(rng.choice([100, 3000, 40000]))))
    docs.append(b"x" * (2 << 20))  # beyond the capacity of this context
    for batch in (docs, docs[:5], docs[8:9] * 40):  # the middle one travels zero-copy, the others are staged through HBM
        got = p.stage1_many(batch)
        assert len(got) == len(batch)
        for d, (err, n, idx) in zip(batch, got):
            if len(d) > (1 << 20):
                assert err == capi.CAPACITY
                continue
            oerr, on, oidx = orc.stage1(d, 0)
            assert err == oerr, (d[:40], err, oerr)
            if checkers.observable(d, 0, oerr, on, oidx)[1:]:
                assert n == on and np.array_equal(idx, oidx), (d[:40], n, on)
    p.close()


def test_contexts_are_cheap_to_make(orc):
    """The reference makes a dom_parser_implementation per parser object (its document_stream tests: 106 000 of them):
    contexts come from a pool, and a small document is one launch and one wait."""
    import time
    doc = b'{"k":[1,2,3],"s":"text"}'
    want = orc.stage1(doc, 0)
    p = capi.DomParserImplementation(1000)
    p.stage1(doc, 0)
    p.close()
    t0 = time.perf_counter()
    for _ in range(3000):
        p = capi.DomParserImplementation(1000)
        err = p.stage1(doc, 0)
        assert err == want[0] and p.n_structural_indexes == want[1]
        p.close()
    dt = time.perf_counter() - t0
    assert dt < 3.0, f"3000 x (make a context, scan 24 bytes, drop it) took {dt:.2f} s"


# ---- the list after the scan, on the device (sjgpu_finish.hip) --------------------------------------------------------------------
def _stream_shapes():
    nd = corpus.amazon_ndjson(2 << 20, 5)[0]
    lines = [bytes(l) for l in bytes(nd[:300_000]).split(b"\n") if l]
    tw = corpus.twitter_like(300_000, 3)[0]
    shapes = {
        "ndjson": bytes(nd), "ndjson cut inside a line": bytes(nd[:-777]), "ndjson cut inside a string": bytes(nd[: len(nd) - 60]) + b' "dangling',
        "one big document": bytes(tw), "one big document, truncated": bytes(tw[:-500]), "leading blanks then a truncated document": b"   \n " + bytes(tw[:-500]),
        "rs": b"".join(b"\x1e" + l + b"\n" for l in lines), "rs cut": b"".join(b"\x1e" + l + b"\n" for l in lines)[:-300],
        "rs runs": b"\x1e \x1e\x1e\n".join(lines[:200]) + b"\x1e", "rs glued scalars": b"\x1e1 \x1e true\x1e\x1e\"s\" \x1e[1]\x1e 2.5",
        "rs single": b"\x1e" + lines[0], "no rs at all": b" ".join(lines[:50]),
        "commas": b",".join(lines), "commas cut": b",".join(lines)[:-300], "commas nested only": b"[" + b",".join(lines[:50]) + b"]",
        "commas and blanks": b" , ".join(lines[:300]) + b" ,", "single value": b"12345", "two scalars": b"1 2", "unbalanced": b"{[}] 1",
        "closers first": b"]} {\"a\":1} [", "bad utf8": b'{"a":"\xff"} {"b":1} {"c"',
    }
    return shapes


def test_device_finish_equals_host_finish(orc):
    """All six streaming modes: the device-side filters / boundary search against the oracle (= the reference's finish()),
    through sjgpu_stage1 with SJGPU_FINISH=device and through the device-resident sjgpu_stage1_finish_device."""
    import torch
    p = _parser("device_finish", 8 << 20)
    q = _parser("fused", 8 << 20)  # raw scans for the device-resident API
    stream = torch.cuda.current_stream().cuda_stream
    for name, data in _stream_shapes().items():
        a = checkers.as_u8(data)
        for mode in range(1, 7):
            assert_same_stage1(p, orc, a, mode, f"device finish: {name}")
            # device-resident: scan, then finish where the list lies
            ln = checkers.trim_partial_utf8_len(a)
            if ln == 0:
                continue
            buf = torch.from_numpy(a[:ln].copy()).cuda()
            idx = torch.zeros(ln + 16, dtype=torch.int32, device="cuda")
            assert q.stage1_device(buf.data_ptr(), ln, idx.data_ptr(), ln + 3, stream) == 0
            n_raw, flags, _ = q.result(stream)
            want = checkers.observable(a, mode, *orc.stage1(a, mode))
            if flags & capi.F_UNESCAPED_CTRL:
                continue
            err, n, nxt = q.stage1_finish_device(buf.data_ptr(), ln, mode, idx.data_ptr(), n_raw, flags, stream)
            got = (err,) if err in checkers.EARLY else (err, n, tuple(int(x) & 0xFFFFFFFF for x in idx[: n + 3].cpu().numpy()))
            assert got == want, (name, mode, got[:2], want[:2])
    p.close()
    q.close()


def test_depth_scan(orc):
    """depth[i] = containers open in front of structural i (the `depth` of the reference's stage 2), as a prefix scan on
    the device; checked against a running count over the reference's own structural list."""
    import torch
    p = capi.DomParserImplementation(64 << 20)
    stream = torch.cuda.current_stream().cuda_stream
    ex = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsonexamples")
    docs = {"twitter.json": np.fromfile(os.path.join(ex, "twitter.json"), dtype=np.uint8), "citm": np.fromfile(os.path.join(ex, "citm_catalog.json"), dtype=np.uint8),
            "large_random 20 MiB": corpus.large_random(20 << 20, 4)[0], "deep_nesting": corpus.deep_nesting(1 << 20),
            "unbalanced": np.frombuffer(b"]]}{[[ 1, 2 ]", np.uint8), "scalar": np.frombuffer(b"17", np.uint8)}
    for name, a in docs.items():
        L = len(a)
        buf = torch.from_numpy(a.copy()).cuda()
        idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
        n, flags, _ = p.result(stream)
        oidx, _ = orc.scan(a)
        assert n == len(oidx)
        depth = torch.full((n + 1,), -99, dtype=torch.int32, device="cuda")
        p.depth_scan_device(buf.data_ptr(), idx.data_ptr(), n, depth.data_ptr(), stream)
        got = depth.cpu().numpy()
        c = a[oidx]
        delta = np.isin(c, [ord("{"), ord("[")]).astype(np.int64) - np.isin(c, [ord("}"), ord("]")]).astype(np.int64)
        want = np.concatenate([[0], np.cumsum(delta)])
        assert np.array_equal(got, want), (name, first_diff(got, want))
        if name in ("twitter.json", "citm", "large_random 20 MiB", "deep_nesting"):
            assert want[-1] == 0 and want.min() == 0  # a valid document closes what it opens
    p.close()


def _minified(a):
    """the same document without whitespace outside strings (no control character: every 16 KiB segment must carry both in-string hypotheses)"""
    import json
    return np.frombuffer(json.dumps(json.loads(bytes(a)), separators=(",", ":"), ensure_ascii=False).encode(), np.uint8)


@pytest.mark.parametrize("road", ["split", "fused", "auto"])
def test_token_stream_beside_the_offsets(orc, road):
    """(round 6: on BOTH roads -- the split pipeline stages the bytes in its scan kernel, the single-pass kernels gather them when they emit -- and under AUTO.)
    sjgpu_stage1_tokens_device (round 5): the bytes under the structurals leave stage 1 WITH the offsets -- tok[i] = buf[idx[i]] -- and a list pass reads
    one coalesced byte per entry instead of gathering it out of the document (the reference's consumers dereference the list the same way:
    src/generic/stage2/json_iterator.h:246-288, find_next_document_index.h:39-98).  Same list, same flags as sjgpu_stage1_device; the stream checked
    byte for byte against the document; the depth scan from the stream equal to the depth scan that gathers.  The documents take both roads of
    the emission kernel: staged by the scan kernel (segments that resolve themselves) and gathered (minified text: no control character; a patched
    candidate bit behind a 64-byte backslash run)."""
    import torch
    p = capi.DomParserImplementation(96 << 20)
    stream = torch.cuda.current_stream().cuda_stream
    ex = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsonexamples")
    tw = np.fromfile(os.path.join(ex, "twitter.json"), dtype=np.uint8)
    runs = b'["' + (b"\\" * 40 + b'\" x') * 3000 + b'", 1, "' + b"\\" * 8191 + b'\"", {"k": "' + b"\\" * 32 + b'"}]'
    docs = {"twitter.json": tw, "twitter.json minified": _minified(tw), "citm": np.fromfile(os.path.join(ex, "citm_catalog.json"), dtype=np.uint8),
            "twitter_like 24 MiB": corpus.twitter_like(24 << 20, 7)[0], "large_random 24 MiB": corpus.large_random(24 << 20, 8)[0],
            "amazon_ndjson 24 MiB": corpus.amazon_ndjson(24 << 20, 9)[0], "deep_nesting": corpus.deep_nesting(3 << 20), "escape_heavy 16 MiB": corpus.escape_heavy(16 << 20, 10)[0],
            "backslash runs across segments": np.frombuffer(runs, np.uint8), "tiny": np.frombuffer(b'{"a":[1,2,{"b":null}]}', np.uint8),
            "ends inside a string": np.frombuffer(b'[1, 2, "abc', np.uint8), "one segment and a byte": np.frombuffer(b"[" + b"1," * 8191 + b"1]", np.uint8)}
    for name, a in docs.items():
        L = len(a)
        buf = torch.from_numpy(a.copy()).cuda()
        idx = torch.full((L + 16,), -1, dtype=torch.int32, device="cuda")
        idx2 = torch.full((L + 16,), -1, dtype=torch.int32, device="cuda")
        tok = torch.full((L + 16,), 0xEE, dtype=torch.uint8, device="cuda")
        p.set_pipeline("split")
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
        n, flags, _ = p.result(stream)
        p.set_pipeline(road)
        assert p.stage1_tokens_device(buf.data_ptr(), L, idx2.data_ptr(), L + 3, tok.data_ptr(), L + 16, stream) == 0
        if road != "auto":
            assert p.last_pipeline() == road, (name, p.last_pipeline())
        n2, flags2, _ = p.result(stream)
        assert (n2, flags2) == (n, flags), (name, n, n2, flags, flags2)
        if flags & capi.F_UNESCAPED_CTRL:
            continue
        assert torch.equal(idx[: n + 3], idx2[: n + 3]), name
        oidx, _ = orc.scan(a)
        assert n == len(oidx), name
        want = torch.from_numpy(a[oidx].copy()).cuda()
        assert torch.equal(tok[:n], want), (name, first_diff(tok[:n].cpu().numpy(), want.cpu().numpy()))
        assert int(tok[n].item()) == 0xEE  # nothing behind the last token
        depth_a = torch.full((n + 1,), -99, dtype=torch.int32, device="cuda")
        depth_b = torch.full((n + 1,), -98, dtype=torch.int32, device="cuda")
        p.depth_scan_device(buf.data_ptr(), idx.data_ptr(), n, depth_a.data_ptr(), stream)
        p.depth_scan_tokens_device(tok.data_ptr(), n, depth_b.data_ptr(), stream)
        assert torch.equal(depth_a, depth_b), name
    p.close()


@pytest.mark.parametrize("road", ["split", "fused"])
def test_token_stream_at_full_size(orc, road):
    """(round 6: both roads.)  BASELINE configs[3]'s shard (1 GiB of amazon NDJSON) and configs[1] (1 GiB large_random) with the token stream: the list equal to the plain call's,
    every token byte equal to the byte of the document under its offset (checked on the device: a gather the test does once)."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    for kind in ("amazon_ndjson", "large_random"):
        a, _ = getattr(corpus, kind)(1 << 30, 77)
        L = len(a)
        p = capi.DomParserImplementation(L)
        buf = torch.from_numpy(a).cuda()
        idx = torch.empty(L // 2 + 16, dtype=torch.int32, device="cuda")
        tok = torch.empty(L // 2 + 16, dtype=torch.uint8, device="cuda")
        p.set_pipeline(road)
        assert p.stage1_tokens_device(buf.data_ptr(), L, idx.data_ptr(), L // 2, tok.data_ptr(), L // 2 + 16, stream) == 0
        n, flags, _ = p.result(stream)
        assert flags == 0 and n > 1000 and p.last_pipeline() == road
        assert torch.equal(tok[:n], buf[idx[:n].to(torch.int64)]), kind
        # ... and against the ORACLE, not only against the library's own plain call: the list's digest, and the token bytes as the oracle's list selects them
        oerr, on, oidx = orc.stage1(a, 0)
        assert oerr == 0 and on == n and orc.fnv(idx[: n + 3].cpu().numpy().view(np.uint32)) == orc.fnv(oidx), kind
        assert orc.fnv(tok[:n].cpu().numpy()) == orc.fnv(a[oidx[:n]]), kind
        del oidx
        idx2 = torch.empty(L // 2 + 16, dtype=torch.int32, device="cuda")
        p.set_pipeline("split")
        assert p.stage1_device(buf.data_ptr(), L, idx2.data_ptr(), L // 2, stream) == 0
        n2, flags2, _ = p.result(stream)
        assert (n2, flags2) == (n, flags) and torch.equal(idx[: n + 3], idx2[: n + 3]), kind
        p.close()
        del buf, idx, idx2, tok


# ---- the strings of a document, unescaped on the device -----------------------------------------------------------------
def _device_strings(p, a, allow=False):
    """-> (err, string buffer bytes, CSR offsets[n + 1], strings, first_bad, n, idx) through stage1_device + parse_strings_device"""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    L = len(a)
    buf = torch.from_numpy(np.ascontiguousarray(a).copy()).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
    n, flags, _ = p.result(stream)
    assert flags == 0
    cap = 5 * (L + 1) // 3 + 64
    out = torch.full((cap,), 0xEE, dtype=torch.uint8, device="cuda")
    off = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    err, used, strings, bad = p.parse_strings_device(buf.data_ptr(), L, idx.data_ptr(), n, out.data_ptr(), cap, off.data_ptr(), allow, stream)
    assert bool((out[used:] == 0xEE).all())  # nothing behind the last record was touched
    return err, out[:used].cpu().numpy(), off.cpu().numpy().view(np.uint32), strings, bad, n, idx[:n].cpu().numpy().view(np.uint32)


def _csr(noffsets, used):
    """the oracle's per-structural offsets (NO_STRING for the others) -> CSR form (offsets[i] = start of the next record)"""
    out = np.empty(len(noffsets) + 1, dtype=np.uint32)
    nxt = used
    out[-1] = used
    for i in range(len(noffsets) - 1, -1, -1):
        if noffsets[i] != checkers.NO_STRING:
            nxt = noffsets[i]
        out[i] = nxt
    return out


def test_string_buffer_is_the_reference_s(orc, ref):
    """document::string_buf of the reference's dom parse ([u32 length][unescaped bytes][0] per string, document order:
    src/generic/stage2/tape_builder.h:415-433) comes out of sjgpu_parse_strings_device byte for byte; the offsets are the payloads
    of the reference's tape entries."""
    p = capi.DomParserImplementation(64 << 20)
    ex = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsonexamples")
    gold = load("strings.json")["buffers"]
    docs = {"twitter.json": np.fromfile(os.path.join(ex, "twitter.json"), dtype=np.uint8), "citm_catalog.json": np.fromfile(os.path.join(ex, "citm_catalog.json"), dtype=np.uint8),
            "twitter_like 24 MiB": corpus.twitter_like(24 << 20, 5)[0], "large_random 8 MiB": corpus.large_random(8 << 20, 3)[0],
            "escape_heavy 2 MiB": corpus.escape_heavy(2 << 20)[0], "empty strings": np.frombuffer(b'["","","a",""]', np.uint8), "no strings": np.frombuffer(b"[1,2,{}]", np.uint8)}
    for name, a in docs.items():
        err, got, off, strings, bad, n, idx = _device_strings(p, a)
        oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n)
        assert (err, strings, bad) == (oerr, ostrings, obad) == (0, ostrings, checkers.NO_STRING), name
        assert np.array_equal(got, want), (name, first_diff(got, want))
        assert np.array_equal(off, _csr(ooff, len(want))), name
        if name in gold:  # the committed known answers of the reference itself
            assert (strings, len(got), orc.fnv(got)) == (gold[name]["strings"], gold[name]["bytes"], gold[name]["fnv1a64"]), name
        if ref is not None and name != "escape_heavy 2 MiB":
            rerr, rbuf, rstrings = ref.dom_string_buf(ref.best_impl(), a)
            if rerr == 0:  # a whole valid document: the live reference agrees
                assert rstrings == strings and np.array_equal(rbuf, got), name
    p.close()


def test_strings_with_every_escape_and_every_error(orc):
    """Every hand-written body of the CPU tier's vectors (all escapes, surrogate pairs, lone / unpaired surrogates, bad hex, bad
    escape characters, escapes around the 4-byte fast path and at the end of the buffer), valid and invalid ones side by side in one
    document, with and without allow_replacement: same records, same first offender, same error code as the oracle (which is pinned
    against the reference's parse_string on exactly these bodies, tests/golden/strings.json)."""
    from test_oracle_vs_reference import STRING_BODIES
    p = capi.DomParserImplementation(8 << 20)
    bodies = list(STRING_BODIES)
    for pre in (0, 1, 2, 3, 4, 5, 7, 8, 9, 30, 33, 64):
        for esc in (b'\\n', b'\\u0041', b'\\ud83d\\ude00', b'\\"', b'\\\\', b'\\q', b'\\ud800', b'\\udc00x', b'\\u00e9', b'\\u12G4'):
            bodies.append(b'a' * pre + esc + b'tail"')
    # only bodies stage 1 accepts as ONE string token followed by a separator can sit in a document
    usable = []
    for b in bodies:
        doc = np.frombuffer(b'["' + b + b',0]', np.uint8)
        e1, n, idx = orc.stage1(doc, 0)
        if e1 == 0 and n == 5 and orc.validate_utf8(doc):
            usable.append(b)
    assert len(usable) > 100
    good = [b for b in usable if orc.parse_string(b, False) is not None]
    for allow in (False, True):
        for chosen, label in ((good, "valid only"), (usable, "valid and invalid")):
            doc = np.frombuffer(b'[' + b','.join(b'"' + b for b in chosen) + b']', np.uint8)
            err, got, off, strings, bad, n, idx = _device_strings(p, doc, allow)
            oerr, want, ooff, ostrings, obad = orc.string_buffer(doc, idx, n, allow)
            assert (err, strings, bad) == (oerr, ostrings, obad), (label, allow)
            assert np.array_equal(got, want), (label, allow, first_diff(got, want))
            assert np.array_equal(off, _csr(ooff, len(want))), (label, allow)
            if label == "valid and invalid" and not allow:
                assert err == checkers.STRING_ERROR and doc[idx[bad]] == 0x22
    # the last string of a buffer: escapes whose look-ahead runs into the end of the input
    for tail in (b'["\\u12"]', b'["\\ud83d\\u"]', b'["x\\', b'["\\ud83d\\ude0'):
        doc = np.frombuffer(tail, np.uint8)
        e1, n, idx = orc.stage1(doc, 0)
        if e1 != 0:
            continue  # stage 1 already rejects it (unclosed string): stage 2 never sees it
        err, got, off, strings, bad, n, idx = _device_strings(p, doc)
        oerr, want, ooff, ostrings, obad = orc.string_buffer(doc, idx, n)
        assert (err, strings, bad) == (oerr, ostrings, obad) and np.array_equal(got, want), tail
    p.close()


def test_string_stream_takes_valid_documents_and_only_those(orc, monkeypatch):
    """The string buffer is produced as a stream compaction of the document (sjgpu_string_stream.hip) when every string is valid and
    listed, by the per-string walk otherwise -- sjgpu_debug_string_path tells which; both roads give the oracle's bytes, and forcing
    the per-string road (SJGPU_STRING_STREAM=0) changes nothing."""
    p = capi.DomParserImplementation(16 << 20)
    S = 16384
    docs = {}
    # escapes of every kind straddling the 64-byte blocks, the 4 KiB chunks and the 16 KiB segments at every phase
    escapes = [b'\\n', b'\\u0041', b'\\u00e9', b'\\u20ac', b'\\ud83d\\ude00', b'\\"', b'\\\\', b'\\/', b'\\t\\r\\b\\f', b'\\ud83d\\ude00\\ud83d\\ude00', b'\\\\\\u0041']
    for boundary in (64, 4096, S, 2 * S):
        parts = []
        for k, esc in enumerate(escapes):
            for back in range(0, 15):
                body = b'a' * (boundary - 2 - back) + esc + b'tail'
                parts.append(b'"' + body + b'"')
                parts.append(b'"' + esc + b'"')
                if len(b",".join(parts)) > 3 * boundary + 200:
                    docs[f"escapes around {boundary} ({k}, {back})"] = b'[' + b",".join(parts) + b']'
                    parts = []
        # every string of this document starts a fixed distance in front of a multiple of `boundary`
        for back in range(0, 14):
            for esc in escapes[:6]:
                lead = b'[' + b' ' * ((boundary - 1 - back - 1) % boundary)
                docs[f"one escape {esc!r} {back} bytes in front of {boundary}"] = lead + b'"' + esc + b'xyz","' + b'q' * boundary + esc + b'"]'
    docs["quotes at the segment boundary"] = b'[' + b' ' * (S - 3) + b'"","","a","",' + b' ' * (S - 20) + b'"\\"","\\\\"]'
    docs["one long string over many segments"] = b'["' + b"lorem ipsum \\\" dolor \\u00e9\\ud83d\\ude00 sit \\n amet " * 9000 + b'", "next"]'
    docs["long runs of backslashes"] = corpus.escape_heavy(300000)[0].tobytes()  # DECLINED: runs of 64 and more in front of segment boundaries
    docs["a run that fills the look-back in front of a segment"] = b'["' + b'a' * (S - 2 - 70) + b'\\' * 70 + b'n", "x"]'  # declined, too
    docs["a run of 62 in front of a segment"] = b'["' + b'a' * (S - 2 - 62) + b'\\' * 62 + b'", "x"]'  # the look-back settles it: the stream takes it
    docs["a long run behind a u within ten bytes of a segment"] = b'["' + b'a' * (S - 2 - 80 - 3) + b'\\' * 79 + b'u00' + b'41", "x"]'  # declined: is that u escaped?
    declined = {"long runs of backslashes", "a run that fills the look-back in front of a segment", "a long run behind a u within ten bytes of a segment"}
    docs["nothing but empty strings"] = b'[' + b'"",' * 20000 + b'""]'  # five bytes out of three: k_strs_write's window takes such chunks in two passes of 32 lanes
    for phase in range(0, 64, 9):
        docs[f"empty strings beside text, phase {phase}"] = (b"[" + b" " * phase + b'"",' * 900 + b'"ab\\u00e9\\n\\ud83d\\ude00c",' * 40 + b'"",' * 2100
                                                             + b'"x\\ty",' + b'"lorem \\" ipsum",' * 300 + b'"z"]')
    docs["empty keys and values"] = b'{' + b'"":"",' * 60000 + b'"k":""}'
    docs["no strings at all"] = b'[' + b'1,' * 20000 + b'2]'
    docs["twitter_like 3 MiB"] = corpus.twitter_like(3 << 20, 11)[0].tobytes()
    taken = {1: 0, 2: 0}
    for name, d in docs.items():
        a = np.frombuffer(d, np.uint8)
        if not orc.validate_utf8(a):
            continue
        results = []
        for forced in (False, True):
            if forced:
                monkeypatch.setenv("SJGPU_STRING_STREAM", "0")
            else:
                monkeypatch.delenv("SJGPU_STRING_STREAM", raising=False)
            err, got, off, strings, bad, n, idx = _device_strings(p, a)
            path = p.string_path()
            oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n)
            assert (err, strings, bad) == (oerr, ostrings, obad), (name, forced)
            assert np.array_equal(got, want), (name, forced, first_diff(got, want))
            assert np.array_equal(off, _csr(ooff, len(want))), (name, forced)
            # (round 4: a look-back that does not settle a segment's carries -- 64 backslashes in front of it -- sends the document to the
            # per-string kernels: the stream reads nothing but those 64 bytes of what lies in front of a segment)
            assert path == (2 if forced or name in declined else 1), (name, forced, path)
            taken[path] += 1
            results.append(got)
        assert np.array_equal(results[0], results[1]), name
    monkeypatch.delenv("SJGPU_STRING_STREAM", raising=False)
    assert taken[1] > 100 and taken[2] == taken[1] + 2 * len(declined)
    # documents the stream must hand over: a string the reference rejects, quotes glued to scalars (not in the list), both
    for name, d in {"bad escape": b'["ok","a\\qb","\\u00e9"]', "lone surrogate": b'["\\ud800","x"]', "bad hex behind a segment of text": b'["' + b'a' * 20000 + b'\\u12G4"]',
                    "a quote glued to a number": b'[1"abc","def"]', "glued, with a bad escape in it": b'[true"a\\qc","def","\\n"]'}.items():
        a = np.frombuffer(d, np.uint8)
        err, got, off, strings, bad, n, idx = _device_strings(p, a)
        oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n)
        assert p.string_path() == 2, name
        assert (err, strings, bad) == (oerr, ostrings, obad) and np.array_equal(got, want) and np.array_equal(off, _csr(ooff, len(want))), name
    # with replacement characters a lone surrogate is a valid string again -- but whether a HIGH surrogate keeps one byte (a pair) or three
    # (U+FFFD) depends on what FOLLOWS it, and the stream's masks look back only (round 4, sj_string_stream.h): such documents take the
    # per-string road, which knows the replacement character; well-formed pairs under the same option stay on the stream
    a = np.frombuffer(b'["\\ud800","x\\udc00\\ud83d","\\ud83d\\ude00"]', np.uint8)
    err, got, off, strings, bad, n, idx = _device_strings(p, a, True)
    oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n, True)
    assert p.string_path() == 2 and (err, strings, bad) == (oerr, ostrings, obad) == (0, 3, checkers.NO_STRING) and np.array_equal(got, want)
    a = np.frombuffer(b'["\\ud83d\\ude00","x\\u00e9\\u20ac","\\ud83d\\ude00\\n"]', np.uint8)
    err, got, off, strings, bad, n, idx = _device_strings(p, a, True)
    oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n, True)
    assert p.string_path() == 1 and (err, strings, bad) == (oerr, ostrings, obad) == (0, 3, checkers.NO_STRING) and np.array_equal(got, want)
    p.close()


def test_string_stream_random_documents(orc):
    """random valid documents and token-level mutations of them, several thousand in one NDJSON-like buffer each way"""
    import jsongen
    p = capi.DomParserImplementation(16 << 20)
    rng = np.random.default_rng(31337)
    for mutate, count in ((False, 600), (True, 600)):
        for _ in range(count):
            d = jsongen.random_document(rng, max_depth=5)
            if mutate:
                d = jsongen.mutate(rng, d)
            a = np.frombuffer(d, np.uint8)
            if len(a) == 0 or not orc.validate_utf8(a):
                continue
            e1, n1, _ = orc.stage1(a, 0)
            if e1 != 0:
                continue
            err, got, off, strings, bad, n, idx = _device_strings(p, a)
            oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n)
            assert (err, strings, bad) == (oerr, ostrings, obad), d
            assert np.array_equal(got, want), (d, first_diff(got, want))
            assert np.array_equal(off, _csr(ooff, len(want))), d
            if not mutate:
                assert p.string_path() == 1, d
    p.close()


@pytest.mark.parametrize("kind", ["twitter_like", "amazon_ndjson"])
def test_full_size_strings(orc, kind):
    """256 MiB through the device path, digest against the oracle's buffer."""
    import time
    import torch
    p = capi.DomParserImplementation(512 << 20)
    a = getattr(corpus, kind)(256 << 20, 9)[0]
    t0 = time.perf_counter()
    err, got, off, strings, bad, n, idx = _device_strings(p, a)
    dt = time.perf_counter() - t0
    oerr, want, ooff, ostrings, obad = orc.string_buffer(a, idx, n)
    assert (err, strings, bad, len(got)) == (oerr, ostrings, obad, len(want))
    assert orc.fnv(got) == orc.fnv(want)
    assert off[-1] == len(want)
    print(f"\n{kind}: {strings} strings, {len(got) / 1e6:.1f} MB of records from {len(a) / 1e6:.1f} MB ({dt * 1e3:.1f} ms incl. copies)")
    p.close()


# ---- inputs beyond one scan: pieces (sjgpu_minify / sjgpu_validate_utf8 have no length limit) -----------------------------------
def test_minify_and_validate_in_pieces(orc, monkeypatch):
    """What inputs of 4 GiB and more go through, forced onto small inputs with 1 MiB pieces: minify cut where only the
    in-string bit crosses, validate_utf8 cut in front of a character's first byte."""
    monkeypatch.setenv("SJGPU_PIECE_MB", "1")
    p = capi.DomParserImplementation(64 << 20)
    tw = corpus.twitter_like(5 << 20, 21)[0]
    M = 1 << 20
    docs = {"twitter_like 5 MiB": tw, "ends inside a string": np.concatenate([tw[: 3 << 20], np.frombuffer(b' "dangling', np.uint8)]),
            "one string across three pieces": np.frombuffer(b'["' + b"lorem ipsum \\\" dolor " * 150000 + b'", 1]', np.uint8),
            "no clean byte for two pieces": np.frombuffer(b'["' + b"x" * (2 * M + 17) + b'"] ' * 3, np.uint8)}
    for shift in range(0, 5):  # multi-byte characters straddling the piece boundary at every phase
        docs[f"4-byte characters at the boundary (-{shift})"] = np.frombuffer(b'["' + b"a" * (M - 2 - shift) + "\U0001F600".encode() * 50000 + b'"]', np.uint8)
    bad = tw.copy()
    bad[M - 1] = 0xF0
    docs["truncated lead at a piece boundary"] = bad
    bad2 = tw.copy()
    bad2[2 * M - 2: 2 * M + 3] = 0x80
    docs["five continuation bytes across a boundary"] = bad2
    for name, a in docs.items():
        gerr, gout = p.minify(a)
        oerr, oout = orc.minify(a)
        assert gerr == oerr and np.array_equal(gout, oout), (name, gerr, oerr, len(gout), len(oout))
        assert p.validate_utf8(a) == orc.validate_utf8(a), name
        for piece in (64, 100_003, 3 << 20):  # sjgpu_validate_utf8_pieces: the piece size in the caller's hand (what the plug-in retries with)
            assert p.validate_utf8(a[: 2 * M + 4096], piece_bytes=piece) == orc.validate_utf8(a[: 2 * M + 4096]), (name, piece)
    p.close()


def test_huge_inputs_have_no_length_limit(orc):
    """validate_utf8 / minify of more than 4 GiB - 1 bytes (the reference's free functions take any size_t)."""
    size = int(os.environ.get("SJGPU_HUGE_FREE_SIZE", str((4 << 30) + (64 << 20))))
    if size == 0:
        pytest.skip("disabled")
    a = corpus.amazon_ndjson(size, 41)[0]
    assert len(a) > 0xFFFFFFFF
    p = capi.DomParserImplementation(1 << 20)  # the parser's capacity does not bound the free functions
    assert p.validate_utf8(a)
    gerr, gout = p.minify(a)
    assert gerr == 0
    step = 1 << 30
    at = 0
    for b in range(0, len(a), step):  # the oracle piece by piece at line boundaries (NDJSON: every line is a document)
        e = min(b + step, len(a))
        while e < len(a) and a[e - 1] != 0x0A:
            e += 1
        if b > 0:
            while a[b - 1] != 0x0A:
                b += 1
        oerr, oout = orc.minify(a[b:e])
        assert oerr == 0 and np.array_equal(gout[at: at + len(oout)], oout), b
        at += len(oout)
    assert at == len(gout)
    a[len(a) // 2] = 0xFF
    assert not p.validate_utf8(a)
    p.close()


def test_page_locked_host_memory():
    """sjgpu_host_alloc / sjgpu_host_register: documents and index arrays in page-locked memory (include/sjgpu.h)."""
    import ctypes
    L = capi.load_library()
    a, _ = corpus.twitter_like(3 << 20, 5)
    n = len(a)
    ptr = L.sjgpu_host_alloc(n)
    assert ptr
    ctypes.memmove(ptr, a.ctypes.data, n)
    pinned = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))
    p = capi.DomParserImplementation(n)
    capi.host_register(p.structural_indexes)
    err = p.stage1(pinned, 0)
    want = checkers.Oracle().stage1(a, 0)
    assert (err, p.n_structural_indexes) == want[:2] and np.array_equal(p.structural_indexes[: want[1] + 3], want[2])
    capi.host_unregister(p.structural_indexes)
    p.close()
    L.sjgpu_host_free(ptr)
    assert L.sjgpu_host_register(None, 10) == -4 and L.sjgpu_host_unregister(None) == -4


def test_capacity_and_empty_guards(gpu):
    small = capi.DomParserImplementation(16)
    assert small.stage1(b"[1,2,3,4,5,6,7,8,9,10]") == capi.CAPACITY
    assert small.stage1(b"") == capi.EMPTY
    assert small.stage1(b"[1]") == capi.SUCCESS and small.n_structural_indexes == 3
    assert small.minify(b"")[0] == capi.SUCCESS
    assert small.validate_utf8(b"")
    small.close()


def test_reference_examples(gpu, orc):
    """BASELINE.json configs[0]: the reference's real example files (fixtures under tests/golden/jsonexamples), every
    stage-1 mode + minify + validate_utf8 against the oracle, and twitter.json's known answer: n = 55 263 (SURVEY
    App. B) and the FNV-1a-64 of the n+3 index words as the reference's icelake / haswell / westmere kernels produce
    them (tests/golden/corpora.json, made by make_golden.py; the digest printed in SURVEY App. B is not reproducible
    from the reference, its n and minified length are)."""
    ex = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsonexamples")
    for fn in ("twitter.json", "citm_catalog.json", "amazon_cellphones.ndjson"):
        assert_same_all(gpu, orc, np.fromfile(os.path.join(ex, fn), dtype=np.uint8), tag=fn, modes=tuple(range(7)))
    tw = np.fromfile(os.path.join(ex, "twitter.json"), dtype=np.uint8)
    err, n, idx = g_stage1(gpu, tw, 0)
    assert (err, n, len(tw)) == (0, 55263, 631515) and orc.fnv(idx) == 9964107509431273939


# ---- shards of ONE document (sjgpu.h "one large document sharded across GPUs") ----------------------
def _scan_in_shards(p, a, parts, with_minify=True):
    """What G ranks do, replayed on one GPU: clean cuts, parity pre-pass per shard, XOR-prefix of the bits, shard
    scans with the carried in-string bit.  -> (global offsets, per-shard flags, concatenated minified bytes)."""
    import torch
    from simdjson_amd import sharded
    stream = torch.cuda.current_stream().cuda_stream
    cuts = sharded.clean_cuts(a, parts)
    assert cuts[0] == 0 and cuts[-1] == len(a) and cuts == sorted(cuts)
    state, offs, flags, mini = 0, [], [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi == lo:
            flags.append(1 if state else 0)
            continue
        L = hi - lo
        buf = torch.from_numpy(np.ascontiguousarray(a[lo:hi])).cuda()  # its own (aligned) allocation, like a rank's
        parity = p.string_parity_device(buf.data_ptr(), L, stream)
        idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
        p.stage1_shard_device(buf.data_ptr(), L, state, idx.data_ptr(), L + 3, stream)
        n, f, _ = p.result(stream)
        assert f & (capi.F_INTERNAL | capi.F_IDX_OVERFLOW) == 0
        assert (f & 1) == (state ^ parity), "carry-out = carry-in xor parity"
        host = idx[: n + 3].cpu().numpy().view(np.uint32)
        assert list(host[n:]) == [L, L, 0]
        offs.append(host[:n].astype(np.int64) + lo)
        flags.append(f)
        if with_minify:
            dst = torch.empty(L + 16, dtype=torch.uint8, device="cuda")
            p.minify_shard_device(buf.data_ptr(), L, state, dst.data_ptr(), stream)
            _, mf, mlen = p.result(stream)
            assert (mf & 1) == (state ^ parity)
            mini.append(dst[:mlen].cpu().numpy())
        state ^= parity
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(offs, np.int64), flags, cat(mini, np.uint8)


def test_document_shards_equal_the_whole_scan(gpu, orc):
    from simdjson_amd import sharded
    rng = np.random.default_rng(5)
    docs = {
        "twitter_like 3 MiB": corpus.twitter_like(3 << 20, 3)[0],
        "large_random 2 MiB": corpus.large_random(2 << 20, 3)[0],
        "twitter_like 20 MiB": corpus.twitter_like(20 << 20, 4)[0],
        "one long string with blanks": np.frombuffer(b'["' + b"lorem ipsum, {dolor}: [sit] \\\" amet " * 40000 + b'", 1]', np.uint8),
        "no clean byte for a long stretch": np.frombuffer(b'["' + b"x" * 300000 + b'", "' + b"y" * 300000 + b'"]', np.uint8),
        "ends inside a string": np.concatenate([corpus.twitter_like(1 << 20, 6)[0], np.frombuffer(b' "dangling text', np.uint8)]),
    }
    ctrl = corpus.twitter_like(1 << 20, 7)[0].copy()
    quotes = np.flatnonzero(ctrl == ord('"'))
    ctrl[int(quotes[len(quotes) // 2]) + 1] = 0x02
    docs["control character inside a string"] = ctrl
    bad = corpus.twitter_like(1 << 20, 8)[0].copy()
    bad[len(bad) // 3] = 0xFF
    docs["invalid UTF-8"] = bad
    alphabet = np.frombuffer(b'"\\ ,:[]{}ab1\n\t\xc3\xa9\xe2\x82\xac', np.uint8)
    docs["random soup (control characters in strings)"] = alphabet[rng.integers(0, len(alphabet), 200000)].copy()
    alphabet = alphabet[(alphabet > 0x1F)]
    docs["random soup"] = alphabet[rng.integers(0, len(alphabet), 200000)].copy()
    for name, a in docs.items():
        whole, wflags = orc.scan(a)
        oerr, omin = orc.minify(a)
        for parts in (2, 3, 8):
            offs, flags, mini = _scan_in_shards(gpu, a, parts)
            assert sharded.document_flags(flags) == wflags, (name, parts, flags, wflags)
            if wflags & capi.F_UNESCAPED_CTRL:
                continue  # UNESCAPED_CHARS: the reference returns before publishing n, the offsets are unobservable
            assert np.array_equal(offs, whole.astype(np.int64)), (name, parts, first_diff(offs, whole.astype(np.int64)))
            if oerr == 0:  # an unclosed string voids the whole document's output (json_minifier.h:42-47)
                assert np.array_equal(mini, omin), (name, parts, first_diff(mini, omin))
            else:
                assert sharded.document_flags(flags) & 1


def test_full_size_document_in_eight_shards(orc):
    """BASELINE.json configs[1] cut into 8 shards (what 8 GPUs would hold), AUTO pipeline per shard."""
    import torch
    from simdjson_amd import sharded
    size = int(os.environ.get("SJGPU_FULL_SIZE", str(1 << 30)))
    a, _ = corpus.large_random(size, 11)
    p = capi.DomParserImplementation(len(a) // 8 + (1 << 20))
    offs, flags, _ = _scan_in_shards(p, a, 8, with_minify=False)
    oerr, on, oidx = orc.stage1(a, 0)
    assert oerr == 0 and sharded.document_flags(flags) == 0 and len(offs) == on
    assert orc.fnv(offs.astype(np.uint32)) == orc.fnv(oidx[:on]), first_diff(offs, oidx[:on].astype(np.int64))
    p.close()


# ---- ranges of one resident buffer + the overlapped host-buffer path built on them --------------
def _range_docs():
    tw = corpus.twitter_like(5 << 20, 12)[0]
    docs = {
        "twitter_like 5 MiB (multi-byte UTF-8 and escapes at the range boundaries)": tw,
        "large_random 3.5 MiB": corpus.large_random((7 << 20) // 2, 12)[0],
        "ends inside a string": np.concatenate([tw[: 3 << 20], np.frombuffer(b' "dangling text', np.uint8)]),
    }
    # the 1 MiB boundary inside a string, inside a backslash run (odd and even), inside a 4-byte character
    M = 1 << 20
    for name, filler in (("backslash run across the boundary, odd", b"\\" * 7), ("backslash run across the boundary, even", b"\\" * 8),
                         ("4-byte character across the boundary", "\U0001F600".encode()), ("quote exactly at the boundary", b'"')):
        for shift in (0, 1, 2, 3):
            head = b'["' + b"a" * (M - 2 - shift)
            body = filler + (b"" if filler == b'"' else b'x"') + b', "tail", {"k": [1, 2, 3]}, "' + b"b" * (M + 77) + b'"]'
            docs[f"{name} (-{shift})"] = np.frombuffer(head + body, np.uint8)
    bad = tw.copy()
    bad[(2 << 20) - 1] = 0xF0  # truncated 4-byte lead just before a range boundary
    docs["invalid UTF-8 at a range boundary"] = bad
    return docs


def test_ranges_equal_the_whole_scan(gpu, orc):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    for name, a in _range_docs().items():
        L = len(a)
        whole, wflags = orc.scan(a)
        oerr, omin = orc.minify(a)
        buf = torch.from_numpy(a.copy()).cuda()
        for chunk in (1 << 20, 3 << 20):
            idx = torch.full((L + 3,), -1, dtype=torch.int32, device="cuda")
            dst = torch.zeros(L + 16, dtype=torch.uint8, device="cuda")
            n = out = s_idx = s_min = flags = 0
            for b in range(0, L, chunk):
                e = min(b + chunk, L)
                gpu.stage1_range_device(buf.data_ptr(), b, e, e < L, s_idx, n, idx.data_ptr(), L + 3, stream)
                n, f, _ = gpu.result(stream)
                assert f & (capi.F_INTERNAL | capi.F_IDX_OVERFLOW) == 0
                flags |= f & ~(1 | capi.F_RANGE_CARRY)
                s_idx = f & (1 | capi.F_RANGE_CARRY)  # what a range hands to the next one: the in-string bit and the escape carry
                gpu.minify_range_device(buf.data_ptr(), b, e, e < L, s_min, out, dst.data_ptr(), stream)
                _, mf, out = gpu.result(stream)
                s_min = mf & (1 | capi.F_RANGE_CARRY)
            assert (flags | (s_idx & 1)) == wflags and (s_min & 1) == (wflags & 1), (name, chunk, flags, s_idx, wflags)
            if not (wflags & capi.F_UNESCAPED_CTRL):
                host = idx[: n + 3].cpu().numpy().view(np.uint32)
                assert n == len(whole) and np.array_equal(host[:n], whole), (name, chunk, first_diff(host[:n], whole))
                assert list(host[n:]) == [L, L, 0]
            if oerr == 0:
                got = dst[:out].cpu().numpy()
                assert np.array_equal(got, omin), (name, chunk, first_diff(got, omin))


def test_backslash_runs_across_range_boundaries(orc, monkeypatch):
    """A backslash run that crosses a short-then-long range boundary, and a streamed document whose tail range is short:
    every range of a larger buffer builds the escape table (a short range used to skip it, its look-back then
    walked over all earlier ranges, and a later range could resolve a pass entry through an entry nobody had written)."""
    import time
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    M = 1 << 20
    # ranges of 2 MiB (short: no table of their own before the fix) then 10 MiB; a 96 KiB backslash run straddles the cut
    head = b'["' + b"a" * (2 * M - 2 - 40000)
    body = b"\\" * 98305 + b'x", "tail", ' + b'{"k": [1, 2, 3]}, ' * 600000 + b'"end"]'
    a = np.frombuffer(head + body, np.uint8)
    L = len(a)
    whole, wflags = orc.scan(a)
    buf = torch.from_numpy(a.copy()).cuda()
    for pipeline in ("fused", "split"):
        p = _parser(pipeline, 16 << 20)
        idx = torch.full((L + 16,), -1, dtype=torch.int32, device="cuda")
        n = s = flags = 0
        for b, e in ((0, 2 * M), (2 * M, L)):
            p.stage1_range_device(buf.data_ptr(), b, e, e < L, s, n, idx.data_ptr(), L + 3, stream)
            n, f, _ = p.result(stream)
            flags |= f & ~(1 | capi.F_RANGE_CARRY)
            s = f & (1 | capi.F_RANGE_CARRY)
        assert (flags | (s & 1)) == wflags and n == len(whole)
        assert np.array_equal(idx[:n].cpu().numpy().view(np.uint32), whole), pipeline
        p.close()
    # a 72 MiB document that is one backslash run, streamed in 16 MiB ranges: the 8 MiB tail range must not walk
    monkeypatch.setenv("SJGPU_STREAM_FROM_MB", "1")
    doc = np.frombuffer(b'["' + b"\\" * ((72 << 20) - 8) + b'", 7]', np.uint8)
    want = orc.stage1(doc, 0)
    p = capi.DomParserImplementation(len(doc))
    p.stage1(doc, 0)
    t0 = time.perf_counter()
    err = p.stage1(doc, 0)
    dt = time.perf_counter() - t0
    assert (err, p.n_structural_indexes) == want[:2] and np.array_equal(p.structural_indexes[: want[1] + 3], want[2])
    assert dt < 0.5, f"72 MiB of backslashes through the streamed path took {dt * 1e3:.0f} ms"
    p.close()


def test_overlapped_host_path(orc, monkeypatch):
    """sjgpu_stage1 / sjgpu_minify with host buffers, forced through the range-by-range path with 1 MiB ranges
    (and once with the defaults on a document large enough to take it by itself)."""
    monkeypatch.setenv("SJGPU_STREAM_FROM_MB", "1")
    monkeypatch.setenv("SJGPU_STREAM_CHUNK_MB", "1")
    for pipeline in ("auto", "fused", "split"):
        p = capi.DomParserImplementation(8 << 20)
        p.set_pipeline(pipeline)
        for name, a in _range_docs().items():
            assert_same_all(p, orc, a, f"streamed {pipeline} {name}")
        nd = corpus.amazon_ndjson(3 << 20, 5)[0]
        for mode in checkers.MODES.values():
            assert_same_stage1(p, orc, nd, mode, f"streamed {pipeline} ndjson")
            assert_same_stage1(p, orc, nd[:-100], mode, f"streamed {pipeline} ndjson cut")
        p.close()
    # ranges above the small-tile limit with the single-pass pipeline forced: the pipelined kernel with begin > 0
    monkeypatch.setenv("SJGPU_STREAM_CHUNK_MB", "9")
    big = corpus.twitter_like(30 << 20, 14)[0]
    for pipeline in ("fused", "split"):
        p = capi.DomParserImplementation(len(big))
        p.set_pipeline(pipeline)
        assert_same_all(p, orc, big, f"streamed {pipeline}, 9 MiB ranges")
        assert_same_all(p, orc, np.concatenate([big[: 20 << 20], np.frombuffer(b' "dangling', np.uint8)]), f"streamed {pipeline}, unclosed")
        p.close()
    monkeypatch.delenv("SJGPU_STREAM_FROM_MB")
    monkeypatch.delenv("SJGPU_STREAM_CHUNK_MB")
    a = corpus.twitter_like(40 << 20, 13)[0]
    p = capi.DomParserImplementation(len(a))
    assert_same_all(p, orc, a, "streamed, default thresholds, 40 MiB")
    # too small an index array is reported, not overrun
    import ctypes
    small = np.zeros(1000, np.uint32)
    n, nxt = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = p.L.sjgpu_stage1(p.h, a.ctypes.data, len(a), 0, small.ctypes.data, len(small), ctypes.byref(n), ctypes.byref(nxt))
    assert rc == -5  # SJGPU_E_OVERFLOW
    p.close()


# ---- full size (BASELINE.json configs 2/3): device-resident path, 1 GiB ------------------------------------------
@pytest.mark.parametrize("pipeline", ["fused", "split"])
@pytest.mark.parametrize("kind", ["large_random", "amazon_ndjson"])
def test_full_size_device_resident(orc, kind, pipeline):
    import torch
    size = int(os.environ.get("SJGPU_FULL_SIZE", str(1 << 30)))
    a, _ = getattr(corpus, kind)(size, 11)
    L = len(a)
    p = capi.DomParserImplementation(L)
    p.set_pipeline(pipeline)
    buf = torch.from_numpy(a).cuda()
    idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
    n, flags, _ = p.result(stream)
    assert flags & capi.F_INTERNAL == 0
    oerr, on, oidx = orc.stage1(a, 0)
    assert capi.stage1_error_from_flags(n, flags) == oerr == 0 and n == on
    # properties: strictly ascending, in range, sentinels
    got = idx[: n + 3]
    assert bool((got[1:n] > got[: n - 1]).all()) and int(got[n - 1]) < L
    assert [int(x) & 0xFFFFFFFF for x in got[n:]] == [L, L, 0]
    # exact: digest of all n+3 words against the oracle's
    host = got.cpu().numpy().view(np.uint32)
    assert orc.fnv(host) == orc.fnv(oidx), first_diff(host, oidx)
    del idx, got
    # minify: exact vs oracle, idempotent, structural count preserved
    dst = torch.empty(L + 16, dtype=torch.uint8, device="cuda")
    assert p.minify_device(buf.data_ptr(), L, dst.data_ptr(), stream) == 0
    _, mflags, mlen = p.result(stream)
    oerr, oout = orc.minify(a)
    assert oerr == 0 and mflags & 1 == 0 and mlen == len(oout)
    mhost = dst[:mlen].cpu().numpy()
    assert orc.fnv(mhost) == orc.fnv(oout), first_diff(mhost, oout)
    dst2 = torch.empty(L + 16, dtype=torch.uint8, device="cuda")
    mbuf = dst[:mlen].clone()
    assert p.minify_device(mbuf.data_ptr(), mlen, dst2.data_ptr(), stream) == 0
    _, mflags2, mlen2 = p.result(stream)
    assert (mlen2, mflags2) == (mlen, 0) and bool((dst2[:mlen] == mbuf).all()), (mlen2, mlen, mflags2)
    idx2 = torch.empty(mlen + 3, dtype=torch.int32, device="cuda")
    assert p.stage1_device(mbuf.data_ptr(), mlen, idx2.data_ptr(), mlen + 3, stream) == 0
    n2, flags2, _ = p.result(stream)
    assert n2 == n and flags2 == 0
    # utf8: valid; one corrupted byte anywhere makes it invalid
    assert p.validate_utf8_device(buf.data_ptr(), L, stream) == 0
    assert p.result(stream)[1] & capi.F_UTF8_ERROR == 0
    for pos in (0, 4095, 4096, L // 2 + 63, L - 1):
        old = int(buf[pos])
        buf[pos] = 0xFF
        assert p.validate_utf8_device(buf.data_ptr(), L, stream) == 0
        assert p.result(stream)[1] & capi.F_UTF8_ERROR, pos
        buf[pos] = old
    p.close()


@pytest.mark.parametrize("kind", ["deep_nesting", "escape_heavy"])
def test_full_size_adversarial(orc, kind):
    """BASELINE.json configs[4] at full size: density 1.0 (4 bytes out per byte in) and escape carries across every
    boundary type; exact digest of all n+3 words against the oracle, both pipelines."""
    import torch
    size = int(os.environ.get("SJGPU_FULL_SIZE", str(1 << 30)))
    gen = corpus.deep_nesting_doc if kind == "deep_nesting" else corpus.escape_heavy
    a, _ = gen(size, 1000)
    L = len(a)
    oerr, on, oidx = orc.stage1(a, 0)
    assert oerr == 0
    want = orc.fnv(oidx)
    del oidx
    buf = torch.from_numpy(a).cuda()
    idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for pipeline in ("fused", "split"):
        p = capi.DomParserImplementation(L)
        p.set_pipeline(pipeline)
        idx.zero_()
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
        n, flags, _ = p.result(stream)
        assert (n, flags) == (on, 0), (pipeline, n, on, flags)
        host = idx[: n + 3].cpu().numpy().view(np.uint32)
        assert orc.fnv(host) == want, pipeline
        del host
        p.close()


# ---- long backslash runs: the escape table (k_escape_local / k_escape_resolve) --------------------------------------------
def _long_run_document(rng, total):
    """Strings that are backslash runs of 16 KiB-ish to MiB-ish lengths, each placed at a chosen distance from a 16 KiB
    segment boundary (so runs start / end exactly on, just before and just after boundaries, and some segments are
    nothing but backslashes), separated by ordinary small values."""
    SEG = 16384
    parts, at = [b"["], 1
    lengths = [SEG - 1, SEG, SEG + 1, 2 * SEG - 1, 2 * SEG, 2 * SEG + 1, 3 * SEG, 5 * SEG + 1, 100000, 100001,
               (1 << 20) + 1, 1 << 20, 4 << 20, (4 << 20) + 1, 63, 64, 65, 4095, 4096, 4097]
    k = 0
    while at < total:
        r = lengths[k % len(lengths)]
        k += 1
        # pad so that the run starts at (boundary + delta)
        delta = int(rng.choice([-2, -1, 0, 1, 2, 63, 64, 4095, 8000]))
        pad = (-(at + 1) + delta) % SEG
        body = b" " * pad + b'"' + b"\\" * r + (b'"' if r % 2 == 0 else b'""') + b', {"k": [1, 2]}, "x\\"y", '
        parts.append(body)
        at += len(body)
    parts.append(b"0]")
    return np.frombuffer(b"".join(parts), np.uint8)


def test_long_backslash_runs_cost_no_walk(orc, monkeypatch):
    import time
    import torch
    rng = np.random.default_rng(99)
    docs = {"runs around segment boundaries, 40 MiB": _long_run_document(rng, 40 << 20),
            "escape_heavy 24 MiB": corpus.escape_heavy(24 << 20, 3)[0],
            "one 32 MiB backslash run in a string": np.frombuffer(b'["' + b"\\" * (32 << 20) + b'", 1]', np.uint8),
            "odd 32 MiB run: the string stays open": np.frombuffer(b'["' + b"\\" * ((32 << 20) + 1) + b'", 1]', np.uint8)}
    for pipeline in ("fused", "split"):
        p = capi.DomParserImplementation(64 << 20)
        p.set_pipeline(pipeline)
        for name, a in docs.items():
            assert_same_all(p, orc, a, f"{pipeline} {name}")
        p.close()
    # range by range with ranges above the small-tile limit: the table of range k continues the table of range k-1
    monkeypatch.setenv("SJGPU_STREAM_FROM_MB", "1")
    monkeypatch.setenv("SJGPU_STREAM_CHUNK_MB", "9")
    p = capi.DomParserImplementation(64 << 20)
    for name, a in docs.items():
        assert_same_all(p, orc, a, f"streamed {name}")
    p.close()
    monkeypatch.delenv("SJGPU_STREAM_FROM_MB")
    monkeypatch.delenv("SJGPU_STREAM_CHUNK_MB")
    # the point of the table: one huge run must not cost a look-back per segment that is as long as the run
    a = np.frombuffer(b'["' + b"\\" * (256 << 20) + b'"]', np.uint8)
    L = len(a)
    want, wflags = orc.scan(a)
    assert wflags == 0 and list(want) == [0, 1, L - 1]
    buf = torch.from_numpy(a.copy()).cuda()
    idx = torch.empty(1024, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for pipeline in ("fused", "split"):
        p = capi.DomParserImplementation(L)
        p.set_pipeline(pipeline)
        p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), 1024, stream)
        assert p.result(stream)[:2] == (3, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), 1024, stream)
        n, flags, _ = p.result(stream)
        dt = time.perf_counter() - t0
        assert (n, flags) == (3, 0) and [int(x) for x in idx[:3]] == [0, 1, L - 1]
        assert dt < 0.05, f"{pipeline}: 256 MiB of backslashes took {dt * 1e3:.1f} ms"
        p.close()


# ---- one host buffer, several GPUs, one process (sjgpu_mgpu.hip; a device may be listed more than once) ------------------------
def test_mgpu_shards_equal_the_single_scan(orc):
    tw = corpus.twitter_like(6 << 20, 31)[0]
    docs = {"twitter_like 6 MiB": tw, "amazon_ndjson 5 MiB": corpus.amazon_ndjson(5 << 20, 31)[0],
            "ends inside a string": np.concatenate([tw[: 2 << 20], np.frombuffer(b' "dangling text', np.uint8)]),
            "no clean byte for a long stretch": np.frombuffer(b'["' + b"x" * 3000000 + b'", "' + b"y" * 300000 + b'"]', np.uint8),
            "tiny": np.frombuffer(b'{"a":[1,2,3]}', np.uint8), "one byte": np.frombuffer(b"7", np.uint8)}
    bad = tw.copy()
    bad[len(bad) // 2] = 0xFF
    docs["invalid UTF-8"] = bad
    ctrl = tw.copy()
    quotes = np.flatnonzero(ctrl == ord('"'))
    ctrl[int(quotes[len(quotes) // 3]) + 1] = 0x01
    docs["control character inside a string"] = ctrl
    for shards in (2, 3, 8):
        m = capi.MultiGpu([0] * shards)
        for name, a in docs.items():
            modes = range(7) if len(a) < (3 << 20) or shards == 3 else (0, 1, 2)
            for mode in modes:
                err = m.stage1(a, mode)
                got = checkers.observable(a, mode, err, m.n_structural_indexes, m.structural_indexes[: m.n_structural_indexes + 3])
                want = checkers.observable(a, mode, *orc.stage1(a, mode))
                assert got == want, (name, shards, mode, got[:2], want[:2])
            gerr, gout = m.minify(a)
            oerr, oout = orc.minify(a)
            assert gerr == oerr and np.array_equal(gout, oout), (name, shards, gerr, oerr)
            assert m.validate_utf8(a) == orc.validate_utf8(a), (name, shards)
        m.close()


# ---- RCCL, world size 1: the collective path of the N-GPU bench with the real backend ----------------------------------------------------
def _rccl_worker(port, q):
    """One rank over the nccl (= RCCL) backend on the one GPU of the box: the shard scan through the HIP library, the variable-length
    gather to the root and the padded all-gather exactly as `bench.py --gpus N` runs them, compared with the oracle's scan."""
    import torch
    import torch.distributed as dist
    from simdjson_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        buf, _ = corpus.amazon_ndjson(24 << 20, 31)
        scanner = sharded.GpuShardScanner(len(buf), device=0)
        local = sharded.scan_shard(buf, 0, 1, scanner)
        positions, counts, flags = sharded.gather_global_indices(local)
        rooted, rcounts = sharded.gather_to_root(local)
        whole, wflags = checkers.Oracle().scan(buf)
        ok = (flags == wflags == 0 and counts == rcounts == [len(whole)]
              and np.array_equal(positions.cpu().numpy(), whole.astype(np.int64)) and np.array_equal(rooted.cpu().numpy(), whole.astype(np.int64)))
        q.put((ok, counts, len(whole)))
    except Exception as e:  # the parent wants a verdict, not a hang
        q.put((False, repr(e), 0))
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_one_smoke():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    ok, counts, total = q.get(timeout=300)
    p.join(timeout=60)
    assert ok, (counts, total)


# ---- stage 2 on the device -- the DOM tape (sjgpu_tape.hip) -------------------------------------------------------------
import jsongen


@pytest.fixture(scope="module")
def tape_parser():
    build.build_sjgpu()
    p = capi.DomParserImplementation(CAP)
    yield p
    p.close()


def _checker_parse(orc, ref):
    """the live reference's dom::parser::parse where its library travelled along, else the oracle's (pinned against it on the CPU tier)"""
    if ref is not None:
        impl = ref.best_impl()
        return lambda d, md=1024: ref.dom_parse(impl, d, md)
    return lambda d, md=1024: orc.dom_parse(d, md)


def _assert_same_parse(p, want_parse, doc, max_depth=1024):
    e_want, t_want, s_want = want_parse(doc, max_depth)
    e_got, t_got, s_got = p.parse(doc, max_depth)
    assert e_got == e_want, (bytes(doc[:200]), e_got, e_want)
    if e_want == 0:
        assert len(t_got) == len(t_want) and np.array_equal(t_got, t_want), (bytes(doc[:200]), first_diff(t_got, t_want))
        assert bytes(s_got) == bytes(s_want), bytes(doc[:200])
    return e_want


@pytest.mark.parametrize("name", ["twitter.json", "citm_catalog.json"])
def test_tape_of_the_reference_s_files(tape_parser, orc, ref, name):
    """dom::parser::parse of the reference's own example files: doc.tape and doc.string_buf word for word"""
    doc = np.fromfile(os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples", name), dtype=np.uint8)
    assert _assert_same_parse(tape_parser, _checker_parse(orc, ref), doc) == 0


def test_tape_of_random_documents(tape_parser, orc, ref):
    want = _checker_parse(orc, ref)
    rng = np.random.default_rng(515)
    for _ in range(1500):
        assert _assert_same_parse(tape_parser, want, jsongen.random_document(rng)) == 0


def test_tape_errors_are_the_reference_s(tape_parser, orc, ref):
    """token-level mutations: the device finds the token the serial walk stops at (same error_code), without walking"""
    want = _checker_parse(orc, ref)
    rng = np.random.default_rng(616)
    seen = {}
    for _ in range(5000):
        doc = jsongen.mutate(rng, jsongen.random_document(rng, max_depth=4))
        if len(doc) == 0:
            continue
        e = _assert_same_parse(tape_parser, want, doc)
        seen[e] = seen.get(e, 0) + 1
    for code in (0, 3, 5, 6, 7, 8, 9, 10):
        assert seen.get(code, 0) > 0, seen


def test_tape_numbers(tape_parser, orc, ref):
    """every corner-case number (rounding boundaries, subnormals, range limits, 800-digit texts) alone and inside containers; the valid
    ones also as ONE array, so that the big-integer kernel sees many tokens at once"""
    want = _checker_parse(orc, ref)
    good = []
    for text in jsongen.number_corner_cases():
        t = text.encode()
        e = _assert_same_parse(tape_parser, want, b"[" + t + b"]")
        _assert_same_parse(tape_parser, want, t)
        _assert_same_parse(tape_parser, want, b'{"k":' + t + b" }")
        if e == 0:
            good.append(t)
    assert len(good) > 500
    assert _assert_same_parse(tape_parser, want, b"[" + b",\n".join(good) + b"]") == 0


def test_tape_staged_token_front(tape_parser, orc, ref):
    """k_tok_stage's groups on the hardware (round 6; the emulator runs the same documents through the same source, tests/test_tape_emu.py): tokens packed
    densely and far apart in one list -- staged groups beside rows read in place --, numbers and atoms at the last bytes of a window and of the document,
    numbers longer than the 64 bytes staged behind a group's last token (parsed again from memory), and the same places broken."""
    import test_tape_emu
    want = _checker_parse(orc, ref)
    rng = np.random.default_rng(14)
    docs = test_tape_emu._staged_front_documents(rng)
    valid = sum(1 for d in docs if _assert_same_parse(tape_parser, want, np.frombuffer(d, dtype=np.uint8)) == 0)
    assert valid >= 30, valid
    seen = {}
    for d in docs[:40]:
        for _ in range(3):
            m = jsongen.mutate(rng, d)
            if len(m):
                e = _assert_same_parse(tape_parser, want, np.frombuffer(bytes(m), dtype=np.uint8))
                seen[e] = seen.get(e, 0) + 1
    assert len(seen) >= 2, seen


def test_tape_depth_limits(tape_parser, orc, ref):
    want = _checker_parse(orc, ref)
    for max_depth in (1, 2, 3, 16, 1024):
        for depth in (1, 2, 3, 15, 16, 17, 1023, 1024, 1025):
            for inner in (b"", b"1", b"{}", b'{"a":[]}'):
                _assert_same_parse(tape_parser, want, b"[" * depth + inner + b"]" * depth, max_depth)
                _assert_same_parse(tape_parser, want, b'{"a":' * depth + (inner or b"0") + b"}" * depth, max_depth)
    for doc in (b"[,]", b"[ ,1]", b'{"a":,}', b"[1,,2]", b'{"a":1,,}', b'{"a":1 "b":2}', b'{"a" "b"}', b'{"a"}', b"[1 2]", b"[}", b"{]", b"[1}", b'{"a":1]', b"]", b"}", b"[]]",
                b"[[]", b"[[1]", b'{"a":{}', b"[", b"{", b'"a" "b"', b"1 2", b"[] []", b"nul", b"tru", b"fals", b"truex", b"[truex]", b"!", b"[!]", b"[1]x", b"x", b'{x:1}',
                b"[:]", b"[1:2]", b'{"a"::1}', b'{,}', b'["\\q"]', b'{"\\q":1}', b'["a","\\ud800"]', b"[-]", b"[0123]", b"{}", b"[]", b"0", b'""', b"null"):
        _assert_same_parse(tape_parser, want, doc)


@pytest.mark.parametrize("kind", ["large_random", "twitter_like"])
def test_tape_of_64_mib_documents(tape_parser, orc, ref, kind):
    """tens of millions of tokens: every scan, both radix passes and the match at scale; compared word for word"""
    a, _ = getattr(corpus, kind)(64 << 20, 2026)
    assert _assert_same_parse(tape_parser, _checker_parse(orc, ref), a) == 0


def test_tape_of_wide_and_deep_documents(tape_parser, orc, ref):
    """a container with more than 0xFFFFFF members (saturating count, tape_builder.h:402-404), nesting at the limit, and a document
    whose brackets span all 64 radix digits and more"""
    want = _checker_parse(orc, ref)
    n = 0xFFFFFF + 3
    assert _assert_same_parse(tape_parser, want, b"[" + b"1," * (n - 1) + b"1]") == 0
    rng = np.random.default_rng(5)
    parts = []
    for d in rng.integers(1, 900, 400):
        parts.append(b"[" * int(d) + b'{"k":[1,2,{"z":null}]}' + b"]" * int(d))
    assert _assert_same_parse(tape_parser, want, b"[" + b",".join(parts) + b"]") == 0


def test_stage2_device_entry_point(tape_parser, orc, ref):
    """the device-resident form: list and document stay in HBM, tape and string buffer are written there"""
    import torch
    want = _checker_parse(orc, ref)
    a, _ = corpus.twitter_like(8 << 20, 99)
    L = len(a)
    buf = torch.from_numpy(a).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
    sbuf = torch.empty(5 * (L // 3) + 256, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert tape_parser.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
    n, flags, _ = tape_parser.result(st)
    assert flags == 0
    err, tw, sb = tape_parser.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), sbuf.numel(), 1024, st)
    e_want, t_want, s_want = want(a)
    assert (err, tw, sb) == (e_want, len(t_want), len(s_want))
    assert np.array_equal(tape[:tw].cpu().numpy().view(np.uint64), t_want)
    assert bytes(sbuf[:sb].cpu().numpy()) == bytes(s_want)
    assert tape_parser.string_path() == 1  # the string buffer came from the stream compaction, the tape wrote its length words
    # the same with the per-string kernels forced, and a document they have to take over (a string the reference rejects far behind)
    os.environ["SJGPU_STRING_STREAM"] = "0"
    try:
        err2, tw2, sb2 = tape_parser.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), sbuf.numel(), 1024, st)
    finally:
        del os.environ["SJGPU_STRING_STREAM"]
    assert (err2, tw2, sb2) == (err, tw, sb) and tape_parser.string_path() == 2
    assert np.array_equal(tape[:tw].cpu().numpy().view(np.uint64), t_want) and bytes(sbuf[:sb].cpu().numpy()) == bytes(s_want)
    bad = np.concatenate([np.frombuffer(b"[", np.uint8), a, np.frombuffer(b',"x\\qy"]', np.uint8)])
    assert _assert_same_parse(tape_parser, want, bad) == checkers.STRING_ERROR and tape_parser.string_path() == 2


def test_stage2_from_the_token_stream(tape_parser, orc, ref):
    """sjgpu_stage2_tokens_device (round 5): stage 2 fed from the token stream stage 1 wrote beside the list -- the same tape and string buffer word for word,
    for valid documents and broken ones (the error code is the reference's)."""
    import torch
    want = _checker_parse(orc, ref)
    st = torch.cuda.current_stream().cuda_stream
    docs = [corpus.twitter_like(8 << 20, 98)[0], corpus.large_random(6 << 20, 97)[0], np.frombuffer(b'{"a":[1,2.5e3,true,null,"x\\u00e9"],"b":{}}', np.uint8),
            np.frombuffer(b'[1,2,,3]', np.uint8), np.frombuffer(b'{"a":tru}', np.uint8), np.frombuffer(b'[[[[1]]]', np.uint8)]
    for a in docs:
        L = len(a)
        buf = torch.from_numpy(a.copy()).cuda()
        idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        tok = torch.empty(L + 16, dtype=torch.uint8, device="cuda")
        tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
        sbuf = torch.empty(5 * (L // 3) + 256, dtype=torch.uint8, device="cuda")
        assert tape_parser.stage1_tokens_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, tok.data_ptr(), L + 16, st) == 0
        n, flags, _ = tape_parser.result(st)
        assert flags == 0
        err, tw, sb = tape_parser.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), sbuf.numel(), 1024, st, tok_ptr=tok.data_ptr())
        e_want, t_want, s_want = want(a)
        assert err == e_want, (bytes(a[:40]), err, e_want)
        if err == 0:
            assert (tw, sb) == (len(t_want), len(s_want))
            assert np.array_equal(tape[:tw].cpu().numpy().view(np.uint64), t_want) and bytes(sbuf[:sb].cpu().numpy()) == bytes(s_want)


def test_comm_world_size_one(orc):
    """sjgpu_comm_* (RCCL below the C-ABI): communicator of one rank -- ncclCommInitRank, the (n, base) all-gather, the root's own slot and
    the widening kernel; the N > 1 sends need N devices (the driver's 8-GPU node) and are covered on the CPU by the gloo twin"""
    import torch
    a, _ = corpus.amazon_ndjson(4 << 20, 31)
    L = len(a)
    p = capi.DomParserImplementation(L)
    buf = torch.from_numpy(a).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
    n, flags, _ = p.result(st)
    assert flags == 0
    comm = capi.Comm(0, 1, capi.comm_unique_id(), 0)
    out = torch.zeros(n + 8, dtype=torch.int64, device="cuda")
    base = 123456789012
    total, counts = comm.gather_indices(idx.data_ptr(), n, base, 0, out.data_ptr(), out.numel(), st)
    torch.cuda.synchronize()
    assert (total, counts) == (n, [n])
    want = orc.stage1(a, 0)[2][:n].astype(np.int64) + base
    assert np.array_equal(out[:n].cpu().numpy(), want)
    with pytest.raises(capi.SjgpuError):
        comm.gather_indices(idx.data_ptr(), n, base, 0, out.data_ptr(), 4, st)  # too small: refused AFTER the exchange has drained
    assert comm.ranks() == 1
    comm.close()
    p.close()


def test_comm_root_that_cannot_allocate_leaves_nobody_waiting(orc, monkeypatch):
    """ADVICE r3 / VERDICT r3 weak #1c: a root whose staging allocation fails used to return before the receives were posted while
    the senders had already decided to send.  Now the room the root has travels with the counts, a root that must grow allocates first
    and spreads its verdict, and on failure every rank returns with nothing in flight: NOMEM here (a world of one is root and only
    rank), and the same communicator works again afterwards."""
    import torch
    a, _ = corpus.amazon_ndjson(1 << 20, 33)
    L = len(a)
    p = capi.DomParserImplementation(L)
    buf = torch.from_numpy(a).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
    n, flags, _ = p.result(st)
    comm = capi.Comm(0, 1, capi.comm_unique_id(), 0)
    out = torch.zeros(n + 8, dtype=torch.int64, device="cuda")
    monkeypatch.setenv("SJGPU_DEBUG_COMM_FAIL_STAGING", "1")
    with pytest.raises(capi.SjgpuError, match="error -3"):
        comm.gather_indices(idx.data_ptr(), n, 0, 0, out.data_ptr(), out.numel(), st)
    monkeypatch.delenv("SJGPU_DEBUG_COMM_FAIL_STAGING")
    total, counts = comm.gather_indices(idx.data_ptr(), n, 7, 0, out.data_ptr(), out.numel(), st)
    torch.cuda.synchronize()
    assert (total, counts) == (n, [n])
    assert np.array_equal(out[:n].cpu().numpy(), orc.stage1(a, 0)[2][:n].astype(np.int64) + 7)
    comm.close()
    p.close()


def test_comm_two_ranks_on_two_devices(orc):
    """sjgpu_comm_gather_indices with a world of two, one rank per device, both in this process (two threads): the exact-count
    ncclSend / ncclRecv leg over the link between the devices.  Needs two GPUs: SKIPPED (and reported as such) on a 1-GPU box."""
    import threading
    import torch
    if capi.device_count() < 2:
        pytest.skip("one GPU on this box: the N > 1 send / receive leg needs two devices")
    shards = [corpus.amazon_ndjson(2 << 20, 41)[0], corpus.amazon_ndjson(3 << 20, 42)[0]]
    bases = [0, len(shards[0])]
    uid = capi.comm_unique_id()
    results, errors = [None, None], []

    def rank(r):
        try:
            torch.cuda.set_device(r)
            a = shards[r]
            L = len(a)
            p = capi.DomParserImplementation(L, device=r)
            buf = torch.from_numpy(a).to(f"cuda:{r}")
            idx = torch.empty(L + 16, dtype=torch.int32, device=f"cuda:{r}")
            st = torch.cuda.current_stream(r).cuda_stream
            assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
            n, flags, _ = p.result(st)
            comm = capi.Comm(r, 2, uid, r)
            out = torch.zeros((len(shards[0]) + len(shards[1])) // 4 if r == 0 else 8, dtype=torch.int64, device=f"cuda:{r}")
            for _ in range(2):  # the second call is the steady state: the staging array has its size, no second round
                total, counts = comm.gather_indices(idx.data_ptr(), n, bases[r], 0, out.data_ptr(), out.numel(), st)
            torch.cuda.synchronize(r)
            results[r] = (comm.ranks(), total, counts, out[:total].cpu().numpy() if r == 0 else None)
            comm.close()
            p.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((r, repr(e)))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors and all(r is not None for r in results), errors
    want = np.concatenate([orc.stage1(shards[r], 0)[2][: orc.stage1(shards[r], 0)[1]].astype(np.int64) + bases[r] for r in range(2)])
    ranks, total, counts, got = results[0]
    assert ranks == 2 and results[1][0] == 2 and total == len(want) and sum(counts) == total
    assert np.array_equal(got, want)


def test_raw_key_matching(orc):
    """On-Demand's raw key comparison for every key of the list in one pass (sjgpu_match_keys_device) against the oracle's loop
    (pinned against the reference's raw_json_string::unsafe_is_equal on the CPU tier)"""
    import torch
    docs = [np.fromfile(os.path.join(_paths.REPO_ROOT, "tests", "golden", "jsonexamples", "twitter.json"), dtype=np.uint8),
            corpus.twitter_like(8 << 20, 5)[0], corpus.large_random(4 << 20, 6)[0],
            np.frombuffer(b'{"id":1,"id ":2,"i":3,"idx":4, "id" :5,"text":"id","na\\"me":6,"":7,"id":"id"}', dtype=np.uint8).copy()]
    names = [b"id", b"text", b"screen_name", b"x", b"", b'na\\"me', b"retweet_count", b"not_there", b"i"]
    p = capi.DomParserImplementation(CAP)
    st = torch.cuda.current_stream().cuda_stream
    for a in docs:
        L = len(a)
        buf = torch.from_numpy(a).cuda()
        idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
        n, flags, _ = p.result(st)
        assert flags == 0
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        m = p.match_keys_device(buf.data_ptr(), L, idx.data_ptr(), n, names, out.data_ptr(), st)
        want, wm = orc.match_keys(a, idx[: n + 1].cpu().numpy().view(np.uint32), n, names)
        assert m == wm
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
    assert wm >= 6
    p.close()


def _walk_windows(stage1_fn, stream, batch):
    """drive stage1 the way document_stream::run_stage1 does (dom/document_stream-inl.h:285-317): consecutive windows, the next one starting
    where the last complete document of this one ended; returns every call's observable result"""
    out, start, guard = [], 0, 0
    total = len(stream)
    while start < total and guard < 100000:
        guard += 1
        remaining = total - start
        final = remaining <= batch
        w = stream[start: start + (remaining if final else batch)]
        mode = capi.STREAMING_FINAL if final else capi.STREAMING_PARTIAL
        err, n, idx = stage1_fn(w, mode)
        out.append((start, err, n, tuple(int(x) for x in idx[: n + 3])))
        if err not in (0, checkers.EMPTY) or final:
            break
        nxt = int(idx[n])  # next_batch_start: structural_indexes[n_structural_indexes]
        if nxt == 0:
            break
        start += nxt
    return out


def test_windows_of_a_registered_stream(orc):
    """parse_many's windows answered from a look-ahead span (sjgpu_stream_register) are word for word what the same calls deliver without it
    -- and what the oracle delivers: NDJSON, tight windows that cut strings and multi-byte characters, streams with errors in them"""
    rng = np.random.default_rng(88)
    nd, _ = corpus.amazon_ndjson(72 << 20, 9)  # crosses two spans of 32 MiB
    tw = []
    for _ in range(3000):
        tw.append(jsongen.random_document(rng, max_depth=3))
    small = np.frombuffer(b"\n".join(tw) + b"\n", dtype=np.uint8).copy()
    broken = small.copy()
    broken[len(broken) // 2] = 0x01  # a control character, most likely inside a string somewhere in the middle
    bad_utf8 = small.copy()
    bad_utf8[len(bad_utf8) // 3] = 0xFF
    cases = [(nd, 1_000_000), (nd[: 3 << 20], 20000), (small, 4096), (small, 700), (broken, 4096), (bad_utf8, 4096)]
    p_plain = capi.DomParserImplementation(8 << 20)
    p_span = capi.DomParserImplementation(8 << 20)
    for stream, batch in cases:
        stream = np.ascontiguousarray(stream)
        plain = _walk_windows(lambda w, m: g_stage1(p_plain, w, m), stream, batch)
        assert capi.stream_register(stream) == 0
        try:
            spanned = _walk_windows(lambda w, m: g_stage1(p_span, w, m), stream, batch)
        finally:
            assert capi.stream_unregister(stream) == 0
        assert len(plain) == len(spanned) and len(plain) >= 3
        for a, b in zip(plain, spanned):
            assert a == b, (batch, a[:3], b[:3])
        if len(stream) <= (4 << 20):
            want = _walk_windows(lambda w, m: orc.stage1(w, m), stream, batch)
            assert [x[:3] for x in want] == [x[:3] for x in plain]
            assert want == plain
    # a NEW stream at the address of an old one (allocators do that) must not be answered from the old one's spans
    again = np.ascontiguousarray(small)
    assert capi.stream_register(again) == 0
    first = _walk_windows(lambda w, m: g_stage1(p_span, w, m), again, 4096)
    assert capi.stream_unregister(again) == 0
    again[:] = np.frombuffer((b'{"k":[1,2,3]}\n' * (len(again) // 14 + 1))[: len(again)], dtype=np.uint8)
    assert capi.stream_register(again) == 0
    second = _walk_windows(lambda w, m: g_stage1(p_span, w, m), again, 4096)
    assert capi.stream_unregister(again) == 0
    assert second == _walk_windows(lambda w, m: g_stage1(p_plain, w, m), again, 4096) and second != first
    p_plain.close()
    p_span.close()
