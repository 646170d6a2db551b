#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference and oracle/_ref/libsjref.so):

    python tests/golden/make_golden.py

Outputs (all small, committed):
  utf8_vectors.json   the reference's own known-answer UTF-8 vectors, extracted from
                      /root/reference/tests/unicode_tests.cpp:191-229 (8 good, 29 bad), plus the
                      validate_tests shapes of tests/dom/basictests.cpp:1814-1909 restated as generators.
  small_cases.json    explicit small inputs x all 7 stage1 modes -> (err, n, idx[0..n+2]), minify, utf8
                      as produced by the reference's x86 kernel (SURVEY App. B + the streaming/RS/comma
                      shapes of tests/dom/document_stream_tests.cpp:592-739,1360-1440).
  corpora.json        digests (len, FNV-1a-64 of buffer / index words / minified bytes) for the
                      synthetic corpora of simdjson_amd.corpus and, when present, jsonexamples/*.
  random_digest.json  one FNV per seed over the reference's outputs on random adversarial inputs.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from simdjson_amd import _paths, build, corpus  # noqa: E402
import checkers  # noqa: E402


def c_string_literals_to_bytes(text):
    """Concatenated C string literals -> bytes (handles \\xH.. maximal munch, \\n etc.)."""
    out = bytearray()
    for lit in re.findall(r'"((?:[^"\\]|\\.)*)"', text):
        i = 0
        while i < len(lit):
            ch = lit[i]
            if ch != "\\":
                out.append(ord(ch)); i += 1; continue
            nxt = lit[i + 1]
            if nxt == "x":
                m = re.match(r"[0-9a-fA-F]+", lit[i + 2:])
                out.append(int(m.group(0), 16) & 0xFF); i += 2 + len(m.group(0))
            else:
                out.append({"n": 10, "t": 9, "r": 13, "0": 0, "\\": 92, '"': 34}[nxt]); i += 2
    return bytes(out)


def extract_array(src, name):
    m = re.search(name + r"\[\]\s*=\s*\{(.*?)\};", src, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    # split on commas that are outside string literals
    items, cur, in_str, esc = [], "", False, False
    for ch in body:
        if in_str:
            cur += ch
            if esc: esc = False
            elif ch == "\\": esc = True
            elif ch == '"': in_str = False
        elif ch == '"':
            in_str = True; cur += ch
        elif ch == ",":
            items.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        items.append(cur)
    return [c_string_literals_to_bytes(it) for it in items if '"' in it]


SMALL_CASES = [
    b'{"a":[1,2.5,true,null,"x\\"y"]} ', b'["a\\\\","b"]', b'["a\\\\\\"","b"]', b'["abc', b'["abc\xff', b'["abc\xff"]',
    b'["a\x01"]', b'["a\x01\xff"]', b"\x0c ab \x1a cd", b"   \n ", b'"a"true', b'"' + b"\\" * 65 + b'""',
    b'"' + b"\\" * 64 + b'"', b'[1,2,3]  {"1":1,"2":3,"4":4} [1,2  ',
    b'[1,2,3]  {"1":1,"2":3,"4":4} "intentionally unclosed string  ', b'[1] [2] "\xe2\x82', b"", b" ", b"1", b'"',
    b"\\", b"\\\\", b'\\"', b"[", b"]", b"{}", b"[][]", b"[] []", b"1 2 3", b'{"a":1}{"b":2}', b'{"a":1} {"b":2} {"c":',
    b'{"a":1},{"b":2},{"c":3}', b'{"a":[1,2]},{"b":2},', b",", b",,", b"[1,2],[3,4],[5", b"1,2,3", b'"a","b"',
    b"\x1e{\"a\":1}\n\x1e{\"b\":2}\n", b"\x1e1\n\x1e2\n\x1e3", b"\x1e\x1e1\n", b"\x1e \x1e \"s\"\n\x1e[1]", b"\x1e", b"\x1e\n",
    b"\x1e{\"a\":1}\n\x1e{\"b\":", b"\x1e\"abc", b"  \x1e  true \x1e false", b"\xef\xbb\xbf[1]", b"\xe2\x82\xac", b"\xe2\x82",
    b"\xe2", b"\xf0\x9f\x98", b'["\xf0\x9f\x98\x80"]', b'["\xed\xa0\x80"]', b'["\xc0\xaf"]', b"[1,\x00]", b'"\x00"', b'"\x1f"',
    b'"\x20"', b'["\\u0000"]', b"nul", b"truefalse", b'tru"e"', b'"a""b"', b'{"a" :1 , "b": [ ] }', b"[[[[[[[[", b"]]]]]]]]",
    b'["' + b"x" * 70 + b'"]', b'["' + b"x" * 62 + b'\\"' + b"y" * 70 + b'"]', b" " * 64 + b"[", b" " * 63 + b'"ab"',
    b'{"k":"' + b"\\\\" * 40 + b'"}', b'{"k":"' + b"\\\\" * 40 + b'\\"}', b"[" + b"1," * 100 + b"1]",
]


def main():
    build.build_corpus()
    build.build_oracle()
    ref = checkers.Reference()
    orc = checkers.Oracle()
    impl = ref.best_impl()
    print("reference kernel:", impl)

    # ---- utf8 vectors ---------------------------------------------------------------------------
    src = open(os.path.join(_paths.REFERENCE_DIR, "tests", "unicode_tests.cpp"), encoding="latin-1").read()
    good = extract_array(src, "goodsequences")
    bad = extract_array(src, "badsequences")
    assert len(good) == 8 and len(bad) == 29, (len(good), len(bad))
    for g in good:
        assert ref.validate_utf8(impl, g)
    for b in bad:
        assert not ref.validate_utf8(impl, b)
    json.dump({"source": "tests/unicode_tests.cpp:191-229", "good": [g.hex() for g in good], "bad": [b.hex() for b in bad]},
              open(os.path.join(HERE, "utf8_vectors.json"), "w"), indent=0)

    # ---- small cases x modes ----------------------------------------------------------------------
    cases = []
    for data in SMALL_CASES:
        entry = {"hex": data.hex(), "stage1": {}}
        for mname, mode in checkers.MODES.items():
            obs = checkers.observable(data, mode, *ref.stage1(impl, data, mode))
            entry["stage1"][mname] = {"err": obs[0]} if len(obs) == 1 else {"err": obs[0], "n": obs[1], "idx": list(obs[2])}
        merr, mout = ref.minify(impl, data)
        entry["minify"] = {"err": merr, "hex": bytes(mout).hex()}
        entry["utf8"] = ref.validate_utf8(impl, data)
        cases.append(entry)
    json.dump({"kernel": impl, "cases": cases}, open(os.path.join(HERE, "small_cases.json"), "w"), indent=0)

    # ---- corpora digests ---------------------------------------------------------------------------
    def digest(name, a):
        a = checkers.as_u8(a)
        err, n, idx = ref.stage1(impl, a, 0)
        merr, mout = ref.minify(impl, a)
        d = {"name": name, "len": int(len(a)), "buf_fnv": orc.fnv(a), "stage1_err": err, "n": n,
             "idx_fnv": orc.fnv(idx) if err not in checkers.EARLY else None, "minify_err": merr, "minify_len": int(len(mout)),
             "minify_fnv": orc.fnv(mout), "utf8": ref.validate_utf8(impl, a)}
        print(d)
        return d

    corp = []
    for kind, fn in (("large_random", corpus.large_random), ("amazon_ndjson", corpus.amazon_ndjson), ("twitter_like", corpus.twitter_like)):
        for target, seed in ((1000, 1), (70000, 2), (1 << 20, 3), (16 << 20, 4), (100 << 20, 5)):
            a, units = fn(target, seed)
            d = digest(f"{kind}:{target}:{seed}", a)
            d["units"] = units
            corp.append(d)
    for k in (1, 31, 32, 33, 2048, 8192, 1 << 20):
        corp.append(digest(f"deep_nesting:{k}", corpus.deep_nesting(k)))
    runs = [1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 16383, 16384, 16385, (1 << 20) - 1, 1 << 20, (1 << 20) + 1]
    for pad in (0, 1, 37):
        corp.append(digest(f"backslash_runs:{pad}", corpus.backslash_runs(runs, pad)))
    ex = os.path.join(_paths.REFERENCE_DIR, "jsonexamples")
    for fn in ("twitter.json", "citm_catalog.json", "amazon_cellphones.ndjson"):
        p = os.path.join(ex, fn)
        if os.path.exists(p):
            corp.append(digest("jsonexamples/" + fn, np.fromfile(p, dtype=np.uint8)))
    json.dump({"kernel": impl, "corpora": corp}, open(os.path.join(HERE, "corpora.json"), "w"), indent=0)

    # ---- random adversarial digests ----------------------------------------------------------------
    rnd = []
    for seed in range(16):
        rng = np.random.default_rng(1000 + seed)
        h = []
        for it in range(400):
            n = int(rng.integers(0, 600))
            a = corpus.random_adversarial(n, int(rng.integers(0, 1 << 31)), ascii_only=bool(it % 3 == 0),
                                          p_backslash=0.15 if it % 4 == 0 else 0.0)
            for mode in range(7):
                obs = checkers.observable(a, mode, *ref.stage1(impl, a, mode))
                h.append(np.array([obs[0]], np.uint32))
                if len(obs) > 1:
                    h.append(np.array([obs[1]], np.uint32)); h.append(np.array(obs[2], np.uint32))
            merr, mout = ref.minify(impl, a)
            h.append(np.array([merr, len(mout)], np.uint32)); h.append(mout.astype(np.uint32))
            h.append(np.array([int(ref.validate_utf8(impl, a))], np.uint32))
        rnd.append({"seed": 1000 + seed, "fnv": orc.fnv(np.concatenate(h))})
    json.dump({"kernel": impl, "digests": rnd}, open(os.path.join(HERE, "random_digest.json"), "w"), indent=0)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
