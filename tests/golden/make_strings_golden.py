#!/usr/bin/env python3
"""Generates tests/golden/strings.json from the REAL reference (stage 2's string work).

Run in the build container only (needs oracle/_ref/libsjref.so):   python tests/golden/make_strings_golden.py

  vectors   : string bodies (hex; the bytes behind the opening quote, closing quote included) x allow_replacement ->
              the unescaped bytes (hex) or null, as dom_parser_implementation::parse_string of the reference's x86 kernel
              returns them (src/generic/stage2/stringparsing.h:150-193; haswell.cpp:151-153) -- the hand-written cases of
              tests/test_oracle_vs_reference.py plus 2 000 seeded random bodies.
  buffers   : per fixture of tests/golden/jsonexamples: number of strings, length and FNV-1a-64 of document::string_buf as the
              reference's dom parse leaves it (src/generic/stage2/tape_builder.h:415-433).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import checkers  # noqa: E402
from test_oracle_vs_reference import STRING_BODIES  # noqa: E402


def main():
    ref, orc = checkers.Reference(), checkers.Oracle()
    impl = ref.best_impl()
    bodies = list(STRING_BODIES)
    for pre in list(range(28, 36)) + list(range(60, 68)) + [127, 128]:
        for esc in (b'\\n', b'\\u0041', b'\\ud83d\\ude00', b'\\"', b'\\\\', b'\\q', b'\\ud800', b'\\udc00x'):
            bodies.append(b'a' * pre + esc + b'tail"')
    rng = np.random.default_rng(20260921)
    alphabet = [b'a', b'Z', b' ', b'\\', b'\\', b'u', b'u', b'd', b'D', b'8', b'c', b'0', b'f', b'F', b'9', b'n', b'"', b'/', b'x', "é".encode(), "日".encode()]
    for _ in range(2000):
        k = int(rng.integers(0, 80))
        bodies.append(b''.join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), k)) + b' "')
    vectors = []
    for body in bodies:
        for allow in (0, 1):
            r = ref.parse_string(impl, body, bool(allow))
            vectors.append([body.hex(), allow, None if r is None else r.hex()])
    buffers = {}
    d = os.path.join(HERE, "jsonexamples")
    for name in sorted(os.listdir(d)):
        if not name.endswith(".json"):
            continue
        data = np.frombuffer(open(os.path.join(d, name), "rb").read(), dtype=np.uint8)
        err, buf, strings = ref.dom_string_buf(impl, data)
        assert err == 0
        buffers[name] = {"strings": strings, "bytes": int(len(buf)), "fnv1a64": orc.fnv(buf)}
    out = {"reference_kernel": impl, "vectors": vectors, "buffers": buffers}
    with open(os.path.join(HERE, "strings.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(len(vectors), "vectors,", buffers)


if __name__ == "__main__":
    main()
