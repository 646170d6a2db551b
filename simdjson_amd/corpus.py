"""Deterministic synthetic JSON corpora for the parity tests and bench.py (host tooling).

The three bulk generators live in csrc/corpus.c (shapes restated from the reference's benchmark
generators, /root/reference/benchmark/large_random/large_random.h:43-60 and
benchmark/large_amazon_cellphones/large_amazon_cellphones.h:66-81); the adversarial cases of
SURVEY.md section 8(d) item 4 are built here with numpy.  Every buffer is a pure function of its
arguments, so golden digests committed under tests/golden/ stay valid on any box.
"""
import ctypes

import numpy as np

from . import _paths

_lib = None


def _corpus_lib():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_paths.LIB_CORPUS)
        for name in ("sjc_large_random", "sjc_amazon_ndjson", "sjc_twitter_like"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64,
                           ctypes.POINTER(ctypes.c_uint64)]
        _lib = lib
    return _lib


def _bulk(name, target, seed):
    cap = int(target) + 8192
    buf = np.empty(cap, dtype=np.uint8)
    units = ctypes.c_uint64(0)
    n = getattr(_corpus_lib(), name)(buf.ctypes.data, cap, int(target), int(seed), ctypes.byref(units))
    if n == 0:
        raise RuntimeError(f"{name}: buffer too small for target {target}")
    return buf[:n], int(units.value)


def large_random(target_bytes, seed=1):
    """large_random-style array of {x,y,z} records, >= target_bytes. Returns (uint8 array, n_records)."""
    return _bulk("sjc_large_random", target_bytes, seed)


def amazon_ndjson(target_bytes, seed=1):
    """amazon_cellphones-style NDJSON (one array per line), >= target_bytes. Returns (array, n_lines)."""
    return _bulk("sjc_amazon_ndjson", target_bytes, seed)


def twitter_like(target_bytes, seed=1):
    """Pretty-printed nested statuses with escapes and multi-byte UTF-8. Returns (array, n_statuses)."""
    return _bulk("sjc_twitter_like", target_bytes, seed)


# ---- adversarial corpus -------------------------------------------------------

def deep_nesting(k):
    """'[' * k + ']' * k : every byte is a structural (density 1.0)."""
    return np.concatenate([np.full(k, ord("["), np.uint8), np.full(k, ord("]"), np.uint8)])


def backslash_runs(run_lengths, lead_pad=0):
    """Array of strings, each a backslash run of the given length followed by a quote char, so that
    odd runs escape the quote (string stays open until the next one) and even runs close it."""
    parts = [b" " * lead_pad, b"["]
    for i, r in enumerate(run_lengths):
        if i:
            parts.append(b",")
        parts.append(b'"' + b"\\" * r + b'"' + (b'x"' if r % 2 else b""))
    parts.append(b"]")
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def deep_nesting_doc(target_bytes, seed=1):
    """bench workload (BASELINE configs[4]): target_bytes of '[' * k + ']' * k.  Returns (array, k)."""
    k = (int(target_bytes) + 1) // 2
    return deep_nesting(k), k


ESCAPE_RUNS = (1, 2, 3, 4, 7, 8, 62, 63, 64, 65, 66, 126, 127, 128, 129, 130, 1, 1, 2, 2, 4094, 4095, 4096, 4097, 4098,
               16382, 16383, 16384, 16385, 16386, 65535, 65536, 65537, 3, 5, 9, 17, 33, 1, 2, 1, 2)


def escape_heavy(target_bytes, seed=1):
    """bench workload (BASELINE configs[4]): an array of strings that are almost nothing but backslash runs -- odd and
    even lengths from 1 to 64 KiB + 1, so escape carries cross blocks, chunks, segments and tiles -- >= target_bytes.
    Returns (array, n_strings)."""
    rot = int(seed) % len(ESCAPE_RUNS)
    runs = ESCAPE_RUNS[rot:] + ESCAPE_RUNS[:rot]
    block = backslash_runs(runs)[1:-1]  # without the enclosing brackets
    reps = max(1, -(-int(target_bytes) // (len(block) + 1)))
    body = np.tile(np.concatenate([block, np.frombuffer(b",", np.uint8)]), reps)
    body[-1] = ord("]")
    return np.concatenate([np.frombuffer(b"[", np.uint8), body]), reps * len(runs)


def boundary_straddle(payload: bytes, boundary: int, offset_before: int, total: int, filler=b" "):
    """Places `payload` so that it starts `offset_before` bytes before a multiple of `boundary`."""
    start = boundary - offset_before
    body = filler * start + payload
    if len(body) < total:
        body += filler * (total - len(body))
    return np.frombuffer(body, dtype=np.uint8).copy()


ADVERSARIAL_ALPHABET = bytes([0x22, 0x5C, 0x5C, 0x7B, 0x7D, 0x5B, 0x5D, 0x3A, 0x2C, 0x20, 0x0A, 0x09, 0x0D,
                              0x61, 0x31, 0x74, 0x01, 0x0C, 0x1A, 0x1E, 0x1F, 0x7F, 0x80, 0xBF, 0xC0, 0xC2,
                              0xE0, 0xED, 0xA0, 0x9F, 0xF0, 0xF4, 0x90, 0x8F, 0xF5, 0xFF, 0xE2, 0x82, 0xAC,
                              0x00, 0x20, 0x22, 0x5C])


def random_adversarial(n, seed, ascii_only=False, p_backslash=0.0):
    """n random bytes from an alphabet of quotes, backslashes, operators, whitespace, control bytes
    (incl. 0x0C/0x1A/0x1E) and valid/broken UTF-8 fragments (the survey's differential alphabet)."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(ADVERSARIAL_ALPHABET, dtype=np.uint8)
    if ascii_only:
        alpha = alpha[alpha < 0x80]
    out = alpha[rng.integers(0, len(alpha), size=n)]
    if p_backslash > 0:
        out = np.where(rng.random(n) < p_backslash, np.uint8(0x5C), out)
    return np.ascontiguousarray(out, dtype=np.uint8)


def fnv1a64(data) -> int:
    """FNV-1a-64 of a bytes-like / uint8 array (pure numpy-free loop is too slow; use the oracle's in tests)."""
    h = 0xCBF29CE484222325
    for b in bytes(data):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h
