"""Where the plug-in path's fixed costs are: context creation, capacity changes, tiny-document calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simdjson_amd import capi, corpus

def t(fn, reps):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6

def create_destroy(cap):
    p = capi.DomParserImplementation(cap); p.close()
print("ctx create+destroy 1 MB  : %.1f us" % t(lambda: create_destroy(1 << 20), 50))
print("ctx create+destroy 64 MB : %.1f us" % t(lambda: create_destroy(64 << 20), 20))
p = capi.DomParserImplementation(1 << 20)
tiny = b'{"a":[1,2,3],"b":"c"}'
doc64k, _ = corpus.twitter_like(65536, 1)
doc1m, _ = corpus.twitter_like(1 << 20, 1)
print("stage1 host path 21 B    : %.1f us" % t(lambda: p.stage1(tiny), 500))
print("stage1 host path 64 KB   : %.1f us" % t(lambda: p.stage1(doc64k), 300))
print("stage1 host path 1 MB    : %.1f us" % t(lambda: p.stage1(doc1m[: (1 << 20) - 8]), 200))
print("minify host path 21 B    : %.1f us" % t(lambda: p.minify(tiny), 500))
print("utf8   host path 21 B    : %.1f us" % t(lambda: p.validate_utf8(tiny), 500))
print("set_capacity 1MB<->2MB   : %.1f us" % t(lambda: (p.set_capacity(2 << 20), p.set_capacity(1 << 20)), 30))
