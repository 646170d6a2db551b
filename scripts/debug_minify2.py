import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import checkers
from simdjson_amd import capi, corpus
o = checkers.Oracle()
p = capi.DomParserImplementation(80 << 20)
for kind in ("amazon_ndjson", "twitter_like"):
    for size in (1 << 20, 16 << 20, 64 << 20):
        a, _ = getattr(corpus, kind)(size, 11)
        e, m1 = o.minify(a)
        ge, g1 = p.minify(a)
        ge2, g2 = p.minify(m1)
        print(kind, size, "pass1 equal", np.array_equal(g1, m1), "pass2 len", len(g2), "want", len(m1), flush=True)
        if len(g2) != len(m1) or not np.array_equal(g2, m1):
            mm = min(len(g2), len(m1))
            d = np.nonzero(g2[:mm] != m1[:mm])[0]
            if len(d):
                i = int(d[0])
                print("first diff at", i, "seg", i // 16384, "in-seg", i % 16384, "blk", (i % 4096) // 64, "byte", i % 64)
                print("want:", bytes(m1[max(0, i - 80): i + 40]))
                print("got :", bytes(g2[max(0, i - 80): i + 40]))
                # stage1 view of the same input for cross-check
                err = p.stage1(m1); n = p.n_structural_indexes
                oe, on, oidx = o.stage1(m1)
                print("stage1 on minified:", err, n, oe, on, np.array_equal(p.structural_indexes[:n+3], oidx))
            break
