"""Phase-latency budget of the single-pass stage-1 kernel from in-kernel wall-clock stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simdjson_amd import capi, corpus
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
a, _ = corpus.large_random(size, 1000)
L = len(a)
p = capi.DomParserImplementation(L)
buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ntiles = (L + 65535) // 65536
for rep in range(2):
    t = p.debug_trace_stage1(buf.data_ptr(), L, idx.data_ptr(), L + 3, ntiles).astype(np.int64)
t0 = t[:, 0].min()
names = ["ticket", "scan(w0)", "wait others", "lookback", "bcast", "emit(w0)"]
d = np.diff(t[:, :7], axis=1) / 100.0  # us
print("tiles", ntiles, "kernel span us", (t[:, 6].max() - t0) / 100.0)
for k, nm in enumerate(names):
    print(f"{nm:12s} mean {d[:, k].mean():8.2f} us  p50 {np.median(d[:, k]):8.2f}  p95 {np.percentile(d[:, k], 95):8.2f}  max {d[:, k].max():8.2f}")
print("tile total   mean", (t[:, 6] - t[:, 0]).mean() / 100.0)
start = (t[:, 0] - t0) / 100.0
order = np.argsort(start)
print("first tiles start(us):", np.round(start[:8], 2), " last tiles:", np.round(start[-4:], 2))
# how far ahead of tile j's scan-end is the aggregate of j-1.. (look-back rounds proxy)
print("lookback by decile of tile index:", [round(float(x), 2) for x in [d[int(q * (ntiles - 1)), 3] for q in np.linspace(0, 1, 11)]])
