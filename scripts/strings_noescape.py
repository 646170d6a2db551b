"""The string pass on a twitter-like document whose escapes have been replaced by plain characters (same size, same strings):
how much of k_strs_count / k_strs_write is escape handling?  python scripts/strings_noescape.py  (under rocprofv3 --kernel-trace)"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simdjson_amd import capi, corpus
host, _ = corpus.twitter_like(256 << 20, 3000)
plain = np.frombuffer(re.sub(rb"\\(.)", b"xx", host.tobytes(), flags=re.S), dtype=np.uint8).copy()
assert len(plain) == len(host)
for name, doc in (("with escapes", host), ("escapes replaced", plain)):
    L = len(doc)
    p = capi.DomParserImplementation(L)
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.from_numpy(doc).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
    n, flags, _ = p.result(st)
    assert flags == 0, flags
    cap = 5 * (L + 1) // 3 + 64
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    for _ in range(4):
        err, used, strings, bad = p.parse_strings_device(buf.data_ptr(), L, idx.data_ptr(), n, out.data_ptr(), cap, off.data_ptr(), False, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        p.parse_strings_device(buf.data_ptr(), L, idx.data_ptr(), n, out.data_ptr(), cap, off.data_ptr(), False, st)
    e1.record()
    torch.cuda.synchronize()
    print(name, L, n, err, used, strings, "path", p.string_path(), "ms per call", round(e0.elapsed_time(e1) / 6, 3))
    p.close()
