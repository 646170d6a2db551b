"""One workload through sjgpu_stage2_device a few times (for rocprofv3 --kernel-trace): python scripts/tape_once.py large_random|twitter_like [bytes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simdjson_amd import capi, corpus
kind = sys.argv[1] if len(sys.argv) > 1 else "large_random"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256 << 20
host, _ = getattr(corpus, kind)(size, 3000)
L = len(host)
p = capi.DomParserImplementation(L)
st = torch.cuda.current_stream().cuda_stream
buf = torch.from_numpy(host).cuda()
idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
n, flags, _ = p.result(st)
tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
scap = 5 * (L // 3) + 256
sbuf = torch.empty(scap, dtype=torch.uint8, device="cuda")
for _ in range(5):
    err, tw, sb = p.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), scap, 1024, st)
print(kind, L, n, err, tw, sb)
