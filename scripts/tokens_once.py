"""The token stream through a kernel trace (rocprofv3 --kernel-trace): stage 1 of one document on the split pipeline without and with the token-byte stream
beside the offsets, the depth scan that gathers and the depth scan fed from the stream, stage 2 both ways -- five calls each.
    python scripts/tokens_once.py amazon_ndjson|twitter_like|large_random [bytes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simdjson_amd import capi, corpus
kind = sys.argv[1] if len(sys.argv) > 1 else "amazon_ndjson"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
host, _ = getattr(corpus, kind)(size, 3000)
L = len(host)
p = capi.DomParserImplementation(L)
p.set_pipeline("split")
st = torch.cuda.current_stream().cuda_stream
buf = torch.from_numpy(host).cuda()
cap = L // 2
idx = torch.empty(cap + 16, dtype=torch.int32, device="cuda")
tok = torch.empty(cap + 16, dtype=torch.uint8, device="cuda")
for _ in range(5):
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), cap, st) == 0
n, flags, _ = p.result(st)
for _ in range(5):
    assert p.stage1_tokens_device(buf.data_ptr(), L, idx.data_ptr(), cap, tok.data_ptr(), cap + 16, st) == 0
n2, flags2, _ = p.result(st)
assert (n, flags) == (n2, flags2) and flags == 0
depth = torch.empty(n + 1, dtype=torch.int32, device="cuda")
depth2 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for _ in range(5):
    p.depth_scan_device(buf.data_ptr(), idx.data_ptr(), n, depth.data_ptr(), st)
for _ in range(5):
    p.depth_scan_tokens_device(tok.data_ptr(), n, depth2.data_ptr(), st)
torch.cuda.synchronize()
assert bool(torch.equal(depth, depth2))
if L <= (512 << 20):
    tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
    scap = 5 * (L // 3) + 256
    sbuf = torch.empty(scap, dtype=torch.uint8, device="cuda")
    for t in (0, tok.data_ptr()):
        for _ in range(5):
            err, tw, sb = p.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), scap, 1024, st, tok_ptr=t)
        print("stage 2", "from the token stream" if t else "gathering", err, tw, sb)
print(kind, L, n)
