"""Lab: where a wave of k_tok_stage spends its time (a build with -DSJGPU_LAB_STAGE_PHASES accumulates wall_clock64 ticks per phase over all waves):
python scripts/lab/stage_phases.py build/ab/libsjgpu_phases.so [kind ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from lib_ab import capi_for
import torch
from simdjson_amd import corpus
mod = capi_for(sys.argv[1], "phases")
kinds = sys.argv[2:] or ["twitter_like", "large_random"]
st = torch.cuda.current_stream().cuda_stream
for kind in kinds:
    host, _ = getattr(corpus, kind)(256 << 20, 3000)
    L = len(host)
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
    scap = 5 * (L // 3) + 256
    sbuf = torch.empty(scap, dtype=torch.uint8, device="cuda")
    p = mod.DomParserImplementation(L)
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
    n, _, _ = p.result(st)
    out = (ctypes.c_ulonglong * 8)()
    for rep in range(3):
        p.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), scap, 1024, st)
        torch.cuda.synchronize()
        assert p.L.sjgpu_lab_stage_phases(out) == 0
    waves, groups = out[7], out[6]
    names = ["list rows parked", "staging", "rows", "numbers", "end barrier"]
    print(kind, "waves", waves, "groups", groups, "(%.2f per wave)" % (groups / max(waves, 1)))
    for k, name in enumerate(names):
        print("  %-18s %8.2f us per wave" % (name, out[k] / 100.0 / max(waves, 1)))
    print("  %-18s %8.2f us per wave" % ("sum", sum(out[:5]) / 100.0 / max(waves, 1)))
    p.close(); del buf, idx, tape, sbuf
