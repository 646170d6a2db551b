"""lab: does the PLACEMENT of the split pipeline's mask plane decide its speed?  (four copies of one library differed by 7 % on escape_heavy in
scripts/lib_ab.py, session R.)  One library built with an environment switch that shifts the masks inside their allocation (SJGPU_LAB_MASK_PAD, bytes); one
context; the pads taking turns, round after round.  Also prints the device addresses involved.  SJGPU_LIB must name the lab build."""
import os, sys, statistics, json
sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pads = [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, (4 << 20) + 65536, 8 << 20]
for kind in ("escape_heavy", "amazon_ndjson"):
    a = getattr(corpus, kind)(1 << 30, 1000)[0]
    L = len(a)
    buf = torch.from_numpy(a).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    for ctx_no in range(3):  # three contexts = three allocations of the workspace
        p = capi.DomParserImplementation(L)
        p.set_pipeline("split")
        for _ in range(3):
            p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        t = {pad: [] for pad in pads}
        for rnd in range(8):
            for pad in (pads if rnd % 2 == 0 else pads[::-1]):
                os.environ["SJGPU_LAB_MASK_PAD"] = str(pad)
                p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
                e0.record()
                for _ in range(8):
                    p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
                e1.record(); e1.synchronize()
                t[pad].append(1e3 * e0.elapsed_time(e1) / 8)
        print(json.dumps({"kind": kind, "context": ctx_no, "buf": hex(buf.data_ptr()), "idx": hex(idx.data_ptr()), "median_us_by_pad": {str(k): round(statistics.median(v), 1) for k, v in t.items()}}), flush=True)
        os.environ["SJGPU_LAB_MASK_PAD"] = "0"
        p.close()
    del buf, idx
