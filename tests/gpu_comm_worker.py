"""GPU tier, run as a SUBPROCESS by tests/test_gpu_comm.py with SJGPU_RCCL_LIB = the loop-back library (build/tests/librccl_loopback_hip.so):
sjgpu_comm_gather_indices of the REAL libsjgpu.so with a world of N ranks on a box that has one GPU.  The ranks are threads of this
process, each with its own HIP stream, parser, NDJSON shard and communicator handle, all on device 0; libsjgpu's dlopen finds the
loop-back library instead of librccl (a process of its own, because the library is opened once per process), so the product's own
all-gather / growth round / grouped exact-count ncclSend + ncclRecv / k_widen_all run with N > 1 -- at BASELINE configs[3]'s full size
when asked (8 shards of 1 GiB: bases up to 7 GiB through k_widen_all).  Every shard is scanned by the reference on the host
(oracle/_ref/libsjref.so, or the oracle where that is absent) and the gathered 64-bit positions must equal base + offset, element for
element (/root/reference/include/simdjson/dom/document_stream-inl.h:250).

    python tests/gpu_comm_worker.py <world> <shard_bytes> [rounds]      prints one JSON line
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world, shard_bytes = int(sys.argv[1]), int(sys.argv[2])
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    import torch
    import checkers  # test infrastructure: the CPU side of the comparison
    from simdjson_amd import capi, corpus
    assert os.environ.get("SJGPU_RCCL_LIB"), "run me with SJGPU_RCCL_LIB = the loop-back library"
    ref = checkers.Reference() if checkers.have_reference_lib() else None
    impl = ref.best_impl() if ref else None
    orc = checkers.Oracle()
    uid = capi.comm_unique_id()
    # shards of different sizes (ends at a newline: zero carry-in); the bases are the running byte offsets
    sizes = [shard_bytes + (r * 4096 if shard_bytes < (1 << 28) else 0) for r in range(world)]
    shards = [None] * world
    results = [None] * world
    errors = []
    gate = threading.Barrier(world)
    t0 = time.time()

    def rank(r):
        try:
            a, _ = corpus.amazon_ndjson(sizes[r], 100 + r)
            shards[r] = a
            gate.wait(600)
            base = sum(len(shards[k]) for k in range(r))
            L = len(a)
            s = torch.cuda.Stream(device=0)
            with torch.cuda.stream(s):
                st = s.cuda_stream
                p = capi.DomParserImplementation(L, device=0)
                buf = torch.from_numpy(a).to("cuda:0", non_blocking=False)
                cap = L // 8 + 1024
                idx = torch.empty(cap + 16, dtype=torch.int32, device="cuda:0")
                assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), cap, st) == 0
                n, flags, _ = p.result(st)
                assert flags == 0, flags
                comm = capi.Comm(r, world, uid, 0)
                outs = {}
                totals = []
                roots = [0] * rounds + ([world - 1] if world > 1 else [])
                for root in roots:
                    out = None
                    if r == root and root not in outs:
                        outs[root] = torch.zeros(sum(len(x) for x in shards) // 8 + 1024 * world, dtype=torch.int64, device="cuda:0")
                    out = outs.get(root) if r == root else None
                    total, counts = comm.gather_indices(idx.data_ptr(), n, base, root, out.data_ptr() if out is not None else 0, out.numel() if out is not None else 0, st)
                    totals.append((root, total, counts))
                s.synchronize()
                results[r] = {"ranks": comm.ranks(), "n": n, "base": base, "totals": totals, "outs": outs, "idx": idx}
                comm.close()
                p.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((r, repr(e)))
            try:
                gate.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(1500) for t in ts]
    if errors or any(x is None for x in results):
        print(json.dumps({"ok": False, "errors": errors}))
        return 1
    t_gather = time.time() - t0
    # the reference's batches, shard by shard: global position = batch_start + structural_indexes[i]
    ok, why = True, ""
    counts_want = []
    refs = []
    for r in range(world):
        if ref and impl:
            err, n, ridx = ref.stage1(impl, shards[r], 0)
        else:
            err, n, ridx = orc.stage1(shards[r], 0)
        assert err == 0
        counts_want.append(n)
        refs.append(torch.from_numpy(ridx[:n].astype(np.int64)).to("cuda:0") + results[r]["base"])
    want = torch.cat(refs)
    for r in range(world):
        res = results[r]
        if res["ranks"] != world or res["n"] != counts_want[r]:
            ok, why = False, f"rank {r}: ranks {res['ranks']}, n {res['n']} vs {counts_want[r]}"
        for root, total, counts in res["totals"]:
            if total != len(want) or list(counts) != counts_want:
                ok, why = False, f"rank {r}, root {root}: total {total} vs {len(want)}, counts {counts} vs {counts_want}"
        for root, out in res["outs"].items():
            if not bool(torch.equal(out[: len(want)], want)):
                ok, why = False, f"root {root}: the gathered positions differ from base + the reference's offsets"
    largest = int(want[-1].item()) if len(want) else 0
    print(json.dumps({"ok": ok, "why": why, "world": world, "n_ranks_seen_by_rccl": results[0]["ranks"], "shard_bytes": sizes, "total_bytes": int(sum(sizes)),
                      "total_structurals": int(len(want)), "largest_global_position": largest, "beyond_32_bits": largest >= (1 << 32),
                      "reference": impl or "oracle", "seconds": round(t_gather, 2),
                      "library": os.path.basename(os.environ["SJGPU_RCCL_LIB"])}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
