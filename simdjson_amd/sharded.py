"""NDJSON / parse_many across the GPUs of one node (BASELINE.json config 4).

Documents of an NDJSON stream are independent units and a raw '\\n' cannot occur inside a valid JSON
string (/root/reference/include/simdjson/base.h:66-74), so a buffer cut at newlines shards with ZERO
carry-in: every rank runs the ordinary single-GPU stage-1 scan over its slice and produces
slice-relative uint32 offsets; global position = slice_base + offset (64-bit), the same convention as
document_stream's batch_start + structural_indexes[i]
(/root/reference/include/simdjson/dom/document_stream-inl.h:250).  There is NO data-path collective in
the scan itself.  The only exchange is optional: concatenating the per-rank index arrays (one
all_gather of the counts, one padded all_gather of the offsets -- RCCL over xGMI on GPUs, gloo in the
CPU tests).

`scan_fn` is the per-shard scan.  The product default is the HIP path (GpuShardScanner); the CPU test
tier injects the oracle instead so that sharding, bases and the gather are covered without a GPU.
"""
from dataclasses import dataclass

import numpy as np


def newline_cuts(buf: np.ndarray, parts: int):
    """parts+1 ascending cut offsets: cut k is just after the first '\\n' at or after k*len/parts."""
    n = len(buf)
    cuts = [0]
    for k in range(1, parts):
        target = max((k * n) // parts, cuts[-1])
        nl = np.flatnonzero(buf[target:min(n, target + (1 << 24))] == 0x0A)
        if len(nl) == 0:
            nl = np.flatnonzero(buf[target:] == 0x0A)
        cuts.append(n if len(nl) == 0 else target + int(nl[0]) + 1)
    cuts.append(n)
    return cuts


@dataclass
class ShardScan:
    base: int          # byte offset of the shard in the whole stream
    length: int
    n: int             # structural count of the shard
    flags: int         # SJGPU_F_* of the shard
    idx: object        # shard-relative uint32 offsets: torch tensor (device) or numpy array, >= n entries


class GpuShardScanner:
    """Per-rank HIP scan: shard bytes -> device-resident index tensor (needs a GPU; no CPU fallback)."""

    def __init__(self, capacity, device):
        import torch
        from . import capi
        self.torch = torch
        self.device = device
        self.parser = capi.DomParserImplementation(capacity, device=device)

    def __call__(self, shard: np.ndarray):
        torch = self.torch
        L = len(shard)
        buf = torch.from_numpy(np.ascontiguousarray(shard)).to(f"cuda:{self.device}")
        idx = torch.empty(L + 3, dtype=torch.int32, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.parser.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream)
        if rc != 0:
            raise RuntimeError(f"stage1_device returned {rc}")
        n, flags, _ = self.parser.result(stream)
        return idx, n, flags


def scan_shard(buf: np.ndarray, rank: int, world: int, scan_fn) -> ShardScan:
    """Rank `rank`'s share of the stream: cut at newlines, scan with zero carry-in."""
    cuts = newline_cuts(buf, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    if hi == lo:
        return ShardScan(lo, 0, 0, 0, np.zeros(0, np.uint32))
    idx, n, flags = scan_fn(buf[lo:hi])
    return ShardScan(lo, hi - lo, n, flags, idx)


def gather_global_indices(local: ShardScan, group=None):
    """Concatenate all ranks' structural positions as global int64 offsets (every rank gets the result).

    Collective: all_gather(counts, bases, flags) then ONE padded all_gather of the uint32 offsets.
    Returns (positions int64 tensor [sum n], per-rank counts list, OR of flags)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    idx = local.idx
    if isinstance(idx, np.ndarray):
        idx = torch.from_numpy(idx.astype(np.int64).astype(np.int32) if idx.dtype != np.int32 else idx)
    dev = idx.device
    meta = torch.tensor([local.n, local.base, local.flags], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    bases = [int(m[1]) for m in metas]
    flags = 0
    for m in metas:
        flags |= int(m[2])
    width = max(max(counts), 1)
    mine = torch.zeros(width, dtype=torch.int32, device=dev)
    mine[: local.n] = idx[: local.n]
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = torch.cat([(parts[r][: counts[r]].to(torch.int64) & 0xFFFFFFFF) + bases[r] for r in range(world)])
    return out, counts, flags
