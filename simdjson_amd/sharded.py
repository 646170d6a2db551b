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

ONE LARGE DOCUMENT ("general inputs") shards too, with one tiny exchange: cut where every
carry except the in-string bit is provably zero (`clean_cuts`, sjgpu_clean_cut), every rank computes the
quote parity of its shard (a read-only pass), ONE all_gather of a bit per rank gives every rank its
in-string carry-in, then every rank scans alone (`scan_document_shard`).

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


def clean_cuts(buf: np.ndarray, parts: int):
    """parts+1 ascending cut offsets of ONE document: cut k is the first position at or after k*len/parts whose
    previous byte is ASCII whitespace or one of , : [ ] { } (sjgpu_clean_cut) -- no escape, no scalar, no UTF-8
    sequence crosses it.  A region without such a byte (one huge string without blanks) yields empty shards."""
    from . import capi
    a = np.ascontiguousarray(buf, dtype=np.uint8)
    n = len(a)
    cuts = [0]
    for k in range(1, parts):
        cuts.append(capi.clean_cut(a, max((k * n) // parts, cuts[-1])))
    cuts.append(n)
    return cuts


F_UNCLOSED_STRING = 1  # SJGPU_F_UNCLOSED_STRING: for a shard, "ends inside a string"


def document_flags(flags_per_rank):
    """Flags of the whole document from the shards' flags: errors OR together, but only the LAST shard ending
    inside a string is an unclosed string."""
    out = 0
    for f in flags_per_rank:
        out |= f & ~F_UNCLOSED_STRING
    return out | (flags_per_rank[-1] & F_UNCLOSED_STRING)


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

    # shards of one document: quote parity (read-only pre-pass), then the scan with the exchanged carry-in
    def _upload(self, shard):
        return self.torch.from_numpy(np.ascontiguousarray(shard)).to(f"cuda:{self.device}")

    def parity(self, shard: np.ndarray):
        self._resident = self._upload(shard)  # stays in HBM for scan() of the same shard
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        return self.parser.string_parity_device(self._resident.data_ptr(), len(shard), stream)

    def scan(self, shard: np.ndarray, in_string: int):
        torch = self.torch
        L = len(shard)
        buf = getattr(self, "_resident", None)
        if buf is None or buf.numel() != L:
            buf = self._upload(shard)
        self._resident = None
        idx = torch.empty(L + 3, dtype=torch.int32, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.parser.stage1_shard_device(buf.data_ptr(), L, in_string, idx.data_ptr(), L + 3, stream)
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


def scan_document_shard(buf: np.ndarray, rank: int, world: int, scanner, group=None) -> ShardScan:
    """Rank `rank`'s share of ONE document.  `scanner` has parity(shard) -> 0/1 and scan(shard, in_string) ->
    (idx, n, flags).  The only collective: all_gather of one int per rank (gloo / RCCL)."""
    import torch
    import torch.distributed as dist
    cuts = clean_cuts(buf, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    mine = torch.tensor([scanner.parity(buf[lo:hi]) if hi > lo else 0], dtype=torch.int32)
    backend = dist.get_backend(group)
    if backend == "nccl":
        mine = mine.cuda()
    bits = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bits, mine, group=group)
    in_string = 0
    for r in range(rank):
        in_string ^= int(bits[r]) & 1
    if hi == lo:  # an empty shard passes the state through
        return ShardScan(lo, 0, 0, F_UNCLOSED_STRING if in_string else 0, np.zeros(0, np.int32))
    idx, n, flags = scanner.scan(buf[lo:hi], in_string)
    return ShardScan(lo, hi - lo, n, flags, idx)


def gather_to_root(local: ShardScan, root=0, group=None):
    """Concatenate all ranks' structural positions on ONE rank, moving exactly the offsets that exist ("one
    variable-length gather of index arrays"): an all_gather of (count, base) -- 16 bytes per rank -- then every rank
    sends its n uint32 offsets to `root`, which receives them straight into their final place and widens them to global
    64-bit positions there (base + offset, document_stream-inl.h:250's convention).  Nothing is padded, nobody but the
    root materialises the list, and the wire carries 4 bytes per structural, not 8.
    Returns (positions int64 [sum n] on the root, None elsewhere; per-rank counts)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    idx = local.idx
    if isinstance(idx, np.ndarray):
        idx = torch.from_numpy(idx.astype(np.int64).astype(np.int32) if idx.dtype != np.int32 else idx)
    dev = idx.device
    meta = torch.tensor([local.n, local.base], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    bases = [int(m[1]) for m in metas]
    if rank != root:
        if local.n:
            dist.send(idx[: local.n].contiguous(), dst=root, group=group)
        return None, counts
    total = sum(counts)
    raw = torch.empty(total, dtype=torch.int32, device=dev)
    at = 0
    pending = []
    for r in range(world):
        if counts[r]:
            if r == root:
                raw[at: at + counts[r]] = idx[: counts[r]]
            else:
                pending.append(dist.irecv(raw[at: at + counts[r]], src=r, group=group))
        at += counts[r]
    for w in pending:
        w.wait()
    base_of = torch.repeat_interleave(torch.tensor(bases, dtype=torch.int64, device=dev), torch.tensor(counts, dtype=torch.int64, device=dev))
    return (raw.to(torch.int64) & 0xFFFFFFFF) + base_of, counts


def gather_global_indices(local: ShardScan, group=None, document=False):
    """Concatenate all ranks' structural positions as global int64 offsets (every rank gets the result).

    Collective: all_gather(counts, bases, flags) then ONE padded all_gather of the uint32 offsets.
    Returns (positions int64 tensor [sum n], per-rank counts list, flags): flags = OR over the shards, or, with
    document=True (shards of ONE document), document_flags()."""
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
    if document:
        flags = document_flags([int(m[2]) for m in metas])
    width = max(max(counts), 1)
    mine = torch.zeros(width, dtype=torch.int32, device=dev)
    mine[: local.n] = idx[: local.n]
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = torch.cat([(parts[r][: counts[r]].to(torch.int64) & 0xFFFFFFFF) + bases[r] for r in range(world)])
    return out, counts, flags
