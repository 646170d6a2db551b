"""CPU tier, world_size 2 over gloo: the N>1 path (newline sharding, slice bases, index concatenation) of
simdjson_amd.sharded.  The per-shard scan is injected (here: the oracle, as the stand-in for the HIP call),
so what is tested is exactly the multi-rank logic that bench.py / a parse_many driver runs on GPUs."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import checkers
from simdjson_amd import corpus, sharded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, size, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf, _ = corpus.amazon_ndjson(size, seed)
    orc = checkers.Oracle()

    def scan_fn(shard):
        idx, flags = orc.scan(shard)
        return idx.view(np.int32), len(idx), flags

    local = sharded.scan_shard(buf, rank, world, scan_fn)
    positions, counts, flags = sharded.gather_global_indices(local)
    rooted, rcounts = sharded.gather_to_root(local)  # the variable-length gather bench.py --gpus N times
    if rank == 0:
        whole, wflags = orc.scan(buf)
        ok = (flags == wflags == 0 and sum(counts) == len(whole)
              and np.array_equal(positions.numpy(), whole.astype(np.int64))
              and rcounts == counts and np.array_equal(rooted.numpy(), whole.astype(np.int64)))
        q.put((ok, counts, len(whole)))
    else:
        assert rooted is None and rcounts == counts
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ndjson_shards_concatenate_to_the_single_scan(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 3 << 20, 9, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, counts, total = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, (counts, total)
    assert len(counts) == world and all(c > 0 for c in counts)


def test_newline_cuts_are_line_aligned():
    buf, _ = corpus.amazon_ndjson(1 << 20, 4)
    for parts in (1, 2, 7, 8):
        cuts = sharded.newline_cuts(buf, parts)
        assert cuts[0] == 0 and cuts[-1] == len(buf) and cuts == sorted(cuts)
        for c in cuts[1:-1]:
            assert buf[c - 1] == 0x0A


# ---- ONE document across ranks: clean cuts + one all_gather of a parity bit per rank ------------------
class OracleDocumentScanner:
    """Stand-in for GpuShardScanner.parity / .scan: a shard that begins inside a string is scanned as
    '"' + shard (the quote supplies the in-string state and nothing else: it is neither a scalar nor a
    backslash), then the quote's own offset is dropped and the rest shifted back by one."""

    def __init__(self):
        self.orc = checkers.Oracle()

    def parity(self, shard):
        return self.orc.scan(shard)[1] & 1  # SJGPU_F_UNCLOSED_STRING of a zero-carry scan = odd number of quotes

    def scan(self, shard, in_string):
        if not in_string:
            idx, flags = self.orc.scan(shard)
        else:
            idx, flags = self.orc.scan(np.concatenate([np.frombuffer(b'"', np.uint8), shard]))
            assert idx[0] == 0
            idx = idx[1:] - 1
        return idx.astype(np.uint32).view(np.int32), len(idx), flags


def _mixed_document(size, seed):
    """Strings with blanks, escapes, multi-byte UTF-8 and structurals inside them: cuts land inside strings."""
    buf, _ = corpus.twitter_like(size, seed)
    return buf


def _doc_worker(rank, world, port, size, seed, damage, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = _mixed_document(size, seed).copy()
    if damage == "unclosed":
        buf = np.concatenate([buf, np.frombuffer(b' "dangling', np.uint8)])
    elif damage == "ctrl":
        pos = int(np.flatnonzero(buf == ord('"'))[len(buf) // 300]) + 1  # just inside some string
        buf[pos] = 0x01
    local = sharded.scan_document_shard(buf, rank, world, OracleDocumentScanner())
    positions, counts, flags = sharded.gather_global_indices(local, document=True)
    if rank == 0:
        whole, wflags = checkers.Oracle().scan(buf)
        ok = (flags == wflags and sum(counts) == len(whole) and np.array_equal(positions.numpy(), whole.astype(np.int64)))
        q.put((ok, counts, len(whole), flags, wflags))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,damage", [(2, None), (3, None), (2, "unclosed"), (2, "ctrl")])
def test_document_shards_concatenate_to_the_single_scan(world, damage):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_doc_worker, args=(r, world, port, 2 << 20, 21, damage, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, counts, total, flags, wflags = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, (counts, total, flags, wflags)
    assert (flags != 0) == (damage is not None)


def test_clean_cuts_leave_only_the_string_bit():
    """Sequential restatement of the exchange on many random adversarial documents: scanning the shards one after
    the other with the carried in-string bit reproduces the whole-buffer scan, for every number of shards."""
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b'"\\\\ ,:[]{}ab1\n\t\xc3\xa9\xe2\x82\xac\x01', np.uint8)
    sc = OracleDocumentScanner()
    orc = checkers.Oracle()
    for trial in range(300):
        n = int(rng.integers(1, 400))
        buf = alphabet[rng.integers(0, len(alphabet), n)].copy()
        whole, wflags = orc.scan(buf)
        for parts in (2, 3, 7):
            cuts = sharded.clean_cuts(buf, parts)
            assert cuts[0] == 0 and cuts[-1] == n and cuts == sorted(cuts)
            state, got, flags = 0, [], []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi == lo:
                    flags.append(sharded.F_UNCLOSED_STRING if state else 0)
                    continue
                p = sc.parity(buf[lo:hi])
                idx, cnt, f = sc.scan(buf[lo:hi], state)
                got.append(idx[:cnt].view(np.uint32).astype(np.int64) + lo)
                flags.append(f)
                assert (f & 1) == (state ^ p)
                state ^= p
            got = np.concatenate(got) if got else np.zeros(0, np.int64)
            assert np.array_equal(got, whole.astype(np.int64)), (trial, parts, cuts)
            # a cut follows an ASCII byte, so even the UTF-8 verdict composes exactly
            assert sharded.document_flags(flags) == wflags, (trial, parts)
