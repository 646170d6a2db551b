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
    if rank == 0:
        whole, wflags = orc.scan(buf)
        ok = (flags == wflags == 0 and sum(counts) == len(whole)
              and np.array_equal(positions.numpy(), whole.astype(np.int64)))
        q.put((ok, counts, len(whole)))
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
