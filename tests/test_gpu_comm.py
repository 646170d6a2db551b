"""GPU tier: the index concatenation below the C-ABI (sjgpu_comm_gather_indices, simdjson_amd/csrc/sjgpu_comm.hip) with MORE THAN ONE RANK on
a one-GPU box.  libsjgpu opens RCCL with dlopen; these tests point it (SJGPU_RCCL_LIB, in a process of their own) at the loop-back library
of tests/stubs/rccl_loopback.cpp, whose ranks are threads and whose transfers are device-to-device copies, so the product's own code --
counts all-gather, growth round, grouped exact-count ncclSend / ncclRecv, k_widen_all with one base per rank -- runs on the real device
with a world of 2, 3 and 8.  The real RCCL with one rank per device is test_comm_two_ranks_on_two_devices (skipped on one GPU) and the
driver's multi-GPU bench."""
import json
import os
import subprocess
import sys

import pytest

from simdjson_amd import _paths, build

pytestmark = pytest.mark.gpu
WORKER = os.path.join(_paths.REPO_ROOT, "tests", "gpu_comm_worker.py")


def _run(world, shard_bytes, rounds=2, timeout=900, lib=None):
    lib = lib or build.LIB_LOOPBACK_HIP
    if not os.path.exists(lib):
        lib = build.build_rccl_loopback()
    env = dict(os.environ, SJGPU_RCCL_LIB=lib, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, WORKER, str(world), str(shard_bytes), str(rounds)], capture_output=True, timeout=timeout, env=env, cwd=_paths.REPO_ROOT)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert lines, (p.returncode, p.stdout.decode()[-1000:], p.stderr.decode()[-3000:])
    d = json.loads(lines[-1])
    assert p.returncode == 0 and d["ok"], (d, p.stderr.decode()[-2000:])
    return d


@pytest.mark.parametrize("world", [2, 3])
def test_gather_with_a_world_of_threads_on_one_device(world):
    d = _run(world, 3 << 20)
    assert d["n_ranks_seen_by_rccl"] == world and d["world"] == world and d["total_structurals"] > 100000
    assert "loopback" in d["library"]


def _real_rccl():
    """the RCCL the process already has (torch's own copy: one library per process), else ROCm's"""
    import torch
    for path in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if os.path.exists(path):
            return path
    return None


def test_the_real_rccl_with_a_world_of_one():
    """everything above runs against the stand-in.  Here the product opens the REAL librccl (the one torch ships, else ROCm's) and runs
    the whole gather -- ncclGetUniqueId, ncclCommInitRank, ncclCommCount, the counts' ncclAllGather, an (empty) ncclGroupStart / ncclGroupEnd, k_widen_all --
    with the one rank a one-GPU box has: the ten entry points bind, the library's stream semantics hold for the calls a root makes, and the positions
    are base + the reference's offsets.  (Two ranks in one real communicator need two devices: the driver's multi-GPU run.)"""
    lib = _real_rccl()
    if lib is None:
        pytest.skip("no librccl on this box")
    d = _run(1, 3 << 20, lib=lib)
    assert d["n_ranks_seen_by_rccl"] == 1 and d["world"] == 1 and d["total_structurals"] > 100000
    assert "loopback" not in d["library"] and "rccl" in d["library"]


def test_eight_shards_of_configs3_at_full_size_on_one_device():
    """BASELINE configs[3] as a dry run: 8 GiB of amazon-style NDJSON in eight newline-aligned shards of 1 GiB, every shard scanned by its
    own rank (thread) on the one device, the eight lists gathered to rank 0 through sjgpu_comm -- 64-bit global positions up to 8 GiB out of
    k_widen_all with eight bases, equal to the union of the reference's batches (batch_start + structural_indexes[i])."""
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < (40 << 30):
        pytest.skip("needs 40 GiB of free HBM")
    d = _run(8, 1 << 30, rounds=1, timeout=1500)
    assert d["n_ranks_seen_by_rccl"] == 8 and d["total_bytes"] >= (8 << 30) and d["beyond_32_bits"] and d["largest_global_position"] > (7 << 30)
    os.makedirs(os.path.join(_paths.REPO_ROOT, "gpurun_out"), exist_ok=True)
    json.dump(d, open(os.path.join(_paths.REPO_ROOT, "gpurun_out", "comm_8x1GiB_dry.json"), "w"), indent=1)
