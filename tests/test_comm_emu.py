"""CPU tier: sjgpu_comm_gather_indices' OWN control flow with N > 1 ranks.  simdjson_amd/csrc/sjgpu_comm.hip is compiled as C++ against
tests/host/emu, RCCL is the loop-back library of tests/stubs/rccl_loopback.cpp -- opened through SJGPU_RCCL_LIB by the product's own
dlopen, ranks = threads -- and tests/host/test_comm_emu.cpp drives worlds of two, three and five through it: every rank as the root,
empty shards, the growth round of the staging array and the steady state behind it, a root that cannot allocate, a root whose array
is too small, bases beyond 2^32 through k_widen_all (global position = base of the sending rank + offset:
/root/reference/include/simdjson/dom/document_stream-inl.h:250).  The gloo twin (tests/test_sharded_gloo.py) shares the protocol with
that file; THIS runs the file."""
import os
import subprocess

import pytest

from simdjson_amd import _paths

CSRC = os.path.join(_paths.PKG_DIR, "csrc")
EMU = os.path.join(_paths.REPO_ROOT, "tests", "host", "emu")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    out = tmp_path_factory.mktemp("commemu")
    inc = ["-I", EMU, "-I", _paths.INCLUDE_DIR, "-I", CSRC]
    jobs = [
        subprocess.Popen(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-attributes", "-Wno-unknown-pragmas", "-x", "c++", *inc, "-c",
                          os.path.join(CSRC, "sjgpu_comm.hip"), "-o", str(out / "sjgpu_comm.o")]),
        subprocess.Popen(["g++", "-std=c++17", "-O2", *inc, "-c", os.path.join(EMU, "sj_emu.cpp"), "-o", str(out / "sj_emu.o")]),
        subprocess.Popen(["g++", "-std=c++17", "-O2", "-Wall", *inc, "-c", os.path.join(_paths.REPO_ROOT, "tests", "host", "test_comm_emu.cpp"),
                          "-o", str(out / "driver.o")]),
        subprocess.Popen(["g++", "-std=c++17", "-O2", "-Wall", "-fPIC", "-shared", *inc, os.path.join(_paths.REPO_ROOT, "tests", "stubs", "rccl_loopback.cpp"),
                          "-o", str(out / "librccl_loopback_host.so"), "-lpthread"]),
    ]
    assert all(j.wait() == 0 for j in jobs)
    exe = str(out / "test_comm_emu")
    subprocess.run(["g++", str(out / "sjgpu_comm.o"), str(out / "sj_emu.o"), str(out / "driver.o"), "-lpthread", "-ldl", "-o", exe], check=True)
    return exe, str(out / "librccl_loopback_host.so")


@pytest.mark.parametrize("world,seed", [(2, 1), (3, 2), (5, 3), (3, 2026)])
def test_the_gather_s_own_code_with_more_than_one_rank(built, world, seed):
    exe, lib = built
    p = subprocess.run([exe, str(world), str(seed)], capture_output=True, timeout=300, env=dict(os.environ, SJGPU_RCCL_LIB=lib))
    out = p.stdout.decode()
    assert p.returncode == 0, (out[-500:], p.stderr.decode()[-2000:])
    assert f"world {world}:" in out and " 0 mismatches" in out
    transfers = int(out.split("collective calls, ")[1].split(" transfers")[0])
    assert transfers >= world - 1  # the offsets travelled through ncclSend / ncclRecv, not around them


def test_without_the_library_the_entry_points_say_so(built):
    """RCCL is opened on first use: a box without it loses sjgpu_comm_* and nothing else -- SJGPU_E_HIP and a reason, no crash."""
    exe, _ = built
    p = subprocess.run([exe, "2", "1"], capture_output=True, timeout=60, env=dict(os.environ, SJGPU_RCCL_LIB="/nonexistent/librccl.so", ROCM_PATH="/nonexistent",
                                                                                 LD_LIBRARY_PATH=""))
    if p.returncode == 0:
        pytest.skip("a real librccl is on the loader's path of this box: the fallback search found it")
    assert p.returncode == 2 and b"sjgpu_comm_unique_id failed" in p.stderr
