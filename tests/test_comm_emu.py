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


def test_the_stand_in_exports_what_the_product_binds_and_the_real_library_has_it():
    """the loop-back library must not drift from RCCL.  Three facts, no GPU needed: (1) the entry points sjgpu_comm.hip binds with dlsym
    (its SJ_SYM list) are EXACTLY the nccl* symbols the stand-in exports -- one more or one less on either side fails; (2) each of them is exported by the
    real librccl of this image; (3) their SIGNATURES agree with the real header: the stand-in is compiled by hipcc against <rccl/rccl.h> itself
    (build.build_rccl_loopback), where a definition that disagrees with the header's extern "C" prototype is a compile error -- rebuilt here from scratch
    so that the claim is checked, not inherited from a prebuilt file."""
    import re
    import shutil
    import subprocess
    from simdjson_amd import _paths, build
    src = open(os.path.join(_paths.CSRC_DIR, "sjgpu_comm.hip")).read()
    bound = set(re.findall(r'SJ_SYM\(\w+, "(nccl\w+)"\)', src))
    assert len(bound) == 10 and {"ncclSend", "ncclRecv", "ncclAllGather", "ncclGroupStart", "ncclGroupEnd", "ncclCommInitRank"} <= bound, bound
    # every binding's type comes from the header too: decltype(&ncclX) in struct rccl_api
    assert set(re.findall(r"decltype\(&(nccl\w+)\)", src)) == bound

    def exported(lib):
        out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, check=True).stdout.decode()
        return {l.split()[-1].split("@")[0] for l in out.splitlines() if l.split() and l.split()[-1].startswith("nccl")}
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        lib = build.build_rccl_loopback(force=True)  # against the real <rccl/rccl.h>: signature drift does not compile
    else:
        lib = build.LIB_LOOPBACK_HIP
    assert os.path.exists(lib)
    assert exported(lib) == bound, (sorted(exported(lib) ^ bound))
    real = next((p for p in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so") if os.path.exists(p)), None)
    if real is None:
        pytest.skip("no librccl in this image")
    assert bound <= exported(real), sorted(bound - exported(real))
