"""CPU tier: sjgpu_mgpu_* (simdjson_amd/csrc/sjgpu_mgpu.hip -- one host buffer, one shard and one host thread per GPU, one bit per shard exchanged through host
memory) with DISTINCT devices.  VERDICT r05 #7(b): the GPU tier can only list the one device of its box several times.  Here the file is compiled as C++ against
tests/host/emu with three fake devices: every thread has a current device, every allocation and stream belongs to the device current when it was made, every
HIP call that names one checks that it runs under its own device; the per-device contexts are mocks of the C-ABI entry points the driver calls, which check
the same discipline and answer from the oracle with the shard semantics of include/sjgpu.h.  Device lists {0,1}, {1,0}, {0,1,2}, {2,0,1,2}, {1}, {2,2,1};
stage 1 in three modes, minify and validate_utf8 of every document against the oracle over the WHOLE document (tests/host/test_mgpu_emu.cpp)."""
import os
import subprocess

import pytest

from simdjson_amd import _paths

CSRC = os.path.join(_paths.PKG_DIR, "csrc")
EMU = os.path.join(_paths.REPO_ROOT, "tests", "host", "emu")


def _build(out, defines=()):
    inc = ["-I", EMU, "-I", _paths.INCLUDE_DIR, "-I", CSRC, "-I", _paths.ORACLE_DIR]
    dev = ["-DSJ_EMU_DEVICES=3"]
    jobs = [
        subprocess.Popen(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-attributes", "-Wno-unknown-pragmas", *dev, *defines, "-x", "c++", *inc, "-c",
                          os.path.join(CSRC, "sjgpu_mgpu.hip"), "-o", str(out / "sjgpu_mgpu.o")]),
        subprocess.Popen(["g++", "-std=c++17", "-O2", *dev, *inc, "-c", os.path.join(EMU, "sj_emu.cpp"), "-o", str(out / "sj_emu.o")]),
        subprocess.Popen(["g++", "-std=c++17", "-O2", "-Wall", *dev, *inc, "-c", os.path.join(_paths.REPO_ROOT, "tests", "host", "test_mgpu_emu.cpp"),
                          "-o", str(out / "driver.o")]),
        subprocess.Popen(["g++", "-std=c++17", "-O2", *inc, "-c", os.path.join(CSRC, "stage1_finish.cpp"), "-o", str(out / "stage1_finish.o")]),
        subprocess.Popen(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-c", os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"), "-o", str(out / "sj_oracle.o")]),
    ]
    assert all(j.wait() == 0 for j in jobs)
    exe = str(out / "test_mgpu_emu")
    subprocess.run(["g++", *[str(out / f) for f in ("sjgpu_mgpu.o", "sj_emu.o", "driver.o", "stage1_finish.o", "sj_oracle.o")], "-lpthread", "-lm", "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("mgpuemu"))


@pytest.mark.parametrize("seed,docs", [(1, 40), (2026, 40)])
def test_shards_on_distinct_devices_equal_the_whole_scan(built, seed, docs):
    p = subprocess.run([built, str(seed), str(docs)], capture_output=True, timeout=900)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0 and f"{docs} documents" in out and " 0 mismatches" in out, (out[-500:], err[-3000:])
    assert err.count("DEVICE VIOLATION") == 2  # the checker's own self-test, announced in front of them; none from the product


def test_the_device_checker_has_teeth(tmp_path):
    """With the shard threads' hipSetDevice compiled out of sjgpu_mgpu.hip every thread stays on device 0: the emulation must find the copies, launches and
    context calls made for devices 1 and 2 under it -- what a box that lists its one GPU several times can never see."""
    exe = _build(tmp_path, ("-DSJGPU_SELFTEST_MGPU_NO_SETDEVICE",))
    p = subprocess.run([exe, "1", "6"], capture_output=True, timeout=900)
    assert p.returncode != 0 and p.stderr.decode().count("DEVICE VIOLATION") > 10 and "under the wrong device" in p.stderr.decode()
