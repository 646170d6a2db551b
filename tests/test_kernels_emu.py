"""CPU tier: the gfx950 KERNEL SOURCES of stage 1 / minify / validate_utf8 run on the CPU -- sjgpu_kernels.hip, sjgpu_fused.hip and
sjgpu_small.hip compiled as C++ against tests/host/emu (a workgroup = an OS thread, a lane = a fiber, wave operations = exchanges
between the fibers of a wave) -- and every launcher is compared with the oracle on documents that are adversarial for the carries
between spans (tests/host/test_kernels_emu.cpp).  What the GPU tier has left to prove is what hipcc and the hardware make of the same
source."""
import os
import subprocess

import pytest

from simdjson_amd import _paths

CSRC = os.path.join(_paths.PKG_DIR, "csrc")
EMU = os.path.join(_paths.REPO_ROOT, "tests", "host", "emu")


def _build(out, defines=()):
    inc = ["-I", EMU, "-I", _paths.INCLUDE_DIR, "-I", CSRC, "-I", _paths.ORACLE_DIR]
    jobs = []
    for name in ("sjgpu_kernels", "sjgpu_fused", "sjgpu_small"):
        jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O1", "-Wno-attributes", "-Wno-unknown-pragmas", *defines, "-x", "c++", *inc, "-c",
                                      os.path.join(CSRC, name + ".hip"), "-o", str(out / (name + ".o"))]))
    jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O2", *inc, "-c", os.path.join(EMU, "sj_emu.cpp"), "-o", str(out / "sj_emu.o")]))
    jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O2", "-Wno-attributes", *inc, "-c",
                                  os.path.join(_paths.REPO_ROOT, "tests", "host", "test_kernels_emu.cpp"), "-o", str(out / "driver.o")]))
    jobs.append(subprocess.Popen(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-c", os.path.join(_paths.ORACLE_DIR, "sj_oracle.c"),
                                  "-o", str(out / "sj_oracle.o")]))
    assert all(j.wait() == 0 for j in jobs)
    exe = str(out / "test_kernels_emu")
    objs = [str(out / f) for f in ("sjgpu_kernels.o", "sjgpu_fused.o", "sjgpu_small.o", "sj_emu.o", "driver.o", "sj_oracle.o")]
    subprocess.run(["g++", *objs, "-lpthread", "-lm", "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("emu"))


@pytest.mark.parametrize("seed,docs,max_kib", [(1, 150, 200), (2026, 150, 200), (7, 10, 2500)])
def test_kernel_sources_against_the_oracle(emu, seed, docs, max_kib):
    p = subprocess.run([emu, str(seed), str(docs), str(max_kib), "all"], capture_output=True, timeout=900)
    assert p.returncode == 0, (p.stdout.decode()[-500:], p.stderr.decode()[-3000:])
    assert f"{docs} documents" in p.stdout.decode() and " 0 mismatches" in p.stdout.decode()


@pytest.mark.parametrize("waves", ["4", "16"])
def test_the_other_workgroup_shapes_of_the_single_pass_kernels(emu, waves):
    """The single-pass kernels run with eight waves per workgroup since round 4 (128 KiB / 64 KiB tiles: half the tickets and look-backs per byte); the
    four-wave shape of rounds 1-3 (SJGPU_PIPE_WAVES=4, SJGPU_MINIFY_WAVES=4) and minify's sixteen-wave one stay selectable for A/B runs -- and stay right."""
    env = dict(os.environ, SJGPU_PIPE_WAVES=waves, SJGPU_MINIFY_WAVES=waves)
    p = subprocess.run([emu, "11", "60", "300", "fused"], capture_output=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout.decode()[-500:], p.stderr.decode()[-3000:])
    assert "60 documents" in p.stdout.decode() and " 0 mismatches" in p.stdout.decode()


def test_segments_with_a_handful_of_candidates(emu):
    """Round 6: k_stage1_summarize ships a segment with at most 32 candidates as a LIST (one word per candidate: offset in the segment, bit 31 its string_tail bit)
    instead of two mask planes, k_stage1_emit selects by hypothesis and copies.  Documents made of such segments only: 0, 1, 2, 31, 32, 33, 34, 64 ... candidates per
    16 KiB, in whitespace, inside scalar runs, inside strings, behind backslashes; resolved by a newline in the first chunk or not (then the list carries both
    hypotheses); quotes opening and closing across segments, so that both hypotheses get selected -- the split pipeline against the oracle, list and flags."""
    for seed, docs, kib in (("5", "60", "400"), ("77", "30", "1500")):
        p = subprocess.run([emu, seed, docs, kib, "sparse"], capture_output=True, timeout=1800)
        assert p.returncode == 0, (p.stdout.decode()[-500:], p.stderr.decode()[-3000:])
        assert f"{docs} documents" in p.stdout.decode() and " 0 mismatches" in p.stdout.decode()


def test_the_sparse_documents_reach_the_list_road(tmp_path):
    """... and the check has teeth: with the list road's selection by hypothesis compiled out (every segment taken as if it began outside a string) the same
    documents must fail -- they do reach the list road, with segments that begin inside strings."""
    exe = _build(tmp_path, ("-DSJGPU_SELFTEST_SPARSE_IGNORES_HYPOTHESIS",))
    p = subprocess.run([exe, "5", "60", "400", "sparse"], capture_output=True, timeout=1800)
    assert p.returncode != 0 and "MISMATCH" in p.stderr.decode(), "the split pipeline passes without the list road's selection"


def test_the_direct_kernel(emu):
    """Round 6: k_stage1_direct (sjgpu_fused.hip) -- the split pipeline's scan with additive prefixes and the emission inside it, for PLAIN input (every 16 KiB
    segment pins its own string state at a control character of its first chunk).  Three documents in four here are NDJSON-like, a third of them broken
    (control character inside a string, bad UTF-8, unclosed string): the kernel answers exactly what the oracle says -- list, count, flags -- or gives up
    (SJGPU_F_INTERNAL: the caller re-runs the split pipeline) and either way leaves the self-cleaning workspace as it found it.  Most must complete."""
    import re
    for seed, docs, kib in (("11", "80", "300"), ("7", "24", "2500"), ("2026", "60", "700")):
        p = subprocess.run([emu, seed, docs, kib, "direct"], capture_output=True, timeout=1800)
        out = p.stdout.decode()
        assert p.returncode == 0 and f"{docs} documents" in out and " 0 mismatches" in out, (out[-500:], p.stderr.decode()[-3000:])
        m = re.search(r"direct: (\d+) completed, (\d+) gave up", out)
        assert m and int(m.group(1)) >= int(docs) // 2 and int(m.group(2)) >= 1, out[-300:]


@pytest.mark.parametrize("part", [1, 2, 3, 4])
def test_the_documents_reach_every_part_of_the_escape_carry(tmp_path, part):
    """The kernels carry the escape state in their scan (sj_xcarry.h: spans that assume, an x word per summary).  With a part of the x
    words compiled out -- 1 all of it, 2 the patched candidate bit, 3 the dependence of a successor's x on the span's own, 4 the flip of
    the in-string hypothesis -- the same documents must FAIL: the comparison above has teeth for every rule."""
    exe = _build(tmp_path, ("-DSJGPU_SELFTEST_NO_XW=%d" % part,))
    p = subprocess.run([exe, "1", "150", "200", "all"], capture_output=True, timeout=900)
    assert p.returncode != 0 and "MISMATCH" in p.stderr.decode(), "the kernels pass without part %d of the x words" % part
