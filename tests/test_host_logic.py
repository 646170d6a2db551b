"""CPU tier: the C-ABI library loads, exports what include/sjgpu.h declares, refuses to run without a
GPU (no CPU fallback), and its host post-pass (stage1_finish.cpp) reproduces the reference's finish()
for all seven modes when fed the raw scan of the oracle."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import checkers
from simdjson_amd import _paths, build, capi
from test_oracle_golden import load, random_case_stream


@pytest.fixture(scope="module")
def lib():
    build.build_sjgpu()
    return capi.load_library()


def test_exports_match_header(lib):
    hdr = open(os.path.join(_paths.INCLUDE_DIR, "sjgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(sjgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_the_library_is_the_one_these_sources_build(lib):
    """The built libsjgpu.so travels to the GPU box; build/tests/STAMP.json records the hash of the sources it was compiled from and the tests
    (both tiers: tests/test_gpu_parity.py repeats this on the box) refuse a library that belongs to other sources -- by content, not by mtime."""
    assert build.sjgpu_is_current()
    stamp = json.load(open(os.path.join(build.TEST_BIN_DIR, "STAMP.json")))
    assert stamp["libsjgpu.so"] == build.sjgpu_source_stamp() and len(stamp["libsjgpu.so"]) == 64


def test_every_barrier_waits_for_lds(lib):
    """Round 2: hipcc dropped the `s_waitcnt lgkmcnt(0)` of a __syncthreads() at a loop header (k_minify_onchip); the
    workgroup's next ticket, stored to LDS at the end of an iteration, was read before the store had been performed
    and one run in two of a 1 GiB minify came out wrong.  The kernels now wait explicitly (lds_writes_done); this walks
    the control flow of every kernel in the built library and fails if any s_barrier can be reached from an LDS store
    without that wait."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_barriers", os.path.join(_paths.REPO_ROOT, "scripts", "check_barriers.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    objects, found = cb.check(os.path.join(_paths.LIB_DIR, "libsjgpu.so"))
    assert objects >= 5          # one code object per .hip source
    assert found == []


def test_no_tile_kernel_flushes_the_l2(lib):
    """An agent-scope fence (__threadfence()) is buffer_wbl2 + buffer_inv on this part: a write-back and an invalidation of the XCD's whole L2.  Two of
    them in the epilogue of the single-pass kernels -- executed once per wave that leaves -- took the headline from 0.49 to 0.72 ms per GiB before any
    counter was looked at (round 5, profiles/r05_selfclean_ab.txt).  The scan kernels hand data from workgroup to workgroup through agent-scope atomics and
    wait for acknowledgements; this reads the built code objects and fails if one of them contains either instruction."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_barriers", os.path.join(_paths.REPO_ROOT, "scripts", "check_barriers.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    counts = cb.l2_flushes(os.path.join(_paths.LIB_DIR, "libsjgpu.so"))
    hot = {k: v for k, v in counts.items() if any(t in k for t in ("k_fused", "k_minify_onchip", "k_stage1_", "k_minify_", "k_validate_utf8", "k_resolve", "k_docs"))}
    assert len(hot) >= 12, sorted(hot)
    assert {k: v for k, v in hot.items() if v} == {}


def test_the_streaming_hint_sits_where_it_pays(lib):
    """Round 5: the kernels of the split pipeline and validate_utf8 run at the pace of their traffic, and the traffic has a switch -- the non-temporal hint on
    loads that consume whole lines ("what the memory system does with a stream"): NDJSON 335 -> 291 us per GiB.  On lane-strided loads the
    same hint fetches every line four times (validate_utf8 181 -> 280 us).  So: the chunk loads of the streaming kernels carry it (load_chunk_stream: four per
    interior chunk; the guarded path of a document's LAST chunk keeps its four plain lane-strided loads), the masks are written and read with it, and the
    offsets -- the output -- are NOT written with it (the hint on the stores costs a pure mover 8 %)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_barriers", os.path.join(_paths.REPO_ROOT, "scripts", "check_barriers.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    h = cb.stream_hints(os.path.join(_paths.LIB_DIR, "libsjgpu.so"))

    def one(fragment):
        hits = [v for k, v in h.items() if fragment in k]
        assert len(hits) == 1, (fragment, [k for k in h if fragment in k])
        return hits[0]
    summ = one("k_stage1_summarizeILb0")
    assert summ["nt_loads"] >= 4 and summ["nt_loads"] >= summ["plain_loads"] and summ["nt_stores"] >= 2, summ  # the input streamed, the masks streamed out
    emit = one("k_stage1_emitILb0")
    assert emit["nt_loads"] >= 2 and emit["nt_stores"] == 0 and emit["plain_stores"] >= 1, emit  # the masks streamed in, the offsets stored plainly
    for name in ("k_validate_utf8", "k_minify_summarize", "k_string_parityEPK"):
        k = one(name)
        assert k["nt_loads"] >= 4 and k["nt_loads"] >= k["plain_loads"], (name, k)
    # the single-pass kernels keep their lane-strided plain loads: the streaming loader was measured in them and not kept
    pipe = one("k_fused_pipelinedILi0ELb0ELj4ELj8ELb0E")
    assert pipe["nt_loads"] == 0 and pipe["plain_loads"] >= 4, pipe


def test_the_exchange_buffer_of_load_chunk_stream_meets_no_bank_conflict():
    """load_chunk_stream (sjgpu_device.h): a chunk arrives as 256 pieces of 16 bytes in request order (piece p = quarter p % 4 of block p / 4) and leaves
    in block order (lane L reads the four quarters of block L).  Both sides are 16-byte LDS accesses, which the hardware serves sixteen lanes at a time over
    64 banks of 4 bytes: conflict-free means the sixteen lanes of a phase touch sixteen different 16-byte columns (slot mod 16).  The slot function is read out
    of the source -- this checks what the kernels compile, not a copy of it."""
    import re
    src = open(os.path.join(_paths.CSRC_DIR, "sjgpu_device.h")).read()
    m = re.search(r"u32 xbuf_slot\(u32 block, u32 quarter\) \{ return (.*?); \}", src)
    assert m, "xbuf_slot not found"
    expr = re.sub(r"(\d+)u\b", r"\1", m.group(1))
    slot = lambda block, quarter: eval(expr, {"block": block, "quarter": quarter})  # noqa: E731
    assert sorted(slot(b, q) for b in range(64) for q in range(4)) == list(range(256))  # every piece has its own slot
    for j in range(4):  # the write side: instruction j, lane i stores piece 64 j + i
        for phase in range(4):
            lanes = range(16 * phase, 16 * phase + 16)
            cols = {slot((64 * j + i) >> 2, (64 * j + i) & 3) % 16 for i in lanes}
            assert len(cols) == 16, ("write", j, phase)
    for k in range(4):  # the read side: instruction k, lane L loads quarter k of block L
        for phase in range(4):
            cols = {slot(L, k) % 16 for L in range(16 * phase, 16 * phase + 16)}
            assert len(cols) == 16, ("read", k, phase)


def test_no_kernel_spills(lib):
    """Round 2's review found 20 B of scratch in the headline kernel and in k_stage1_summarize (spills inside the hot loop of a
    kernel that is short of issue slots).  What the code objects of the built library tell the hardware to reserve
    (.private_segment_fixed_size) must be zero for every kernel -- the large-input paths in particular."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_barriers", os.path.join(_paths.REPO_ROOT, "scripts", "check_barriers.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    res = cb.kernel_resources(os.path.join(_paths.LIB_DIR, "libsjgpu.so"))
    names = " ".join(res)
    for must in ("k_fused_pipelined", "k_stage1_summarize", "k_stage1_emit", "k_minify_onchip", "k_validate_utf8"):
        assert must in names, must
    # one kernel is MEANT to use private memory: k_tape_slow_numbers keeps two 516-byte big integers per thread for the handful of
    # number tokens whose rounding needs exact arithmetic (sj_number.h); it exists so that the kernel that parses the numbers (k_tok_stage since round 6) needs none
    spilling = {k: v["scratch"] for k, v in res.items() if v.get("scratch", 0) != 0 and "k_tape_slow_numbers" not in k}
    assert spilling == {}, spilling
    assert any("k_tok_stage" in k for k in res) and any("k_tape_match" in k for k in res)
    # the occupancy the launch bounds ask for is the occupancy the register count allows (MI355X_MICROARCH.md, register file table)
    for k, v in res.items():
        if "k_fused_pipelined" in k or "k_minify_onchip" in k:
            assert v["vgpr"] <= 128, (k, v)
        if "k_tok_stage" in k:  # 37 KiB of LDS (16-bit offsets, number list, a 4 KiB window per wave): four workgroups per CU = 128 VGPRs (launch bounds)
            assert v["vgpr"] <= 128 and v["lds"] <= 40 * 1024, (k, v)
        if "k_stage1_summarize" in k:  # 31 KiB of LDS (the UTF-8 rows and load_chunk_stream's exchange buffers): FIVE workgroups of four waves per CU = five waves
            # per SIMD = 96 VGPRs; <true>, the token-stream variant: 47 KiB, three workgroups, 128 VGPRs -- what amdgpu_waves_per_eu asks for in sjgpu_kernels.hip
            tokens = "ILb1E" in k
            assert v["vgpr"] <= (128 if tokens else 96) and v["lds"] <= (53 * 1024 if tokens else 32 * 1024), (k, v)
        if "k_strs_write" in k:  # a window for HALF a chunk's worst case + the patch list: 29 KB, FIVE workgroups of four waves per CU = 96 VGPRs (rounds 3-6a carried
            # a window for a whole chunk: 49.5 KB, three workgroups -- found in session AU when 4 KiB more made it two)
            assert v["vgpr"] <= 96 and v["lds"] <= 32 * 1024, (k, v)


def test_barrier_check_finds_a_dropped_wait(tmp_path):
    """The checker is not vacuous: with the explicit waits compiled out (-DSJGPU_SELFTEST_DROP_LDS_WAIT) it reports the loop-top
    barriers hipcc leaves unguarded -- k_minify_onchip's among them, the one that produced wrong output on the GPU."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("check_barriers", os.path.join(_paths.REPO_ROOT, "scripts", "check_barriers.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    so = str(tmp_path / "nowait.so")
    csrc = os.path.join(_paths.PKG_DIR, "csrc")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSJGPU_SELFTEST_DROP_LDS_WAIT", "-I", _paths.INCLUDE_DIR, "-I", csrc,
                    os.path.join(csrc, "sjgpu_fused.hip"), "-o", so], check=True, capture_output=True)
    objects, found = cb.check(so)
    assert objects == 1
    assert any("k_minify_onchip" in func for func, _ in found), found


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert capi.device_count() == 0
    h = ctypes.c_void_p()
    assert lib.sjgpu_ctx_create(0, 1024, ctypes.byref(h)) == -1  # SJGPU_E_NO_DEVICE
    with pytest.raises(capi.SjgpuError):
        capi.DomParserImplementation(1024)


def host_stage1(lib, orc, data, mode):
    """sjgpu_stage1's host half: guards + trim, raw scan (here: the oracle's), sjgpu_stage1_finish_host."""
    a = checkers.as_u8(data)
    n_io = ctypes.c_uint32(0)
    idx = np.zeros(len(a) + 3, dtype=np.uint32)
    if len(a) == 0:
        return capi.EMPTY, 0, idx[:3]
    ln = len(a)
    if mode != 0:
        ln = lib.sjgpu_trim_partial_utf8(a.ctypes.data, ln)
        if ln == 0:
            return capi.UTF8_ERROR, 0, idx[:3]
    raw, flags = orc.scan(a[:ln])
    idx[: len(raw)] = raw
    nxt = ctypes.c_uint32(77)
    err = lib.sjgpu_stage1_finish_host(a.ctypes.data, ln, mode, idx.ctypes.data, len(raw), flags, ctypes.byref(n_io), ctypes.byref(nxt))
    assert nxt.value == (77 if err in checkers.EARLY else 0)  # next_structural_index reset exactly where finish() does
    return err, int(n_io.value), idx[: n_io.value + 3].copy()


def test_finish_host_small_cases(lib):
    orc = checkers.Oracle()
    for case in load("small_cases.json")["cases"]:
        data = bytes.fromhex(case["hex"])
        for mname, mode in checkers.MODES.items():
            obs = checkers.observable(data, mode, *host_stage1(lib, orc, data, mode))
            got = {"err": obs[0]} if len(obs) == 1 else {"err": obs[0], "n": obs[1], "idx": list(obs[2])}
            assert got == case["stage1"][mname], (data, mname)


def test_finish_host_random(lib):
    orc = checkers.Oracle()
    for seed in (1000, 1007):
        for a in random_case_stream(seed):
            for mode in range(7):
                want = checkers.observable(a, mode, *orc.stage1(a, mode))
                assert checkers.observable(a, mode, *host_stage1(lib, orc, a, mode)) == want, (bytes(a), mode)


def test_error_from_flags(lib):
    f = capi.stage1_error_from_flags
    assert f(5, 0) == capi.SUCCESS and f(0, 0) == capi.EMPTY and f(5, 4) == capi.UTF8_ERROR
    assert f(5, 1 | 2 | 4) == capi.UNCLOSED_STRING and f(5, 2 | 4) == capi.UNESCAPED_CHARS and f(0, 4) == capi.EMPTY


def test_clean_cut_definition(lib):
    """sjgpu_clean_cut(buf, len, target) = the first c >= target whose previous byte is ASCII whitespace or one of
    , : [ ] { } (0 stays 0, len when there is none)."""
    import numpy as np
    clean = set(b" \t\n\r,:[]{}")
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b'ab"\\ ,:[]{}\n\t\r\x01\xc3\xa9', np.uint8)
    for _ in range(300):
        n = int(rng.integers(0, 200))
        a = alphabet[rng.integers(0, len(alphabet), n)].copy() if n else np.zeros(0, np.uint8)
        for target in sorted({0, 1, n // 2, max(n - 1, 0), n}):
            want = 0 if target == 0 else next((c for c in range(target, n) if int(a[c - 1]) in clean), n)
            assert capi.clean_cut(a, target) == want, (bytes(a), target)


def test_two_registrations_over_one_base_leave_by_name():
    """sjgpu_stream_unregister(base) cannot say which of two registrations over one base leaves and assumes the longest did -- safe, but a short-lived
    registration of a 100-byte prefix then caps the span-served extent at 100 bytes for the life of the long stream.  sjgpu_stream_unregister_len names the one
    that leaves; the extent is the shortest of those that STAY.  (Host logic only: buffers this small are never page-locked, no device is touched.)"""
    import numpy as np
    from simdjson_amd import capi
    buf = np.full(4096, 0x20, dtype=np.uint8)
    long_, short = buf[:4000], buf[:100]
    assert capi.stream_extent(buf) == 0
    assert capi.stream_register(long_) == 0 and capi.stream_extent(buf) == 4000
    assert capi.stream_register(short) == 0 and capi.stream_extent(buf) == 100      # what BOTH vouch for
    assert capi.stream_unregister(short, named=True) == 0 and capi.stream_extent(buf) == 4000  # the short one left: the long one's extent is back
    assert capi.stream_register(short) == 0 and capi.stream_extent(buf) == 100
    assert capi.stream_unregister(short) == 0 and capi.stream_extent(buf) == 100    # unnamed: the longest is assumed gone, the extent cannot grow
    assert capi.stream_unregister(long_, named=True) == 0 and capi.stream_extent(buf) == 0    # last reference: the registration is gone
    assert capi.stream_unregister(buf) != 0                                                  # nothing left to unregister
    # a length nobody registered falls back to "the longest"
    assert capi.stream_register(long_) == 0 and capi.stream_register(short) == 0
    assert capi.stream_unregister(buf[:7], named=True) == 0 and capi.stream_extent(buf) == 100
    assert capi.stream_unregister(short, named=True) == 0 and capi.stream_extent(buf) == 0
