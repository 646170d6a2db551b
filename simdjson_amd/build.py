"""In-tree native builds (explicit compiler invocations; outputs under simdjson_amd/lib/, build/tests/ and oracle/_ref/).

    libsjgpu.so            hipcc --offload-arch=gfx950   HIP kernels + C-ABI (include/sjgpu.h)   [product]
    libsjcorpus.so         gcc                           synthetic corpora                         [tooling]
    libsimdjson_mi355x.so  g++ against the reference's public headers: the simdjson::implementation
                           plug-in shim.  Needs /root/reference at BUILD time only; the built file travels.
    oracle/_ref/*.so       make -C oracle                CPU checkers                              [tests]
    build/tests/*          g++                           test programs that link the reference (plugin_test, the reference's
                           own test programs): test infrastructure, git-ignored, never under simdjson_amd/.  They are
                           built where /root/reference exists and travel prebuilt; build/tests/STAMP.json records the
                           hashes of the sources they were built from, and tests refuse a stale binary.

Run `python -m simdjson_amd.build` or call build_all(); each target is rebuilt only when a source
is newer than its output.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

from . import _paths

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
GFX_ARCH = "gfx950"


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def _run(cmd, cwd=None):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=cwd)


def _csrc(*names):
    return [os.path.join(_paths.CSRC_DIR, n) for n in names]


def build_corpus(force=False):
    srcs = _csrc("corpus.c")
    if force or _stale(_paths.LIB_CORPUS, srcs):
        os.makedirs(_paths.LIB_DIR, exist_ok=True)
        _run(["gcc", "-O2", "-std=c99", "-fopenmp", "-fPIC", "-shared", *srcs, "-o", _paths.LIB_CORPUS])
    return _paths.LIB_CORPUS


SJGPU_SOURCES = ("sjgpu_kernels.hip", "sjgpu_fused.hip", "sjgpu_small.hip", "sjgpu_finish.hip", "sjgpu_strings.hip", "sjgpu_string_stream.hip", "sjgpu_tape.hip",
                 "sjgpu_mgpu.hip", "sjgpu_comm.hip", "sjgpu_capi.hip", "sjgpu_capi_host.hip", "sjgpu_capi_stage2.hip", "stage1_finish.cpp")
SJGPU_HEADERS = ("sj_block.h", "sj_number.h", "sj_tape_rules.h", "sj_string_stream.h", "sj_xcarry.h", "sj_pow5_table.inc", "sjgpu_internal.h", "sjgpu_device.h", "sjgpu_ctx.h")


def sjgpu_source_stamp():
    """sha256 over everything libsjgpu.so is compiled from (names and bytes): what build/tests/STAMP.json records under "libsjgpu.so" when the
    library is built, and what the tests compare with -- a prebuilt library that travelled with OTHER sources is found out by content, not by mtime."""
    h = hashlib.sha256()
    for f in [*_csrc(*SJGPU_SOURCES), *_csrc(*SJGPU_HEADERS), os.path.join(_paths.INCLUDE_DIR, "sjgpu.h")]:
        h.update(os.path.relpath(f, _paths.REPO_ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def sjgpu_is_current():
    path = os.path.join(TEST_BIN_DIR, "STAMP.json")
    out = os.path.join(_paths.LIB_DIR, "libsjgpu.so")
    return os.path.exists(path) and os.path.exists(out) and json.load(open(path)).get("libsjgpu.so") == sjgpu_source_stamp()


def build_sjgpu(force=False):
    out = os.path.join(_paths.LIB_DIR, "libsjgpu.so")  # always the in-tree default, never an SJGPU_LIB override
    srcs = _csrc(*SJGPU_SOURCES)
    if force or not sjgpu_is_current():
        os.makedirs(_paths.LIB_DIR, exist_ok=True)
        _run([HIPCC, f"--offload-arch={GFX_ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
              "-I", _paths.INCLUDE_DIR, "-I", _paths.CSRC_DIR, *srcs, "-o", out, "-ldl"])  # RCCL is opened by sjgpu_comm_* on first use, not linked
        os.makedirs(TEST_BIN_DIR, exist_ok=True)
        path = os.path.join(TEST_BIN_DIR, "STAMP.json")
        d = json.load(open(path)) if os.path.exists(path) else {}
        d["libsjgpu.so"] = sjgpu_source_stamp()
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    return out


def build_plugin(force=False):
    """simdjson::implementation shim; skipped (prebuilt kept) when the reference headers are absent."""
    srcs = _csrc("plugin/mi355x_implementation.cpp")
    hdr = os.path.join(_paths.REFERENCE_DIR, "include", "simdjson.h")
    if not os.path.exists(hdr) or not all(os.path.exists(s) for s in srcs):
        return _paths.LIB_PLUGIN if os.path.exists(_paths.LIB_PLUGIN) else None
    deps = srcs + _csrc("plugin/mi355x_implementation.h") + [os.path.join(_paths.INCLUDE_DIR, "sjgpu.h")]
    if force or _stale(_paths.LIB_PLUGIN, deps):
        os.makedirs(_paths.LIB_DIR, exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DSIMDJSON_THREADS_ENABLED=1",
              "-I", os.path.join(_paths.REFERENCE_DIR, "include"), "-I", _paths.INCLUDE_DIR, "-I", _paths.CSRC_DIR,
              *srcs, "-o", _paths.LIB_PLUGIN, f"-L{_paths.LIB_DIR}", "-lsjgpu", "-Wl,-rpath,$ORIGIN"])
    return _paths.LIB_PLUGIN


TEST_BIN_DIR = os.path.join(_paths.REPO_ROOT, "build", "tests")
_RPATH = "-Wl,-rpath,$ORIGIN/../../simdjson_amd/lib"


def _stamp_sources():
    """Everything of OURS a prebuilt test binary embeds (libsjgpu.so / the plug-in .so are linked dynamically)."""
    plug = os.path.join(_paths.REPO_ROOT, "tests", "plugin")
    files = [os.path.join(_paths.INCLUDE_DIR, "sjgpu.h"), *_csrc("plugin/mi355x_implementation.h", "plugin/intree/simdjson_mi355x.patch")]
    files += sorted(os.path.join(plug, f) for f in os.listdir(plug) if f.endswith((".cpp", ".h")))
    return files


def source_stamp():
    h = hashlib.sha256()
    for f in _stamp_sources():
        h.update(os.path.relpath(f, _paths.REPO_ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _write_stamp(name):
    path = os.path.join(TEST_BIN_DIR, "STAMP.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[name] = source_stamp()
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


def binary_is_current(name):
    """True iff build/tests/<name> exists and was built from the sources as they are now."""
    path = os.path.join(TEST_BIN_DIR, "STAMP.json")
    if not (os.path.exists(path) and os.path.exists(os.path.join(TEST_BIN_DIR, name))):
        return False
    return json.load(open(path)).get(name) == source_stamp()


def build_plugin_test(force=False):
    """tests/plugin/plugin_test.cpp -> build/tests/plugin_test: unmodified reference library + our shim.
    Needs the reference (headers + oracle/_ref/simdjson_ref.o); the binary travels to the GPU box."""
    out = os.path.join(TEST_BIN_DIR, "plugin_test")
    src = os.path.join(_paths.REPO_ROOT, "tests", "plugin", "plugin_test.cpp")
    ref_obj = os.path.join(_paths.ORACLE_OUT, "simdjson_ref.o")
    hdr = os.path.join(_paths.REFERENCE_DIR, "include", "simdjson.h")
    if not (os.path.exists(hdr) and os.path.exists(ref_obj) and os.path.exists(_paths.LIB_PLUGIN)):
        return out if os.path.exists(out) else None
    if force or not binary_is_current("plugin_test") or _stale(out, [src, ref_obj, _paths.LIB_PLUGIN, _paths.LIB_CORPUS]):
        os.makedirs(TEST_BIN_DIR, exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-DSIMDJSON_THREADS_ENABLED=1", "-I", os.path.join(_paths.REFERENCE_DIR, "include"),
              "-I", os.path.join(_paths.CSRC_DIR, "plugin"), "-I", _paths.INCLUDE_DIR, src, ref_obj, "-o", out,
              f"-L{_paths.LIB_DIR}", "-lsimdjson_mi355x", "-lsjgpu", "-lsjcorpus", "-lpthread", _RPATH])
        _write_stamp("plugin_test")
    return out


REFERENCE_TESTS = {  # reference test programs that need no external data files
    "ref_unicode_tests": "tests/unicode_tests.cpp",
    "ref_dom_document_stream_tests": "tests/dom/document_stream_tests.cpp",
    "ref_dom_document_stream_fuzz_tests": "tests/dom/document_stream_fuzz_tests.cpp",
    "ref_ondemand_document_stream_tests": "tests/ondemand/ondemand_document_stream_tests.cpp",
    "ref_ondemand_document_stream_fuzz_tests": "tests/ondemand/ondemand_document_stream_fuzz_tests.cpp",
}


def build_reference_tests(force=False):
    """The reference's OWN test programs, compiled in place from /root/reference/tests and linked with the
    mi355x plug-in + an activator TU, so they exercise our backend unmodified.  Outputs build/tests/ref_*."""
    ref = _paths.REFERENCE_DIR
    ref_obj = os.path.join(_paths.ORACLE_OUT, "simdjson_ref.o")
    act = os.path.join(_paths.REPO_ROOT, "tests", "plugin", "activate_mi355x.cpp")
    built = []
    for name, rel in REFERENCE_TESTS.items():
        out = os.path.join(TEST_BIN_DIR, name)
        src = os.path.join(ref, rel)
        if not (os.path.exists(src) and os.path.exists(ref_obj) and os.path.exists(_paths.LIB_PLUGIN)):
            if os.path.exists(out):
                built.append(out)
            continue
        if force or not binary_is_current(name) or _stale(out, [src, act, ref_obj, _paths.LIB_PLUGIN]):
            os.makedirs(TEST_BIN_DIR, exist_ok=True)
            _run(["g++", "-O1", "-std=c++17", "-w", "-DSIMDJSON_THREADS_ENABLED=1", "-I", os.path.join(ref, "include"), "-I", os.path.join(ref, "tests"),
                  "-I", os.path.join(ref, "tests", "dom"), "-I", os.path.join(ref, "tests", "ondemand"),
                  "-I", os.path.join(_paths.CSRC_DIR, "plugin"), "-I", _paths.INCLUDE_DIR,
                  # data files come from the committed fixtures: /root/repo/... exists here AND on the GPU box (symlink)
                  '-DSIMDJSON_BENCHMARK_DATA_DIR="tests/golden/jsonexamples/"',
                  src, act, ref_obj, "-o", out, f"-L{_paths.LIB_DIR}", "-lsimdjson_mi355x", "-lsjgpu", "-lpthread",
                  _RPATH])
            _write_stamp(name)
        built.append(out)
    return built


INTREE_DIR = os.path.join(_paths.REPO_ROOT, "build", "intree")
INTREE_PATCH = os.path.join(_paths.CSRC_DIR, "plugin", "intree", "simdjson_mi355x.patch")
INTREE_TESTS = {"intree_basictests": "tests/dom/basictests.cpp", "intree_errortests": "tests/dom/errortests.cpp",
                # parse_many with the stream registered by document_stream::start() (the patch's second half): windows cut out of look-ahead spans
                "intree_document_stream_tests": "tests/dom/document_stream_tests.cpp", "intree_document_stream_fuzz_tests": "tests/dom/document_stream_fuzz_tests.cpp",
                # iterate_many: On-Demand's document_stream::start() registers its buffer the same way (round 4)
                "intree_ondemand_document_stream_tests": "tests/ondemand/ondemand_document_stream_tests.cpp",
                "intree_ondemand_document_stream_fuzz_tests": "tests/ondemand/ondemand_document_stream_fuzz_tests.cpp"}


def build_intree(force=False):
    """In-tree registration: a SCRATCH copy of the reference's src/ and include/ under build/intree/
    (git-ignored, never committed) gets simdjson_mi355x.patch -- an `mi355x` entry in the compile-time kernel list of
    src/implementation.cpp -- and is compiled with -DSIMDJSON_IMPLEMENTATION_MI355X=1; the reference's own basictests /
    errortests are then built against THAT library, so `-a mi355x` and SIMDJSON_FORCE_IMPLEMENTATION=mi355x select the
    backend by name, without any activation code in the program.  Outputs build/tests/intree_*."""
    ref = _paths.REFERENCE_DIR
    if not (os.path.exists(os.path.join(ref, "src", "simdjson.cpp")) and os.path.exists(_paths.LIB_PLUGIN)):
        return [os.path.join(TEST_BIN_DIR, n) for n in INTREE_TESTS if os.path.exists(os.path.join(TEST_BIN_DIR, n))]
    tree = os.path.join(INTREE_DIR, "simdjson")
    obj = os.path.join(INTREE_DIR, "simdjson_mi355x.o")
    if force or _stale(obj, [INTREE_PATCH, os.path.join(ref, "src", "implementation.cpp")]):
        shutil.rmtree(tree, ignore_errors=True)
        os.makedirs(tree, exist_ok=True)
        for sub in ("src", "include"):
            shutil.copytree(os.path.join(ref, sub), os.path.join(tree, sub))
        for dirpath, _, files in os.walk(tree):  # the reference tree is read-only; the scratch copy must not be
            os.chmod(dirpath, 0o755)
            for f in files:
                os.chmod(os.path.join(dirpath, f), 0o644)
        _run(["patch", "-p1", "-s", "-i", INTREE_PATCH], cwd=tree)
        _run(["g++", "-O2", "-std=c++17", "-DSIMDJSON_THREADS_ENABLED=1", "-DSIMDJSON_IMPLEMENTATION_MI355X=1", "-I", os.path.join(tree, "include"),
              "-I", os.path.join(tree, "src"), "-c", os.path.join(tree, "src", "simdjson.cpp"), "-o", obj])
    built = []
    for name, rel in INTREE_TESTS.items():
        out = os.path.join(TEST_BIN_DIR, name)
        src = os.path.join(ref, rel)
        if force or not binary_is_current(name) or _stale(out, [src, obj, _paths.LIB_PLUGIN]):
            os.makedirs(TEST_BIN_DIR, exist_ok=True)
            _run(["g++", "-O1", "-std=c++17", "-w", "-DSIMDJSON_THREADS_ENABLED=1", "-DSIMDJSON_IMPLEMENTATION_MI355X=1", "-I", os.path.join(tree, "include"),
                  "-I", os.path.join(ref, "tests"), "-I", os.path.join(ref, "tests", "dom"), "-I", os.path.join(ref, "tests", "ondemand"),
                  '-DSIMDJSON_BENCHMARK_DATA_DIR="tests/golden/jsonexamples/"', src, obj, "-o", out,
                  f"-L{_paths.LIB_DIR}", "-lsimdjson_mi355x", "-lsjgpu", "-lpthread", _RPATH])
            _write_stamp(name)
        built.append(out)
    return built


def build_oracle():
    _run(["make", "-s", "-C", _paths.ORACLE_DIR, f"REFERENCE={_paths.REFERENCE_DIR}"])


LOOPBACK_SRC = os.path.join(_paths.REPO_ROOT, "tests", "stubs", "rccl_loopback.cpp")
LIB_LOOPBACK_HIP = os.path.join(TEST_BIN_DIR, "librccl_loopback_hip.so")


def build_rccl_loopback(force=False):
    """tests/stubs/rccl_loopback.cpp against the real HIP / RCCL headers -> build/tests/librccl_loopback_hip.so: the stand-in for librccl
    whose ranks are threads of one process (test infrastructure: SJGPU_RCCL_LIB points libsjgpu's dlopen at it in tests/test_gpu_comm.py, so
    sjgpu_comm_gather_indices runs with a world of two and three on a one-GPU box).  Host code only; hipcc supplies the include paths."""
    if force or _stale(LIB_LOOPBACK_HIP, [LOOPBACK_SRC]):
        os.makedirs(TEST_BIN_DIR, exist_ok=True)
        _run([HIPCC, "-std=c++17", "-O2", "-fPIC", "-shared", LOOPBACK_SRC, "-o", LIB_LOOPBACK_HIP, "-lpthread"])
    return LIB_LOOPBACK_HIP


SAN_DIR = os.path.join(_paths.REPO_ROOT, "build", "san")


def build_sanitizers(force=False):
    """AddressSanitizer / ThreadSanitizer builds of the host side (scripts/sanitize.sh build: libsjgpu's host code, the plug-in, plugin_test) ->
    build/san/*, which travel to the GPU box where tests/test_plugin.py::test_sanitizers_over_the_host_shim runs them.  Needs the reference; rebuilt
    when the library's or the plug-in's sources changed (the stamp in build/tests/STAMP.json); a failure leaves the test skipped, not the build broken."""
    hdr = os.path.join(_paths.REFERENCE_DIR, "include", "simdjson.h")
    bins = [os.path.join(SAN_DIR, f"plugin_test_{s}") for s in ("address", "thread")]
    if not os.path.exists(hdr):
        return bins if all(os.path.exists(b) for b in bins) else None
    want = hashlib.sha256((sjgpu_source_stamp() + source_stamp()).encode()).hexdigest()
    path = os.path.join(TEST_BIN_DIR, "STAMP.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    if not force and d.get("san") == want and all(os.path.exists(b) for b in bins):
        return bins
    try:
        _run(["bash", os.path.join(_paths.REPO_ROOT, "scripts", "sanitize.sh"), "build"])
    except subprocess.CalledProcessError as e:
        print("[build] sanitizer builds failed:", e, flush=True)
        return None
    d = json.load(open(path)) if os.path.exists(path) else {}
    d["san"] = want
    os.makedirs(TEST_BIN_DIR, exist_ok=True)
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    return bins


def build_all(force=False):
    build_corpus(force)
    build_sjgpu(force)
    build_oracle()
    build_rccl_loopback(force)
    build_plugin(force)
    build_plugin_test(force)
    build_reference_tests(force)
    build_intree(force)
    build_sanitizers(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
